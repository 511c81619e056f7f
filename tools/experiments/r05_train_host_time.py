"""Is the pipelined training loop host-bound?  Host time spent inside step() per iteration vs the wall time per iteration (round 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')]
import torch
import bench
import __graft_entry__ as ge
ge.build()
from ppyolo_hip import synth
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
from ppyolo_hip.train import TrainStep, lr_at
dev = torch.device('cuda', 0)
wl = bench.WORKLOADS['r50vd_608']
model, sd, cfg = bench.build_model(wl['cfg'], dev)
hc = cfg.head
x = synth.synth_images(8, 608, seed=1234).to(dev)
bb, cc, ss = synth_ground_truth(8, 50)
targets = [torch.from_numpy(t).to(dev) for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, 608)]
gt = torch.from_numpy(bb).to(dev)
ts = TrainStep(model, cfg, 1)
lr = lr_at(4000, cfg)
for pre in (True, False):
    ts._pref = None
    for _ in range(5):
        ts.step(x, gt, targets, lr, next_x=x if pre else None)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(30):
        h0 = time.perf_counter()
        ts.step(x, gt, targets, lr, next_x=x if pre else None)
        host += time.perf_counter() - h0
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('prefetch %s: wall %.2f ms per step, host time inside step() %.2f ms per step, all steps issued after %.2f ms per step' % (
        pre, wall / 30 * 1e3, host / 30 * 1e3, t_issue / 30 * 1e3))
