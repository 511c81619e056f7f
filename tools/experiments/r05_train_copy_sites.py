"""Where do the ~73 device-to-device buffer copies of one training step come from?  Counts Tensor.copy_ / clone / zero_ / fill_ calls
of ONE TrainStep.step by call site (round 5)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
from ppyolo_hip import synth  # noqa: E402
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth  # noqa: E402
from ppyolo_hip.train import TrainStep, lr_at  # noqa: E402

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
for _ in range(3):
    ts.step(x, gt, targets, lr)
torch.cuda.synchronize()
sites = collections.Counter()


def wrap(name, fn):
    def w(self, *a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        key = name + ' <- ' + ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st))
        extra = ''
        if name == 'copy_' and a and torch.is_tensor(a[0]):
            extra = ' [%s%s -> %s%s]' % (tuple(a[0].shape), '' if a[0].is_contiguous() else ' nc', tuple(self.shape), '' if self.is_contiguous() else ' nc')
        sites[key + extra] += 1
        return fn(self, *a, **k)
    return w


for nm in ('copy_', 'clone', 'zero_', 'fill_', 'contiguous', 'add_'):
    setattr(torch.Tensor, nm, wrap(nm, getattr(torch.Tensor, nm)))
real_zeros, real_empty = torch.zeros, torch.empty


def zeros(*a, **k):
    st = traceback.extract_stack(limit=3)[:-1]
    sites['torch.zeros <- ' + ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st))] += 1
    return real_zeros(*a, **k)


torch.zeros = zeros
ts.step(x, gt, targets, lr)
torch.cuda.synchronize()
for k, v in sites.most_common(40):
    print('%4d  %s' % (v, k))
