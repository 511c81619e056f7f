#!/usr/bin/env python
"""Where do the small memcpy / fill / elementwise launches of a training step come from?  Runs three steps of bench.py's
training setup under torch.profiler (with Python stacks) and prints, per aten op that launches device work outside the
library's own kernels, the call count per step and the train.py / ops.py line it comes from.   (GPU box)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import bench  # noqa: E402


def main():
    from ppyolo_hip import synth
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    from ppyolo_hip.train import TrainStep, lr_at
    dev = torch.device('cuda:0')
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
    steps = 2
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            ts.step(x, gt, targets, lr)
        torch.cuda.synchronize()
    by = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.name in ('aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view', 'aten::slice',
                                                            'aten::select', 'aten::as_strided', 'aten::expand', 'aten::permute', 'aten::reshape'):
            continue
        where = '?'
        for fr in ev.stack or []:
            if 'ppyolo_hip' in fr or 'bench.py' in fr:
                where = fr.strip()
                break
        by[(ev.name, where)] += 1
    for (name, where), n in sorted(by.items(), key=lambda kv: -kv[1])[:60]:
        print('%6.1f / step  %-28s %s' % (n / steps, name, where[-110:]))


if __name__ == '__main__':
    main()
