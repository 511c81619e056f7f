#!/usr/bin/env python
"""A/B on one box: the plain `forward_padded` of R50vd-608 bs 8 as one executor vs as two half-batch lanes
(PPYOLO_HIP_FORWARD_SPLIT=1, runtime.PlanCache.run_split), alternating; results compared."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from ppyolo_hip import synth  # noqa: E402

dev = torch.device('cuda', 0)
wl = sys.argv[1] if len(sys.argv) > 1 else 'r50vd_608'
W = bench.WORKLOADS[wl]
models = {}
for split in (0, 1):
    m, _, _ = bench.build_model(W['cfg'], dev)
    m._plans.split_forward = bool(split)
    m.pin_weights()
    models[split] = m
x = synth.synth_images(8, W['size']).to(dev)
ims = synth.synth_im_size(8).to(dev)
outs = {}
for split, m in models.items():
    d, c, k = m.forward_padded(x, ims)
    torch.cuda.synchronize()
    outs[split] = (d.clone(), c.clone(), k.clone())
print('counts equal', torch.equal(outs[0][1], outs[1][1]), 'keep equal', torch.equal(outs[0][2], outs[1][2]),
      'max |dets diff|', float((outs[0][0] - outs[1][0]).abs().max()))
for rnd in range(3):
    for split, m in models.items():
        for _ in range(20):
            m.forward_padded(x, ims)
        torch.cuda.synchronize()
        n, t0 = 300, time.perf_counter()
        for _ in range(n):
            m.forward_padded(x, ims)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('round %d  split=%d  %.1f img/s  %.3f ms per forward' % (rnd, split, 8 * n / dt, dt / n * 1e3), flush=True)
