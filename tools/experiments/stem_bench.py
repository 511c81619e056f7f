#!/usr/bin/env python
"""Time the stem launch of R50vd-608 bs 8 in its two forms (fp32 fma chain as row segments / bf16x3 on the MFMA), back to back."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ppyolo_hip import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
N, S = 8, 608
x = torch.randn(N, 3, S, S, generator=g).cuda()
w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2).cuda()
sc, sh = (torch.rand(32, generator=g) + 0.5).cuda(), torch.randn(32, generator=g).cuda()
y = torch.zeros(N, S // 2, S // 2, 32).cuda()
amax = ops.amax_slots(device='cuda', N=N)
for mfma in (False, True, False, True):
    for _ in range(3):
        ops.stem_conv(x, w, sc, sh, ops.View(y), 'relu', amax_out=amax, mfma=mfma)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.stem_conv(x, w, sc, sh, ops.View(y), 'relu', amax_out=amax, mfma=mfma)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / 20
        best = t if best is None else min(best, t)
    byt = x.numel() * 4 + y.numel() * 4
    print('stem %-22s %6.1f us  (%.0f MB algorithmic = %.2f TB/s)' % ('bf16x3 MFMA' if mfma else 'fp32 fma, row segments', best * 1e3, byt / 1e6, byt / best / 1e9))
