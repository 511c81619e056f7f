#!/usr/bin/env python
"""Layer-level timing of the stage-4 "expand" layer of R50vd-608 bs 8 (38x38, C256 -> K1024, + shortcut + ReLU) on every f16x2
tile configuration and the streaming kernel where it applies (20 launches per captured graph).  (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch
from ppyolo_hip import ops
from ppyolo_hip._lib import PPYoloHipError

for (N, H, C, K) in ((8, 38, 256, 1024), (8, 76, 128, 512), (8, 19, 512, 2048)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, H, C, generator=g).cuda()
    wk = (torch.randn(K, 1, 1, C, generator=g) * 0.06).cuda()
    sc, sh = torch.ones(K).cuda(), torch.zeros(K).cuda()
    r = torch.randn(N, H, H, K, generator=g).cuda()
    wf = ops.split_weights_f16x2(wk, sc)
    y = torch.empty(N, H, H, K).cuda()
    am = ops.amax_slots(x)
    res = []
    first = ops.stream_first_cfg()
    for cfg in list(range(40, 67)) + list(range(85, 94)) + [first, first + 1] + list(range(ops.ws_first_cfg(), ops.ws_first_cfg() + 12)):
        def run():
            ops.conv2d_bn_act(ops.View(x), wk, sc, sh, ops.View(y), 1, 0, 'relu', residual=ops.View(r), cfg=cfg, splitk=1, w_f16=wf, amax_in=am,
                              amax_out=ops.amax_slots(N=N, device=y.device))
        try:
            run()
        except (PPYoloHipError, AssertionError):
            continue
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                run()
        gr.replay()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(5):
            gr.replay()
        en.record()
        en.synchronize()
        res.append((st.elapsed_time(en) * 1e3 / 100, cfg))
    res.sort()
    mb = (N * H * H * (C + 2 * K) * 4) / 1e6
    print('expand %dx%d C%d -> K%d bs %d (%.0f MB): streaming cfgs %d / %d' % (H, H, C, K, N, mb, first, first + 1))
    for us, cfg in res[:6] + [t for t in res if t[1] in (first, first + 1)]:
        print('   cfg %3d  %6.1f us  %.2f TB/s' % (cfg, us, mb / us))
