#!/usr/bin/env python
"""Time the fused DCNv2 kernel (csrc/dcn_fused.hip) on the three deformable layers of R50vd-608, batch 8: every
math scheme x tile x split-K, each as 20 nodes of a captured graph.  usage: dcn_bench.py [N]   (-> stdout table)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch
from ppyolo_hip import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
TILES = ['128x128', '64x128', '128x64', '64x64', '64x256', '128x256']
TILES8 = ['64x512 (8 waves)', '64x256 (8 waves)', '128x512 (8 waves)']
for H, stride in ((38, 2), (19, 1)):
    C = K = 512
    g = torch.Generator().manual_seed(H)
    Ho, Wo = ops.dcn_out_hw(H, H, stride, 1)
    x = torch.randn(N, H, H, C, generator=g).cuda()
    om = torch.randn(N, Ho, Wo, 27, generator=g).cuda() * 1.5
    w = (torch.randn(K, 3, 3, C, generator=g) * (1.0 / (9 * C)) ** 0.5).cuda()
    scale, shift = torch.ones(K).cuda(), torch.zeros(K).cuda()
    w3, wf, amax = ops.split_weights_bf16x3(w), ops.split_weights_f16x2(w, scale), ops.amax_slots(x)
    y = torch.empty(N, Ho, Wo, K, device='cuda')
    ws = torch.empty(16 * N * Ho * Wo * K, device='cuda')
    gflop = 2.0 * N * Ho * Wo * K * 9 * C / 1e9
    nt = ops.DCN_TILES_4W
    print('DCNv2 %dx%d stride %d, N=%d, C=K=512: %.2f GFLOP' % (H, H, stride, N, gflop))
    for mode, name in enumerate(('fp32', 'bf16x3', 'f16x2')):
        best = None
        for t in range(nt):
            row = []
            for s in (1, 2, 3, 4, 6, 9, 12):
                def run():
                    ops.dcnv2(ops.View(x), w, scale, shift, ops.View(om), ops.View(y), stride, 1, 'relu', ws, cfg=mode * nt + t,
                              splitk=s, w_x3=w3, w_f16=wf, amax_in=amax)
                run()
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
                us = st.elapsed_time(en) * 1e3 / 100
                row.append(us)
                if best is None or us < best[0]:
                    best = (us, TILES[t], s)
            print('  %-6s tile %-8s split 1/2/3/4/6/9/12: %s us' % (name, TILES[t], ' '.join('%6.1f' % u for u in row)))
        print('  %-6s best %.1f us = %.0f TFLOP/s  (tile %s, split-K %d)' % (name, best[0], gflop / best[0] * 1e3, best[1], best[2]))
    for t8 in range(ops.dcnv2_num_configs() - 3 * nt):      # the eight-wave f16x2 tiles
        row = []
        for s in (1, 2, 3, 4, 6, 9, 12):
            def run():
                ops.dcnv2(ops.View(x), w, scale, shift, ops.View(om), ops.View(y), stride, 1, 'relu', ws, cfg=3 * nt + t8, splitk=s, w_x3=w3,
                          w_f16=wf, amax_in=amax)
            run()
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
            row.append(st.elapsed_time(en) * 1e3 / 100)
        print('  f16x2  tile %-18s split 1/2/3/4/6/9/12: %s us  (best %.0f TFLOP/s)' % (TILES8[t8], ' '.join('%6.1f' % u for u in row), gflop / min(row) * 1e3))
