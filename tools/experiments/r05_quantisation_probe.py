#!/usr/bin/env python
"""Round 5: what a perfect stream-K could give the five 1x1 shapes the round-4 review named.  Each layer with its table
configuration at batch 8 and at batches whose tile count fills whole rounds of 256 CUs better (and 16 / 32): the time per IMAGE at
the best batch is what a schedule without quantisation loss and with amortised fill / drain would reach for the batch-8 layer --
an upper bound for stream-K, before the cost of its fix-up.  Launches are timed as 16 captured graph nodes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch  # noqa: E402
from ppyolo_hip import ops  # noqa: E402

SHAPES = [((38, 38, 1024, 256, 0), 98, 5), ((19, 19, 2048, 512, 0), 99, 4), ((76, 76, 512, 128, 0), 54, 3),
          ((38, 38, 256, 1024, 1), 92, 5), ((76, 76, 128, 512, 1), 47, 2)]
out = []
for (H, W, C, K, use_res), cfg, per_step in SHAPES:
    rows = []
    for N in (8, 9, 10, 11, 12, 14, 16, 24, 32):
        x = torch.randn(N, H, W, C, device='cuda')
        w = torch.randn(K, 1, 1, C, device='cuda') * 0.05
        sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
        y = torch.empty(N, H, W, K, device='cuda')
        res = torch.randn(N, H, W, K, device='cuda') if use_res else None
        ws = torch.empty(16 << 20, device='cuda')
        w3, wf = ops.split_weights_bf16x3(w), ops.split_weights_f16x2(w, sc)
        amax, amax_out = ops.amax_slots(x), ops.amax_slots(device='cuda', N=N)

        def run():
            ops.conv2d_bn_act(ops.View(x), w, sc, sh, ops.View(y), 1, 0, 'relu', residual=None if res is None else ops.View(res), cfg=cfg,
                              splitk=1, ws=ws, w_x3=w3, w_f16=wf, amax_in=amax, amax_out=amax_out)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(16):
                run()
        g.replay()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            e.synchronize()
            best = min(best, s.elapsed_time(e) / 16)
        g.reset()
        rows.append((N, best * 1e3, best * 1e3 / N, 2.0 * N * H * W * K * C / best / 1e9))
    b8 = rows[0]
    bb = min(rows, key=lambda r: r[2])
    out.append('%dx%d C%d -> K%d%s, cfg %d (x%d per step): batch 8 %.1f us = %.2f us/image (%.0f TF); best batch %d: %.2f us/image (%.0f TF) -> a '
               'perfect schedule would save %.1f us per launch, %.0f us per step' % (H, W, C, K, ' + shortcut' if use_res else '', cfg, per_step, b8[1], b8[2], b8[3],
                                                                                 bb[0], bb[2], bb[3], b8[1] - 8 * bb[2], per_step * (b8[1] - 8 * bb[2])))
    out.append('    ' + '  '.join('N=%d %.1f us (%.2f/img)' % (r[0], r[1], r[2]) for r in rows))
print('\n'.join(out))
with open(os.path.join(ROOT, 'gpurun_out', 'r05', 'quantisation_probe.txt'), 'w') as fh:
    fh.write('\n'.join(out) + '\n')
