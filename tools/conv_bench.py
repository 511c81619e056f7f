#!/usr/bin/env python
"""Micro-benchmark of ppy_conv2d_bn_act_f32 on given layer shapes / tile configs.
usage: conv_bench.py "N,H,W,C,K,R,stride[,res]" cfg[,cfg...] [splitk]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch  # noqa: E402
from ppyolo_hip import ops  # noqa: E402


def main():
    shp = [int(v) for v in sys.argv[1].split(',')]
    N, H, W, C, K, R, stride = shp[:7]
    use_res = len(shp) > 7 and shp[7]
    cfgs = [int(c) for c in sys.argv[2].split(',')]
    splitk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    pad = (R - 1) // 2
    Ho, Wo = ops.conv_out_hw(H, W, R, R, stride, pad)
    x = torch.randn(N, H, W, C, device='cuda')
    w = torch.randn(K, R, R, C, device='cuda') * 0.05
    sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    y = torch.empty(N, Ho, Wo, K, device='cuda')
    res = torch.randn(N, Ho, Wo, K, device='cuda') if use_res else None
    ws = torch.empty(64 << 20, device='cuda')
    w3 = ops.split_weights_bf16x3(w)
    wf = ops.split_weights_f16x2(w, sc)
    amax = ops.amax_slots(x)
    amax_out = ops.amax_slots(device='cuda', N=N) if os.environ.get('PPY_BENCH_TRACK', '1') == '1' else None
    flops = 2.0 * N * Ho * Wo * K * R * R * C
    for cfg in cfgs:
        def run():
            ops.conv2d_bn_act(ops.View(x), w, sc, sh, ops.View(y), stride, pad, 'relu',
                              residual=None if res is None else ops.View(res), cfg=cfg, splitk=splitk, ws=ws,
                              w_x3=w3, w_f16=wf, amax_in=amax, amax_out=amax_out)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record()
            e.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        print('dbg=%s shape=%s cfg=%d split=%d: %.4f ms  %.1f TF' % (os.environ.get('PPY_CONV_DBG', '0'), sys.argv[1], cfg,
                                                                 splitk, best, flops / best / 1e9))


if __name__ == '__main__':
    main()
