"""The DCNv2 offset convolutions (C512 -> K27, 3x3) on the K <= 32 kernel (csrc/conv_narrow.hip) against the tuned tile +
split-K, timed as nodes of a captured graph (16 launches per replay).  python tools/narrow_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')]
import torch
from ppyolo_hip import ops


def timed(fn, reps=16, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        g.replay()
    en.record(); en.synchronize()
    return st.elapsed_time(en) / (reps * iters) * 1e3


def main():
    ws = torch.empty(64 << 20, device='cuda')
    for N, H, W, C, K, stride, base in ((8, 38, 38, 512, 27, 2, (53, 9)), (8, 19, 19, 512, 27, 1, (44, 9)), (4, 19, 19, 512, 27, 1, (44, 9)), (1, 19, 19, 512, 27, 1, (44, 9))):
        x = torch.randn(N, H, W, C, device='cuda')
        wk = torch.randn(K, 3, 3, C, device='cuda') * 0.02
        sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
        Ho = (H + 2 - 3) // stride + 1
        y = torch.empty(N, Ho, Ho, K, device='cuda')
        wf = ops.split_weights_f16x2(wk, sc)
        am_in, am_out = ops.amax_slots(x), ops.amax_slots(N=N, device='cuda')
        def run(c, s):
            return lambda: ops.conv2d_bn_act(ops.View(x), wk, sc, sh, ops.View(y), stride, 1, None, cfg=c, splitk=s, ws=ws, w_f16=wf, amax_in=am_in, amax_out=am_out)
        t0 = timed(run(*base))
        t1 = [timed(run(ops.narrow_cfg() + i, 1)) for i in range(7)]
        print('N%d %dx%d s%d: tile cfg %d split %d %.1f us   narrow %s us' % (N, H, W, stride, base[0], base[1], t0, ' '.join('%.1f' % t for t in t1)))


if __name__ == '__main__':
    main()
