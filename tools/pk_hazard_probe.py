#!/usr/bin/env python
"""Reproducer for the packed-fp32 corruption that made this library drop v_pk_*_f32 (ppyolo_hip/build.py).

A victim stream runs the decode kernel over fixed head outputs again and again; an aggressor stream runs one kind of
load beside it.  Every decode result must be bit-identical to the one computed alone.

    tools/pk_hazard_probe.py [--packed] [--iters 400] [loads ...]

--packed   use a library variant built WITH packed fp32 ops (built into ppyolo_hip/lib/variants/packed.so on first use)
loads      none | bf16x3 | f16x2 | fp32 | gemm (torch bf16 matmul) ; a trailing 'z' = all-zero operands (low power)

Measured (profiles/r01_packed_fp32_hazard.txt): with packed ops, bf16x3 / f16x2 loads corrupt 80-95 % / 15-20 % of the
decodes (also with zero operands, so not a power effect); fp32-MFMA, hipBLASLt and no load: none.  Without packed
ops: none under any load."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pytorch-ppyolo_amd')
sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--packed', action='store_true')
    ap.add_argument('--iters', type=int, default=400)
    ap.add_argument('loads', nargs='*', default=['none', 'bf16x3', 'bf16x3z', 'f16x2', 'fp32', 'gemm'])
    a = ap.parse_args()
    if a.packed:
        var = os.path.join(PKG, 'ppyolo_hip', 'lib', 'variants', 'packed.so')
        if not os.path.exists(var):
            os.makedirs(os.path.dirname(var), exist_ok=True)
            os.environ['PPY_ALLOW_PACKED_FP32'] = '1'
            from ppyolo_hip import build
            build.build(verbose=False, out=var)
        os.environ['PPYOLO_HIP_LIB'] = var
    import torch
    from conftest import build_model
    from config import PPYOLO_r18vd_Config
    from ppyolo_hip import ops, synth

    cfg = PPYOLO_r18vd_Config()
    model, _ = build_model(cfg, 0, 'cuda')
    x, ims = synth.synth_images(4, 320, seed=300).cuda(), synth.synth_im_size(4).cuda()
    model(x, ims)
    ex = model._plans.executor(x)
    ex._run_decode()
    torch.cuda.synchronize()
    ref = ex.boxes.clone()
    hist = torch.empty((a.iters,) + tuple(ref.shape), device='cuda')

    N, H, W, C, K, R = 8, 76, 76, 256, 256, 3
    sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    yb = torch.empty(N, H, W, K, device='cuda')
    ws = torch.empty(64 << 20, device='cuda')
    A = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    B = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    Cm = torch.empty(8192, 8192, device='cuda', dtype=torch.bfloat16)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for load in a.loads:
        zero, kind = load.endswith('z'), load.rstrip('z')
        xb = torch.randn(N, H, W, C, device='cuda') * (0.0 if zero else 1.0)
        wb = torch.randn(K, R, R, C, device='cuda') * (0.0 if zero else 0.05)
        w3, wf, amax = ops.split_weights_bf16x3(wb), ops.split_weights_f16x2(wb, sc), ops.amax_slots(xb)

        def aggressor():
            if kind in ('bf16x3', 'f16x2'):         # tile 128x128a of the scheme
                ops.conv2d_bn_act(ops.View(xb), wb, sc, sh, ops.View(yb), 1, 1, 'relu', cfg=32 if kind == 'bf16x3' else 41,
                                  splitk=1, ws=ws, w_x3=w3, w_f16=wf, amax_in=amax)
            elif kind == 'fp32':
                ops.conv2d_bn_act(ops.View(xb), wb, sc, sh, ops.View(yb), 1, 1, 'relu', cfg=-1, splitk=1, ws=ws)
            elif kind == 'gemm':
                torch.matmul(A, B, out=Cm)
        torch.cuda.synchronize()
        for s in (sa, sb):
            s.wait_stream(torch.cuda.current_stream())
        for i in range(a.iters):
            with torch.cuda.stream(sb):
                aggressor()
                aggressor()
            with torch.cuda.stream(sa):
                ex.boxes.zero_()
                ex._run_decode()
                hist[i].copy_(ex.boxes)
        torch.cuda.synchronize()
        bad = [int((hist[i] != ref).sum()) for i in range(a.iters) if not torch.equal(hist[i], ref)]
        print('library %s | load %-8s | %d decodes, %d differ from the solo result%s' % (
            'WITH packed fp32 ops' if a.packed else 'without packed fp32 ops (product)', load, a.iters, len(bad),
            (' (wrong floats per bad decode: %s ...)' % bad[:6]) if bad else ''), flush=True)


if __name__ == '__main__':
    main()
