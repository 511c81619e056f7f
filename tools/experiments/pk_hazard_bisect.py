#!/usr/bin/env python
"""Bisection of the packed-fp32 hazard by pairing MINIMAL and REAL kernels on two streams:

  A. minimal victim (tools/probes/pk_hazard_min.hip: nothing but v_pk_{add,fma}_f32 on registers, self-checking) beside the
     REAL convolution kernels -- bf16x3 / f16x2 / exact fp32 -- and beside ablated builds of the 16-bit-MFMA kernel
     (csrc/conv_x3.hip PPY_X3_ABL: 1 = no DMA in the loop, 2 = no operand split, 5 = MFMA only, 7 = operand delivery only);
  B. the REAL victim (the decode kernel of a library built WITH packed ops, as tools/pk_hazard_probe.py) beside the minimal
     one-ingredient aggressors of pk_hazard_min.hip.

Prebuilt inputs (build container): tools/probes/bin/libpk_hazard_min.so, ppyolo_hip/lib/variants/{packed,abl1,abl2,abl5,abl7}.so.
Each variant library is loaded in its own process (one libppyolo_hip.so per process).  usage: pk_hazard_bisect.py [rounds]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pytorch-ppyolo_amd')
VAR = os.path.join(PKG, 'ppyolo_hip', 'lib', 'variants')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def minlib():
    L = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'bin', 'libpk_hazard_min.so'))
    L.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.pk_aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]
    L.pk_aggressor_name.restype = ctypes.c_char_p
    return L


def part_a(tag, rounds):
    """minimal victim beside the real conv kernels of the library selected by PPYOLO_HIP_LIB"""
    import torch
    from ppyolo_hip import ops
    M = minlib()
    bad = torch.zeros(32, device='cuda')
    N, H, W, C, K, R = 8, 76, 76, 256, 256, 3
    sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    yb = torch.empty(N, H, W, K, device='cuda')
    ws = torch.empty(64 << 20, device='cuda')
    xb = torch.randn(N, H, W, C, device='cuda')
    wb = torch.randn(K, R, R, C, device='cuda') * 0.05
    w3, wf, amax = ops.split_weights_bf16x3(wb), ops.split_weights_f16x2(wb, sc), ops.amax_slots(xb)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    kinds = [('bf16x3 128x128', 32), ('f16x2 128x128', 41), ('f16x2 128x128 3 stages', 50), ('exact fp32 (glds)', 19)]
    if tag != 'product':
        kinds = kinds[:2]
    for name, cfg in kinds:
        torch.cuda.synchronize()
        errs = torch.zeros(rounds, dtype=torch.int32, device='cuda')
        for s_ in (sa, sb):
            s_.wait_stream(torch.cuda.current_stream())
        for r in range(rounds):                     # everything queued back to back: sustained co-residency, as in the model
            with torch.cuda.stream(sb):
                for _ in range(2):
                    ops.conv2d_bn_act(ops.View(xb), wb, sc, sh, ops.View(yb), 1, 1, 'relu', cfg=cfg, splitk=1, ws=ws, w_x3=w3,
                                      w_f16=wf, amax_in=amax)
            with torch.cuda.stream(sa):
                for _ in range(4):
                    M.pk_victim_launch(1024, 2000, errs[r:].data_ptr(), bad.data_ptr(), sa.cuda_stream)
        torch.cuda.synchronize()
        e = errs.cpu()
        nbad, lanes = int((e > 0).sum()), int(e.sum())
        first = [round(v, 3) for v in bad[:4].tolist()] if nbad else None
        print('A | conv library %-8s | aggressor %-24s | minimal pk victim: %3d of %d launches wrong (%d lanes)%s'
              % (tag, name, nbad, rounds, lanes, ' first (dv.lo dv.hi w.lo w.hi): %s' % first if first else ''), flush=True)


def part_b(rounds):
    """real decode victim (packed library) beside the minimal aggressors"""
    import torch
    from conftest import build_model
    from config import PPYOLO_r18vd_Config
    from ppyolo_hip import synth
    M = minlib()
    cfg = PPYOLO_r18vd_Config()
    model, _ = build_model(cfg, 0, 'cuda')
    x, ims = synth.synth_images(4, 320, seed=300).cuda(), synth.synth_im_size(4).cuda()
    model(x, ims)
    ex = model._plans.executor(x)
    ex._run_decode()
    torch.cuda.synchronize()
    ref = ex.boxes.clone()
    src, sink = torch.zeros(1 << 19, device='cuda'), torch.zeros(1024, device='cuda')
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for kind in range(M.pk_num_aggressors()):
        for wgs, lds in ((512, 64 << 10), (256, 96 << 10)):
            hist = torch.empty((rounds,) + tuple(ref.shape), device='cuda')
            torch.cuda.synchronize()
            for s_ in (sa, sb):
                s_.wait_stream(torch.cuda.current_stream())
            for r in range(rounds):                 # queued back to back, no host sync in between
                with torch.cuda.stream(sb):
                    M.pk_aggressor_launch(kind, wgs, lds, 1500, src.data_ptr(), sink.data_ptr(), sb.cuda_stream)
                with torch.cuda.stream(sa):
                    ex.boxes.zero_()
                    ex._run_decode()
                    hist[r].copy_(ex.boxes)
            torch.cuda.synchronize()
            nbad = sum(int(not torch.equal(hist[r], ref)) for r in range(rounds))
            print('B | decode of the PACKED library | aggressor %-52s lds %3d KB | %3d of %d decodes wrong'
                  % (M.pk_aggressor_name(kind).decode(), lds >> 10, nbad, rounds), flush=True)


def part_c(rounds):
    """foreign victims: PyTorch's own fp32 elementwise / reduction kernels (hipcc -O3 code with packed ops, not ours) beside
    the real convolution kernels and the minimal 16x16x32 chain -- the advisor's question whether InFlight endangers
    OTHER people's kernels on other streams."""
    import torch
    from ppyolo_hip import ops
    M = minlib()
    g = torch.Generator(device='cuda').manual_seed(1)
    a, b, c = (torch.randn(1 << 20, device='cuda', generator=g) for _ in range(3))
    m1, m2 = torch.randn(512, 512, device='cuda', generator=g), torch.randn(512, 512, device='cuda', generator=g)
    victims = {'addcmul + mul + add (fp32 elementwise)': lambda: torch.addcmul(a, b, c) * 1.5 + b,
               'sigmoid * exp (fp32 elementwise)': lambda: torch.sigmoid(a) * torch.exp(b * 0.1),
               'pow(x, 0.6) * pow(y, 0.4)': lambda: torch.pow(a.abs() + 0.1, 0.6) * torch.pow(b.abs() + 0.1, 0.4),
               'softmax + sum (reductions)': lambda: torch.softmax(m1, 1).sum(0),
               'fp32 matmul 512^3 (rocBLAS / hipBLASLt)': lambda: m1 @ m2}
    N, H, W, C, K, R = 8, 76, 76, 256, 256, 3
    sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    yb, ws = torch.empty(N, H, W, K, device='cuda'), torch.empty(64 << 20, device='cuda')
    xb, wb = torch.randn(N, H, W, C, device='cuda'), torch.randn(K, R, R, C, device='cuda') * 0.05
    w3, wf, amax = ops.split_weights_bf16x3(wb), ops.split_weights_f16x2(wb, sc), ops.amax_slots(xb)
    src, sink = torch.zeros(1 << 19, device='cuda'), torch.zeros(1024, device='cuda')
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def conv(cfg):
        return lambda: ops.conv2d_bn_act(ops.View(xb), wb, sc, sh, ops.View(yb), 1, 1, 'relu', cfg=cfg, splitk=1, ws=ws, w_x3=w3,
                                         w_f16=wf, amax_in=amax)
    aggr = {'conv bf16x3 128x128': conv(32), 'conv f16x2 128x128': conv(41),
            'minimal 16x16x32_bf16 chain, 2 WG/CU': lambda: M.pk_aggressor_launch(6, 512, 64 << 10, 1500, src.data_ptr(), sink.data_ptr(),
                                                                                 torch.cuda.current_stream().cuda_stream)}
    for vname, vf in victims.items():
        ref = vf().clone()
        for aname, af in aggr.items():
            outs = []
            torch.cuda.synchronize()
            for s_ in (sa, sb):
                s_.wait_stream(torch.cuda.current_stream())
            for r in range(rounds):
                with torch.cuda.stream(sb):
                    af()
                    af()
                with torch.cuda.stream(sa):
                    outs.append(vf())
            torch.cuda.synchronize()
            nbad = sum(int(not torch.equal(o, ref)) for o in outs)
            print('C | torch victim %-42s | aggressor %-38s | %3d of %d results differ from the solo result'
                  % (vname, aname, nbad, rounds), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        if sys.argv[2] == 'B':
            part_b(int(sys.argv[3]))
        elif sys.argv[2] == 'C':
            part_c(int(sys.argv[3]))
        else:
            part_a(sys.argv[2], int(sys.argv[3]))
    else:
        rounds = sys.argv[1] if len(sys.argv) > 1 else '60'
        for tag in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('product', 'abl1', 'abl2', 'abl5', 'abl7', 'B', 'C')):
            env = dict(os.environ)
            if tag == 'B':
                env['PPYOLO_HIP_LIB'] = os.path.join(VAR, 'packed.so')
            elif tag not in ('product', 'C'):
                env['PPYOLO_HIP_LIB'] = os.path.join(VAR, tag + '.so')
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tag, rounds], env=env)
