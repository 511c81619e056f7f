#!/usr/bin/env python
"""Per-workgroup timeline of one conv launch (debug hook ppy_debug_set_trace).
usage: conv_trace.py "N,H,W,C,K,R,stride[,res]" cfg splitk"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch
from ppyolo_hip import ops, _lib

shp = [int(v) for v in sys.argv[1].split(',')]
N, H, W, C, K, R, stride = shp[:7]
cfg, splitk = int(sys.argv[2]), int(sys.argv[3])
pad = (R - 1) // 2
Ho, Wo = ops.conv_out_hw(H, W, R, R, stride, pad)
x = torch.randn(N, H, W, C, device='cuda'); w = torch.randn(K, R, R, C, device='cuda') * 0.05
mode = os.environ.get('PPY_TRACE_DATA', 'randn')      # operand data changes the MFMA power draw -> the clock
if mode == 'zeros': x.zero_(); w.zero_()
if mode == 'ones': x.fill_(1.0); w.fill_(0.5)
if mode == 'relu': x.clamp_(min=0)
sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
y = torch.empty(N, Ho, Wo, K, device='cuda'); ws = torch.empty(64 << 20, device='cuda')
res = torch.randn(N, Ho, Wo, K, device='cuda') if len(shp) > 7 and shp[7] else None      # residual input (expand layers)
tr = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
L = _lib.lib()._handle
lib = ctypes.CDLL(_lib.LIB_PATH)
w3 = ops.split_weights_bf16x3(w); wf16 = ops.split_weights_f16x2(w, sc); amax0 = ops.amax_slots(x)
xs = None
if os.environ.get('PPY_TRACE_GP', '0') == '1':      # the input as a PRE-SPLIT tensor (what a linked producer stores): per pixel and 32-channel
    # group 32 fp16 first terms then 32 fp16 second terms of x * s_image -- the ':g' layers of the plan
    mx = x.abs().amax(dim=(1, 2, 3))
    xs = torch.pow(2.0, 13 - torch.floor(torch.log2(mx)))
    v = x * xs.view(-1, 1, 1, 1)
    h0 = v.half()
    h1 = (v - h0.float()).half()
    packed = torch.stack([h0.view(N, H, W, C // 32, 32), h1.view(N, H, W, C // 32, 32)], dim=4).contiguous()      # [N,H,W,C/32,2,32] fp16
    x = packed.view(torch.float32).view(N, H, W, C).contiguous()
def run():
    ops.conv2d_bn_act(ops.View(x), w, sc, sh, ops.View(y), stride, pad, 'relu', residual=None if res is None else ops.View(res), cfg=cfg,
                      splitk=splitk, ws=ws, w_x3=w3, w_f16=wf16, amax_in=amax0, x_split=xs)
for _ in range(int(os.environ.get('PPY_TRACE_WARM', '3'))): run()
torch.cuda.synchronize()
lib.ppy_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
run(); torch.cuda.synchronize()
lib.ppy_debug_set_trace(ctypes.c_void_p(0))
t = tr.cpu().view(-1, 4)
t = t[t[:, 1] != 0]
nb = t.shape[0]
t0 = int(t[:, 0].min())
start = (t[:, 0] - t0).double() / 100.0   # readcyclecounter ticks at 100 MHz -> us
end = (t[:, 1] - t0).double() / 100.0
print('blocks', nb, 'span %.1f us' % end.max().item(), 'mean block duration %.1f us' % (end - start).mean().item())
hw = t[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = (t[:, 2] >> 32) & 0xf
cuid = (xcc * 8 + se) * 32 + sh_ * 16 + cu
per = collections.defaultdict(list)
for i in range(nb): per[int(cuid[i])].append((start[i].item(), end[i].item()))
print('distinct CU ids', len(per), ' blocks per CU: min %d max %d' % (min(len(v) for v in per.values()), max(len(v) for v in per.values())))
# histogram of start times
import numpy as np
st = np.sort(start.numpy()); en = np.sort(end.numpy())
print('start pct 0/25/50/75/100:', np.percentile(st, [0, 25, 50, 75, 100]).round(1))
print('end   pct 0/25/50/75/100:', np.percentile(en, [0, 25, 50, 75, 100]).round(1))
dur = (end - start).numpy()
order = np.argsort(start.numpy())
q = len(order) // 4
for k in range(4):
    sel = order[k * q:(k + 1) * q]
    print('  start-quartile %d: mean duration %.1f us' % (k, dur[sel].mean()))
# active blocks over time
T = np.linspace(0, en.max(), 21)
act = [(int(((start.numpy() <= tt) & (end.numpy() > tt)).sum())) for tt in T]
print('active workgroups over time:', act)

ph = t[:, 3]
if int(ph.max()) > 0:      # bf16x3 kernels: shader-clock phase lengths
    a = (ph & 0x1fffff).double() * 16; m = ((ph >> 21) & 0x1fffff).double() * 16; e = ((ph >> 42) & 0x1fffff).double() * 16
    cyc = a + m + e
    us = (end - start)
    print('shader clock during the launch: %.0f MHz (mean over workgroups)' % (cyc / us).mean().item())
    print('phase cycles mean: setup %.0f  main loop %.0f  epilogue %.0f   (fractions %.1f%% / %.1f%% / %.1f%%)' % (
        a.mean(), m.mean(), e.mean(), 100 * (a / cyc).mean(), 100 * (m / cyc).mean(), 100 * (e / cyc).mean()))
    chunks = R * R * C // 32 // splitk
    print('main loop: %.0f cycles per 32-deep chunk' % (m.mean().item() / chunks))
