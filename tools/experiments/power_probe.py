#!/usr/bin/env python
"""Socket power and shader clock (rocm-smi) while one conv configuration runs back to back for a few seconds.
usage: power_probe.py "N,H,W,C,K,R,stride" cfg [seconds] [randn|zeros]"""
import os, subprocess, sys, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch
from ppyolo_hip import ops

shp = [int(v) for v in sys.argv[1].split(',')]
N, H, W, C, K, R, stride = shp[:7]
cfg = int(sys.argv[2]); secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
mode = sys.argv[4] if len(sys.argv) > 4 else 'randn'
pad = (R - 1) // 2
Ho, Wo = ops.conv_out_hw(H, W, R, R, stride, pad)
x = torch.randn(N, H, W, C, device='cuda'); w = torch.randn(K, R, R, C, device='cuda') * 0.05
if mode == 'zeros': x.zero_(); w.zero_()
sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
y = torch.empty(N, Ho, Wo, K, device='cuda'); ws = torch.empty(64 << 20, device='cuda'); w3 = ops.split_weights_bf16x3(w); wf16 = ops.split_weights_f16x2(w, sc); amax0 = ops.amax_slots(x)
samples = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r'Power \(W\): ([0-9.]+)', o); ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
            samples.append((time.time(), float(pw.group(1)) if pw else -1, int(ck.group(1)) if ck else -1))
        except Exception as e:
            samples.append((time.time(), -1, -1))
        time.sleep(0.3)
th = threading.Thread(target=poll); th.start()
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for _ in range(200):
        ops.conv2d_bn_act(ops.View(x), w, sc, sh, ops.View(y), stride, pad, 'relu', cfg=cfg, splitk=1, ws=ws, w_x3=w3, w_f16=wf16, amax_in=amax0)
    torch.cuda.synchronize(); it += 200
dt = time.time() - t0
stop = True; th.join()
flops = 2.0 * N * Ho * Wo * K * R * R * C
print('cfg %d %s: %.1f TFLOP/s sustained over %.1f s' % (cfg, mode, flops * it / dt / 1e12, dt))
print('power (W) / sclk (MHz) samples:', [(int(p), c) for _, p, c in samples])
