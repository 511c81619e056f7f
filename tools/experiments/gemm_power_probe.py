#!/usr/bin/env python
"""Reference point for the energy argument of DESIGN.md 4.1: socket power / shader clock / TFLOP/s of the vendor
library's dense 16-bit GEMM (torch.matmul -> hipBLASLt) on random operands, sampled like tools/power_probe.py.
usage: gemm_power_probe.py [fp16|bf16] [n] [seconds]"""
import re
import subprocess
import sys
import threading
import time

import torch

dt_name = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
dt = torch.float16 if dt_name == 'fp16' else torch.bfloat16
A = torch.randn(n, n, device='cuda', dtype=dt)
B = torch.randn(n, n, device='cuda', dtype=dt)
C = torch.empty(n, n, device='cuda', dtype=dt)
samples, stop = [], False


def poll():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r'Power \(W\): ([0-9.]+)', o)
            ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
            samples.append((float(pw.group(1)) if pw else -1, int(ck.group(1)) if ck else -1))
        except Exception:
            samples.append((-1, -1))
        time.sleep(0.3)


for _ in range(5):
    torch.matmul(A, B, out=C)
torch.cuda.synchronize()
th = threading.Thread(target=poll)
th.start()
t0, it = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(20):
        torch.matmul(A, B, out=C)
    torch.cuda.synchronize()
    it += 20
el = time.time() - t0
stop = True
th.join()
print('hipBLASLt %s GEMM %d^3 randn: %.1f TFLOP/s sustained over %.1f s' % (dt_name, n, 2.0 * n ** 3 * it / el / 1e12, el))
print('power (W) / sclk (MHz) samples:', [(int(p), c) for p, c in samples])
