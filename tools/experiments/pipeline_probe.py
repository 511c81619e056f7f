#!/usr/bin/env python
"""Probe: D executors (D batches in flight, one hipGraph each) replayed round-robin on D streams
vs one executor.  usage: pipeline_probe.py [workload] [depth] [batch]"""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from ppyolo_hip import synth  # noqa: E402

faulthandler.enable()
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'r50vd_608']
D = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda')
streams = [torch.cuda.Stream() for _ in range(D)]
exs = []
for k in range(D):
    model, sd, cfg = bench.build_model(wl['cfg'], dev)
    x = synth.synth_images(B, wl['size'], seed=1234 + k).to(dev)
    ims = synth.synth_im_size(B).to(dev)
    ex = model._plans.executor(x)
    ex.set_inputs(x, ims)
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[k]):          # capture and replay on the stream it will live on
        for _ in range(3):
            ex.run()
    torch.cuda.synchronize()
    exs.append((model, ex))


def timed(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def single(steps):
    with torch.cuda.stream(streams[0]):
        for _ in range(steps):
            exs[0][1].run()


OFFSET = float(os.environ.get('PROBE_OFFSET', '0'))       # start lane k with a delay of k * OFFSET steps


def multi(steps):
    for i in range(steps):
        with torch.cuda.stream(streams[i % D]):
            if OFFSET > 0 and 0 < i < D:
                torch.cuda._sleep(int(OFFSET * i * STEP_S * 2.4e9))
            exs[i % D][1].run()


K = 60
STEP_S = timed(single, 10) / 10
timed(single, 10)
timed(multi, 10)
t1 = timed(single, K)
t2 = timed(multi, K)
print('offset %.2f ' % OFFSET + '%s bs%d streams=%s: one in flight %.1f img/s   %d in flight %.1f img/s  (%+.1f%%)' % (
    sys.argv[1] if len(sys.argv) > 1 else 'r50vd_608', B, os.environ.get('PPYOLO_HIP_STREAMS', 'dflt'), B * K / t1, D,
    B * K / t2, 100 * (t1 / t2 - 1)))
