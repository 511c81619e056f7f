#!/usr/bin/env python
"""Energy per launch of every kernel of the inference plan (round 5).

The two-lane headline sits at the board's power cap (bench.py --power-trace: ~1350 W, shader clock managed down to ~2.0 GHz) and the
energy per batch is the same with one lane, two lanes and CU-masked lanes (profiles/r05_cu_mask_ab.txt), so at the cap
images/s = watts / (joules per image): the currency of the step is JOULES, and this tool prices every launch of the plan in them.
Each op is captured into a hipGraph of `burst` launches and replayed back to back for `--seconds` while socket power and shader
clock are sampled (tools/power_meter.py: amdsmi in-process, else rocm-smi); J per launch = mean power x mean duration.
Back-to-back replay of ONE layer is not the layer inside the step (caches are warmer, the governor settles on this layer's own
clock), so the column to read is the ranking and the watts, and the sum over the plan is printed beside the step's measured energy.

    python tools/energy_layers.py [--workload r50vd_608] [--batch 8] [--seconds 1.2] [--out gpurun_out/r05/energy_layers.json]
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


from power_meter import PowerMeter as Power  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='r50vd_608')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seconds', type=float, default=1.5)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r05', 'energy_layers.json'))
    a = ap.parse_args()
    import bench
    import __graft_entry__ as ge
    ge.build()
    from ppyolo_hip import synth
    from ppyolo_hip.engine import tune_key
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS[a.workload]
    model, sd, cfg = bench.build_model(wl['cfg'], dev)
    x = synth.synth_images(a.batch, wl['size'], seed=1234).to(dev)
    ims = synth.synth_im_size(a.batch).to(dev)
    ex = model._plans.executor(x)
    ex.set_inputs(x, ims)
    ex.use_graph = False
    ex.run()
    torch.cuda.synchronize()
    total_flops, per_op = bench.conv_flops(ex.plan)
    pw = Power()
    print('power source:', pw.source, flush=True)
    idle = pw.read()

    def measure(fn, burst):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(burst):
                fn()
        g.replay()
        torch.cuda.synchronize()
        pw.start()
        t0 = time.perf_counter()
        n = 0
        t_meas, n_meas = None, 0
        while True:
            for _ in range(8):
                g.replay()
            n += 8
            torch.cuda.synchronize()
            now = time.perf_counter()
            if t_meas is None and now - t0 >= 0.35 * a.seconds:      # let the governor and the power average settle first
                t_meas, n_meas = now, n
            if now - t0 >= a.seconds:
                break
        t1 = time.perf_counter()
        w, f, ns = pw.stop(t_meas)
        us = (t1 - t_meas) / max(1, (n - n_meas) * burst) * 1e6
        g.reset()
        return us, w, f, ns

    rows = []
    for i, op in enumerate(ex.plan.ops):
        if op.get('owner') is not None:
            continue
        key = tune_key(op) if op['op'] in ('conv', 'dcn') else op['op']
        us, w, f, ns = measure(lambda: ex._run_op(op), 16)
        rows.append(dict(i=i, key=key, cfg=op.get('cfg'), splitk=op.get('splitk'), us=round(us, 2), watts=None if w is None else round(w, 1),
                         sclk_mhz=None if f is None else round(f), samples=ns, gflop=round(per_op[i] / 1e9, 3),
                         mj=None if w is None else round(w * us * 1e-3, 3)))
        print('%3d %-46s %8.1f us %7.1f W %5.0f MHz %8.3f mJ  %6.1f TF' % (i, key, us, w or -1, f or -1, (w or 0) * us * 1e-3,
                                                                          per_op[i] / max(us, 1e-9) / 1e6), flush=True)
    us, w, f, ns = measure(ex._run_decode, 4)
    rows.append(dict(i=-1, key='decode+nms', us=round(us, 2), watts=None if w is None else round(w, 1), sclk_mhz=None if f is None else round(f),
                     samples=ns, gflop=0.0, mj=None if w is None else round(w * us * 1e-3, 3)))
    # the whole step, one lane (graph replay), the same way
    us, w, f, ns = measure(ex._launch_all, 1)
    step = dict(us=round(us, 1), watts=None if w is None else round(w, 1), sclk_mhz=None if f is None else round(f),
                mj=None if w is None else round(w * us * 1e-3, 1))
    tot_mj = sum(r['mj'] or 0 for r in rows)
    tot_us = sum(r['us'] for r in rows)
    print('sum over the launches: %.1f us, %.1f mJ; whole step, one lane: %.1f us at %.0f W = %.1f mJ; idle reading %s' % (
        tot_us, tot_mj, us, w or -1, (w or 0) * us * 1e-3, idle))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(dict(power_source=pw.source, idle=idle, rows=rows, step_one_lane=step, sum_mj=round(tot_mj, 1), sum_us=round(tot_us, 1)), fh, indent=0)
    # by class
    cls = {}
    for r in rows:
        k = r['key']
        c = 'other'
        if k.startswith('dcnf'):
            c = 'dcn'
        elif k.startswith('conv'):
            c = '3x3' if ':R3:' in k else '1x1'
        elif k == 'decode+nms':
            c = 'decode+nms'
        d = cls.setdefault(c, [0.0, 0.0, 0.0])
        d[0] += r['us']; d[1] += r['mj'] or 0; d[2] += r['gflop']
    for c, d in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        print('%-12s %8.1f us %8.1f mJ (%4.1f %%) mean %6.0f W  %7.1f GFLOP -> %6.2f pJ/FLOP' % (c, d[0], d[1], 100 * d[1] / max(tot_mj, 1e-9),
                                                                                                 d[1] / max(d[0], 1e-9) * 1e3, d[2],
                                                                                                 d[1] * 1e-3 / max(d[2] * 1e9, 1) * 1e12))


if __name__ == '__main__':
    main()
