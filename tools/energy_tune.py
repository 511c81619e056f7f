#!/usr/bin/env python
"""Second tuning stage by ENERGY per launch (round 5).

With two batches in flight the board sits at its power limit and images/s = watts / (joules per image)
(profiles/r05_cu_mask_ab.txt, profiles/r05_energy_layers.txt): the tile configuration a layer wants is the one with the smallest
duration x power, not the smallest duration.  HipExecutor.autotune() ranks all configurations by time; this tool takes each layer's
front-runners (within `--window` of the fastest), replays every one back to back for `--seconds` with the socket power sampled
(tools/power_meter.py) and keeps the configuration with the fewest millijoules -- unless the saving over the fastest one is inside
the measurement noise (`--min-gain`).  Layers with the same table key are measured once.

    python tools/energy_tune.py --out gpurun_out/r05/tuned_energy.json [--workload r50vd_608] [--batch 8] [--topk 4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
from power_meter import PowerMeter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='r50vd_608')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seconds', type=float, default=1.2)
    ap.add_argument('--topk', type=int, default=4)
    ap.add_argument('--window', type=float, default=1.12, help='candidates: the front-runners within this factor of the fastest')
    ap.add_argument('--min-gain', type=float, default=0.015, help='keep the fastest unless another saves at least this share of its energy')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r05', 'tuned_energy.json'))
    ap.add_argument('--report', default=None)
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
    t0 = time.time()
    ex.autotune(iters=5)
    print('autotune: %.0f s' % (time.time() - t0), flush=True)
    ex._unlink_splits()
    pw = PowerMeter()
    print('power source:', pw.source, flush=True)

    def energy(op):
        ex._run_op(op)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(16):
                ex._run_op(op)
        g.replay()
        torch.cuda.synchronize()
        pw.start()
        t_start = time.perf_counter()
        n, t_meas, n_meas = 0, None, 0
        while True:
            for _ in range(8):
                g.replay()
            n += 8
            torch.cuda.synchronize()
            now = time.perf_counter()
            if t_meas is None and now - t_start >= 0.35 * a.seconds:
                t_meas, n_meas = now, n
            if now - t_start >= a.seconds:
                break
        t_end = time.perf_counter()
        w, f, ns = pw.stop(t_meas)
        g.reset()
        us = (t_end - t_meas) / max(1, (n - n_meas) * 16) * 1e6
        return us, w, f, ns

    table, table_time, report, seen = {}, {}, [], {}
    gp_scales = torch.ones(a.batch, dtype=torch.float32, device=dev)
    for op in ex.plan.ops:
        if op['op'] not in ('conv', 'dcn') or not op.get('_front'):
            continue
        key = tune_key(op)
        if key in seen:
            op['cfg'], op['splitk'] = seen[key]
            continue
        front = [c for c in op['_front'] if c[0] <= a.window * op['_front'][0][0]][:a.topk]
        rows = []
        for ms, c, s in front:
            op['cfg'], op['splitk'] = c, s
            op.pop('x_split', None)
            if op.get('gp_in') and s <= 1 and ex._split_capable(c, True):
                op['x_split'] = gp_scales
            us, w, f, ns = energy(op)
            rows.append(dict(cfg=c, splitk=s, tune_ms=round(ms, 4), us=round(us, 2), watts=round(w or -1, 1), sclk=round(f or -1), mj=round((w or 0) * us * 1e-3, 3)))
        op.pop('x_split', None)
        fastest = min(rows, key=lambda r: r['us'])
        best = min(rows, key=lambda r: r['mj'])
        if best is not fastest and best['mj'] > (1.0 - a.min_gain) * fastest['mj']:
            best = fastest
        op['cfg'], op['splitk'] = best['cfg'], best['splitk']
        seen[key] = (best['cfg'], best['splitk'])
        table[key] = [best['cfg'], best['splitk'], round(best['us'] * 1e-3, 4)]
        table_time[key] = [fastest['cfg'], fastest['splitk'], round(fastest['us'] * 1e-3, 4)]
        report.append(dict(key=key, chosen=best, fastest=fastest, candidates=rows))
        print('%-48s fastest cfg %3d/%d %7.1f us %6.0f W %7.2f mJ | chosen cfg %3d/%d %7.1f us %6.0f W %7.2f mJ%s' % (
            key, fastest['cfg'], fastest['splitk'], fastest['us'], fastest['watts'], fastest['mj'], best['cfg'], best['splitk'], best['us'],
            best['watts'], best['mj'], '  <-- changed' if best is not fastest else ''), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(table, fh, indent=0, sort_keys=True)
    with open(a.out.replace('.json', '_time_only.json'), 'w') as fh:      # the same fresh measurement, fastest configuration per layer (A/B partner)
        json.dump(table_time, fh, indent=0, sort_keys=True)
    with open(a.report or a.out.replace('.json', '_report.json'), 'w') as fh:
        json.dump(report, fh, indent=0)
    ch = [r for r in report if r['chosen'] is not r['fastest'] and (r['chosen']['cfg'], r['chosen']['splitk']) != (r['fastest']['cfg'], r['fastest']['splitk'])]
    print('%d of %d layer shapes changed; energy of the changed shapes %.1f -> %.1f mJ per launch set' % (
        len(ch), len(report), sum(r['fastest']['mj'] for r in ch), sum(r['chosen']['mj'] for r in ch)))


if __name__ == '__main__':
    main()
