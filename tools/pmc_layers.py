#!/usr/bin/env python
"""Per-LAYER memory traffic of the convolution launches against their algorithmic bytes (the review's "wasted traffic" item):
which launches move more than inputs + shortcut + weights read once and outputs written once, and by how much.

Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters need their own passes and process -- the guide's HBM section) of
bench.py's eager one-lane child; the dispatches of the convolution kernel families are matched, in order, to the plan's launching
conv / DCN ops (a split-K combine launch is added to the layer in front of it).  gfx950: FETCH_SIZE reports half of a 16 B/lane
streaming read, hence 2 x FETCH + WRITE (both KB).  The counters sit on the fabric side of the L2s: Infinity-Cache hits are
included, L2 hits are not.  usage (GPU box): tools/pmc_layers.py [--workload r50vd_608] [--batch 8] [--out profiles/rNN_pmc_layers.txt]"""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
MAIN_K = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_b2b', 'dcn_fused')


def dispatches(counter, argv_tail):
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    d = tempfile.mkdtemp(prefix='ppy_pmcl_')
    cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
           sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-child'] + argv_tail
    subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get('Counter_Name') == counter:
                    rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    shutil.rmtree(d, ignore_errors=True)
    rows.sort()
    # passes of the plan: from one stem launch to the next; keep the conv-family dispatches (+ combines) of each
    passes, cur = [], None
    for _, k, v in rows:
        if 'stem_conv' in k:
            cur = []
            passes.append(cur)
        elif cur is not None and (any(n in k for n in MAIN_K) or 'splitk_reduce' in k):
            cur.append((k, v))
    return passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='r50vd_608')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import torch
    import bench
    from ppyolo_hip import synth
    from ppyolo_hip.engine import tune_key
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device('cuda:0')
    model, _, _ = bench.build_model(wl['cfg'], dev)
    x = synth.synth_images(a.batch, wl['size'], seed=1234).to(dev)
    ex = model._plans.executor(x)
    layers = []
    for op in ex.plan.ops:
        if op['op'] not in ('conv', 'dcn') or op.get('b2b_of') is not None:
            continue
        K, R, S, C = op['w'].shape
        y = op['y']
        xin = op['x']
        M = y.N * y.H * y.W
        Min = xin.N * xin.H * xin.W
        rd = 4 * Min * xin.C + 2 * 2 * K * R * S * C          # activations once + both fp16 weight planes
        wr = 4 * M * K                                         # (an upsampling store writes the enlarged tensor: y IS that tensor)
        if op.get('res') is not None:
            rd += 4 * M * K
        if op['op'] == 'dcn':
            rd += 4 * M * op['om'].C
        if op.get('pool') is not None:
            wr += M * K
        if op.get('mpool') is not None:
            wr = 4 * (M // 4) * K
        key = tune_key(op)
        b = op.get('b2b')
        if b is not None:
            Kb = b['w'].shape[0]
            rd += 4 * M * Kb + 2 * 2 * Kb * K
            wr = 4 * M * Kb + (M * Kb if b.get('pool') is not None else 0)
            key += ' + ' + tune_key(b)
        layers.append(dict(key=key, cfg=op['cfg'], splitk=op['splitk'], read=rd, write=wr))
    del ex, model
    torch.cuda.empty_cache()
    tail = ['--workload', a.workload, '--batch', str(a.batch)]
    per = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        passes = [p for p in dispatches(counter, tail) if sum(1 for k, _ in p if 'splitk_reduce' not in k) == len(layers)]
        if not passes:
            raise SystemExit('no pass of the plan with %d convolution launches in the %s run' % (len(layers), counter))
        acc = [0.0] * len(layers)
        for p in passes:
            i = -1
            for k, v in p:
                if 'splitk_reduce' not in k:
                    i += 1
                acc[i] += v
        per[counter] = [v / len(passes) * 1024.0 for v in acc]
    lines = ['per-layer traffic (2 x FETCH_SIZE + WRITE_SIZE, fabric side of the L2s) against algorithmic bytes, %s batch %d' % (a.workload, a.batch),
             '%3s %-92s %4s %2s %9s %9s %6s %9s %9s %6s' % ('#', 'layer', 'cfg', 'sk', 'read MB', 'alg MB', 'x', 'write MB', 'alg MB', 'x')]
    tr = tw = ar = aw = 0.0
    worst = []
    for i, L in enumerate(layers):
        rd, wr = 2.0 * per['FETCH_SIZE'][i], per['WRITE_SIZE'][i]
        tr += rd; tw += wr; ar += L['read']; aw += L['write']
        lines.append('%3d %-92s %4d %2d %9.1f %9.1f %6.2f %9.1f %9.1f %6.2f' % (i, L['key'][:92], L['cfg'], L['splitk'], rd / 1e6, L['read'] / 1e6, rd / L['read'],
                                                                              wr / 1e6, L['write'] / 1e6, wr / max(L['write'], 1)))
        worst.append((rd + wr - L['read'] - L['write'], i, L['key']))
    lines.append('total: read %.2f GB (algorithmic %.2f, x %.2f), written %.2f GB (algorithmic %.2f, x %.2f), all %.2f GB vs %.2f GB' % (
        tr / 1e9, ar / 1e9, tr / ar, tw / 1e9, aw / 1e9, tw / aw, (tr + tw) / 1e9, (ar + aw) / 1e9))
    lines.append('largest excess (MB per launch): ' + '; '.join('#%d %+.0f' % (i, e / 1e6) for e, i, _ in sorted(worst, reverse=True)[:12]))
    text = '\n'.join(lines)
    print(text)
    if a.out:
        with open(a.out, 'w') as fh:
            fh.write(text + '\n')


if __name__ == '__main__':
    main()
