#!/usr/bin/env python
"""Per-stream busy time and overlap of the TWO-LANE timed loop of bench.py from a rocprofv3 kernel trace
(tools/prof_two_lane.sh -> profiles/<tag>_two_lane_timeline.txt).

    two_lane_timeline.py <dir with *kernel_trace.csv> <out.txt> [--skip 40] [--count 160]

A step ends with its lane's nms_finish_kernel; the window is the span between the end of finish number `skip` and the end of
finish number `skip + count` (ordered by end time), i.e. `count` whole steps in the middle of the timed loop, both lanes busy.
Reported per step: sum of kernel durations (all / convolution / other), busy time of each hardware queue (union of its
kernels' intervals), busy time of the device (union over both queues), the overlap (time in which kernels of both queues are
resident), idle time (no kernel resident), and the wall time -- the number `value` is made of."""
import argparse
import csv
import glob
import os
from collections import defaultdict

CONV = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_b2b', 'dcn_fused', 'splitk_reduce')


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('out')
    ap.add_argument('--skip', type=int, default=40)
    ap.add_argument('--count', type=int, default=160)
    a = ap.parse_args()
    rows = []
    for path in glob.glob(os.path.join(a.dir, '**', '*kernel_trace.csv'), recursive=True):
        with open(path) as fh:
            rows += list(csv.DictReader(fh))
    ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows]
    fin = sorted(e for s, e, q, n in ks if 'nms_finish_kernel' in n)
    if len(fin) < a.skip + a.count + 1:
        raise SystemExit('only %d steps in the trace' % len(fin))
    t0, t1 = fin[a.skip], fin[a.skip + a.count]
    win = [(max(s, t0), min(e, t1), q, n) for s, e, q, n in ks if e > t0 and s < t1]
    per_q = defaultdict(list)
    for s, e, q, n in win:
        per_q[q].append((s, e))
    steps = float(a.count)
    us = lambda ns: ns / steps / 1e3
    wall = t1 - t0
    busy_q = {q: union(iv) for q, iv in per_q.items()}
    busy = union([iv for v in per_q.values() for iv in v])
    ksum = sum(e - s for s, e, q, n in win)
    kconv = sum(e - s for s, e, q, n in win if any(c in n for c in CONV))
    lines = ['two-lane timed loop, %d steps (finish %d .. %d of %d in the trace), all figures per step' % (a.count, a.skip, a.skip + a.count, len(fin)),
             'wall time                          %9.1f us   (= ms_per_step of bench.py)' % us(wall),
             'sum of kernel durations            %9.1f us   (convolution / DCN launches %.1f, everything else %.1f)' % (us(ksum), us(kconv), us(ksum - kconv))]
    for q in sorted(busy_q, key=lambda q: -busy_q[q]):
        lines.append('queue %-4s busy                    %9.1f us   (%.1f %% of the wall time; %d kernels per step)' % (
            q, us(busy_q[q]), 100.0 * busy_q[q] / wall, round(len(per_q[q]) / steps)))
    both = sum(busy_q.values()) - busy
    lines += ['device busy (union of the queues)  %9.1f us   (%.1f %% of the wall time)' % (us(busy), 100.0 * busy / wall),
              'both queues resident (overlap)     %9.1f us   (%.1f %% of the wall time; %.1f %% of the summed queue-busy time)' % (
                  us(both), 100.0 * both / wall, 100.0 * both / max(1, sum(busy_q.values()))),
              'no kernel resident                 %9.1f us   (%.1f %% of the wall time)' % (us(wall - busy), 100.0 * (wall - busy) / wall),
              '(kernel durations inside the window are stretched by sharing the CUs with the other lane: the same kernels solo sum to',
              ' conv_kernel_ms_per_step_solo of the bench line; overlap fraction of the line = 1 - two-lane / one-lane step time)']
    tot = defaultdict(lambda: [0, 0])
    for s, e, q, n in win:
        k = n.replace('(anonymous namespace)::', '')[:70]
        tot[k][0] += e - s
        tot[k][1] += 1
    lines.append('')
    lines.append('%-72s %8s %10s %9s' % ('kernel (in the window, both lanes)', 'per step', 'us / step', 'avg us'))
    for k, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
        lines.append('%-72s %8.1f %10.1f %9.2f' % (k, c / steps, us(ns), ns / c / 1e3))
    with open(a.out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
