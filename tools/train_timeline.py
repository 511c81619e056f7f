#!/usr/bin/env python
"""Busy / idle time of the TRAINING step from a rocprofv3 kernel trace of `bench.py --train` (one stream).

    train_timeline.py <dir with *kernel_trace.csv> <out.txt> [--skip 3] [--count 4]

A step is delimited by its first yolo_loss_kernel launch (three per step: one per head level); the window is `count` whole
steps after `skip`.  Reported per step: launches, sum of kernel durations, device busy time (union of the kernels' intervals),
idle time (no kernel resident) and how it splits over gap lengths -- the part a captured graph could remove."""
import argparse
import csv
import glob
import os


def short(name):
    """Kernel name without return type, namespaces, template and call arguments."""
    import re
    n = name[5:] if name.startswith('void ') else name
    n = n.replace('(anonymous namespace)::', '')
    n = re.split(r'[<(]', n)[0]
    return n.split('::')[-1][:40] or name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('out')
    ap.add_argument('--skip', type=int, default=3)
    ap.add_argument('--count', type=int, default=4)
    a = ap.parse_args()
    rows = []
    for path in glob.glob(os.path.join(a.dir, '**', '*kernel_trace.csv'), recursive=True):
        with open(path) as fh:
            rows += list(csv.DictReader(fh))
    ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
    loss = [s for s, e, n in ks if 'yolo_loss_kernel' in n]
    marks = loss[::3]
    t0, t1 = marks[a.skip], marks[a.skip + a.count]
    win = [(s, e, n) for s, e, n in ks if s >= t0 and s < t1]
    steps = float(a.count)
    ksum = sum(e - s for s, e, n in win)
    busy, gaps, ce, last = 0, [], None, ''
    for s, e, n in win:
        if ce is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            busy += ce - cs
            gaps.append((s - ce, short(last) + ' -> ' + short(n)))
            cs, ce = s, e
        last = n
    busy += ce - cs
    wall = t1 - t0
    lines = ['training step timeline: %d steps, %d launches per step' % (a.count, len(win) / steps),
             'wall %.1f us / step   sum of kernel durations %.1f   device busy %.1f   idle %.1f (%.1f %%)'
             % (wall / steps / 1e3, ksum / steps / 1e3, busy / steps / 1e3, (wall - busy) / steps / 1e3, 100.0 * (wall - busy) / wall)]
    for lo, hi in ((0, 2000), (2000, 5000), (5000, 10000), (10000, 20000), (20000, 50000), (50000, 10 ** 12)):
        g = [x for x, _ in gaps if lo <= x < hi]
        lines.append('gaps %6.0f - %-8s us: %5.1f per step, %7.1f us per step' % (lo / 1e3, '%.0f' % (hi / 1e3) if hi < 10 ** 12 else 'inf',
                                                                              len(g) / steps, sum(g) / steps / 1e3))
    big = sorted(gaps, reverse=True)[:12]
    lines.append('largest gaps (us, the kernel that follows): ' + '; '.join('%.0f %s' % (g / 1e3, n) for g, n in big))
    open(a.out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
