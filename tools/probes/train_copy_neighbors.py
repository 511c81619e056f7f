#!/usr/bin/env python
"""Which kernels run right before / after the __amd_rocclr_copyBuffer launches of a training step?  Reads the kernel trace CSV
of `rocprofv3 --kernel-trace` (directory argument) and prints the most frequent (previous, next) neighbours."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name']))
rows.sort()
short = lambda n: n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:60]
c = collections.Counter()
for i, (t, n) in enumerate(rows):
    if 'copyBuffer' in n:
        c[(short(rows[i - 1][1]) if i else '-', short(rows[i + 1][1]) if i + 1 < len(rows) else '-')] += 1
for (a, b), n in c.most_common(25):
    print('%5d  %-60s -> copy -> %s' % (n, a, b))
