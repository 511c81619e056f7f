#!/bin/bash
# which kernels sit around the ~73 __amd_rocclr_copyBuffer dispatches of a training step? (round 5)
O=gpurun_out/r05; mkdir -p $O
SCR=/tmp/copy_nb; rm -rf $SCR; mkdir -p $SCR
REPO=$PWD
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $SCR -o t -- python $REPO/bench.py --train --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline > $REPO/$O/copy_nb.log 2>&1 )
python - $SCR <<'PY' | tee $O/copy_neighbours.txt
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r):
    n = r['Kernel_Name']
    return n.replace('(anonymous namespace)::', '')[:70]
c = collections.Counter()
for i, r in enumerate(rows):
    if 'copyBuffer' in r['Kernel_Name']:
        g = (r.get('Grid_Size') or r.get('Grid_Size_X') or '?')
        c[(nm(rows[i - 1]) if i else '-', g, nm(rows[i + 1]) if i + 1 < len(rows) else '-')] += 1
for (p, g, n), v in c.most_common(25):
    print('%4d  grid %-8s after %-62s before %s' % (v, g, p, n))
PY
