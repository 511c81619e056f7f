#!/bin/bash
# rocprofv3 kernel-trace summary of the training step (bench.py --train), per step.  usage: tools/train_prof.sh [out.txt] [steps]
OUT=${1:-/root/repo/gpurun_out/train_kernel_stats.txt}; STEPS=${2:-5}
mkdir -p "$(dirname "$OUT")"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tprof -o t -- python /root/repo/bench.py --train --steps $STEPS --warmup 2 > /tmp/tprof_bench.json 2>/dev/null
python - "$OUT" $((STEPS + 2)) <<'P'
import csv, glob, sys
out, nsteps = sys.argv[1], int(sys.argv[2])
f = glob.glob('/tmp/tprof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(out, 'w') as fh:
    fh.write('# rocprofv3 --kernel-trace --stats -- python bench.py --train --steps %d --warmup 2   (R50vd-608, 8 images, one MI355X; %d steps)\n' % (nsteps - 2, nsteps))
    fh.write('total kernel time per step: %.3f ms\n' % (tot / nsteps / 1e6))
    for r in rows[:40]:
        name = r['Name'].replace('(anonymous namespace)::', '')[:96]
        fh.write('%-96s calls/step %6.1f avg %8.1f us %7.2f ms/step %s %%\n' % (name, float(r['Calls']) / nsteps, float(r['AverageNs']) / 1e3,
                                                                              float(r['TotalDurationNs']) / nsteps / 1e6, r['Percentage']))
P
