#!/bin/bash
# lane phase sweep: do two lanes in lockstep waste complementary phases?
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
 for ph in 0 0.25 0.5 0.75; do
  timeout 300 $B --lane-phase $ph > $O/phase_${ph}_$rep.json 2> $O/phase_${ph}_$rep.err
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/phase_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-40s value %8.1f sustained %8.1f' % (f, d['value'], d['sustained']['value']))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
