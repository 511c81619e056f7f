#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
run() { PPYOLO_HIP_LANE_PRIORITY="$3" timeout 300 $B --in-flight $2 > $O/prio2_$1.json 2> $O/prio2_$1.err; }
for rep in 1 2 3; do
  run d2_base_$rep 2 ""
  run d3_hi00_$rep 3 "-1,0,0"
  run d3_base_$rep 3 ""
  run d4_hi000_$rep 4 "-1,0,0,0"
  run d3_hihi0_$rep 3 "-1,-1,0"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/prio2_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-40s value %8.1f sustained %8.1f' % (f, d['value'], d['sustained']['value']))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
