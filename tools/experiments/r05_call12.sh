#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
 for pr in "" "-1,0" "0,-1"; do
  tag=$(echo "p$pr" | tr -c 'a-z0-9\n' '_')
  PPYOLO_HIP_LANE_PRIORITY="$pr" timeout 300 $B > $O/prio_${tag}_$rep.json 2> $O/prio_${tag}_$rep.err
 done
 PPYOLO_HIP_LANE_PRIORITY="-1,0,0" timeout 300 $B --in-flight 3 > $O/prio_d3_$rep.json 2> $O/prio_d3_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/prio_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-40s value %8.1f sustained %8.1f' % (f, d['value'], d['sustained']['value']))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
