#!/bin/bash
# Alternating A/B of one environment switch on ONE box: tools/ab_env.sh VAR "v0 v1" [rounds] [bench args] -> value / one lane / traffic per run
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-alt-math --no-host-input --no-worst-case "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$VAR=$v  two lanes %.1f img/s  one lane %s  conv solo %.1f TFLOP/s (frac %.4f)  traffic %s GB/step' % (d['value'], d.get('one_batch_at_a_time',{}).get('value'), r['achieved'], r['frac'], round((r.get('traffic_source') or {}).get('hbm_bytes_per_step',0)/1e9,3)))"
  done
done
