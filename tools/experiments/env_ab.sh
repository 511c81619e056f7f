#!/bin/bash
# Alternating A/B of one environment switch (VAR=a against VAR=b) on one box, two lanes and one.  usage: env_ab.sh VAR a b [bench args]
VAR=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do
  for V in $A $B; do
    for L in 2 1; do
      env $VAR=$V python bench.py --in-flight $L --no-cpu-baseline --no-alt-math --no-host-input "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$V lanes=$L run $i: %.1f img/s  frac %.4f' % (d['value'], d['roofline']['frac']))"
    done
  done
done
