#!/bin/bash
# Alternating A/B of candidate table entries (JSON files given as arguments, loaded on top of the committed table through
# PPYOLO_HIP_TUNE_CACHE; "base" = the committed table alone) on one box, two lanes and one.  -> gpurun_out/table_ab.log
mkdir -p gpurun_out
OUT=gpurun_out/table_ab.log
: > $OUT
for i in 1 2 3; do
  for V in base "$@"; do
    if [ $V = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$V; fi
    for L in 2 1; do
      python bench.py --in-flight $L --no-cpu-baseline --no-alt-math --no-host-input ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V lanes=$L run $i: %.1f img/s  frac %.4f' % (d['value'], d['roofline']['frac']))" >> $OUT
    done
  done
done
cat $OUT
