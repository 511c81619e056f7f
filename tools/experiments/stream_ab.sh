#!/bin/bash
# A/B of the streaming 1x1 kernel + pooled epilogue and of the patch kernel for the 3x3 stem layers against the tiles they replaced (tools/probes/old_c64_table.json = the table
# entries of the C = 64 / C = 128 1x1 layers and of the C = 32 3x3 layers before csrc/conv_stream.hip), alternating on one box.  -> gpurun_out/stream_ab.log
mkdir -p gpurun_out
OUT=gpurun_out/stream_ab.log
: > $OUT
for i in 1 2 3; do
  for V in new old; do
    if [ $V = old ]; then export PPYOLO_HIP_TUNE_CACHE=tools/probes/old_c64_table.json PPYOLO_HIP_POOL_FOLD=0; else unset PPYOLO_HIP_TUNE_CACHE PPYOLO_HIP_POOL_FOLD; fi
    for L in 2 1; do
      python bench.py --in-flight $L --no-cpu-baseline --no-alt-math --no-host-input 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V lanes=$L run $i: %.1f img/s  frac %.4f' % (d['value'], d['roofline']['frac']))" >> $OUT
    done
  done
done
cat $OUT
