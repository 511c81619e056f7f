#!/bin/bash
# in-situ re-ranking for the other two math modes of the headline workload (value_fp32_exact, bf16x3)
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for mode in fp32 bf16x3; do
 export PPYOLO_HIP_MATH=$mode
 T=$O/tuned_insitu_$mode.json
 unset PPYOLO_HIP_TUNE_CACHE
 timeout 2400 $B --autotune --insitu-tune --save-tuning $T > /dev/null 2> $O/insitu_$mode.err; grep "layers changed" $O/insitu_$mode.err
 for rep in 1 2; do
  for v in base new; do
   if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$T; fi
   timeout 300 $B 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$mode table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
  done
 done
done
