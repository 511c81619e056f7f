#!/bin/bash
# in-situ re-ranking for the other workloads of the refresh set; A/B per workload through PPYOLO_HIP_TUNE_CACHE
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
IFS=";" read -ra LIST <<< "${SPECS:-r18vd_320 1;r50vd_608 1;r18vd_416 8;r18vd_320 8}"
for spec in "${LIST[@]}"; do
 set -- $spec
 T=$O/tuned_insitu_$1_bs$2.json
 timeout 1200 $B --workload $1 --batch $2 --autotune --insitu-tune --save-tuning $T > /dev/null 2> $O/insitu_$1_bs$2.err; grep "layers changed" $O/insitu_$1_bs$2.err
 for rep in 1 2; do
  for v in base new; do
   if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$T; fi
   timeout 300 $B --workload $1 --batch $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1 bs$2 table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'])"
  done
 done
 unset PPYOLO_HIP_TUNE_CACHE
done
