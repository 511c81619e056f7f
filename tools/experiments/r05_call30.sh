#!/bin/bash
# headline workload: autotune + in-situ re-ranking + co_tune (best neighbour of a second lane), A/B against the committed table
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
S=$(date +%s)
timeout 2400 $B --autotune --insitu-tune --co-tune --save-tuning $O/tuned_insitu_co.json > /dev/null 2> $O/insitu_co.err; echo "tune seconds $(( $(date +%s) - S ))"; grep "layers changed" $O/insitu_co.err
for rep in 1 2 3; do
 for v in base new; do
  if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$O/tuned_insitu_co.json; fi
  timeout 300 $B 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
 done
done
