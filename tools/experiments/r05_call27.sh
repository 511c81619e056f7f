#!/bin/bash
# split-K combine inside the launch (ppy_conv2d_splitk_counters): op test, model tests, A/B at batch 1 and batch 8
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "splitk_combines or specialised_wave or random_shapes or every_tile" > $O/pytest_fixup.txt 2>&1; tail -5 $O/pytest_fixup.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "full_size_batch or cu_masked or in_flight or lanes" >> $O/pytest_fixup.txt 2>&1; tail -3 $O/pytest_fixup.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for spec in "r50vd_608 1" "r18vd_320 1" "r18vd_416 8" "r50vd_608 8"; do
 set -- $spec
 for rep in 1 2; do
  for v in 0 1; do
   PPYOLO_HIP_SPLITK_FIXUP=$v timeout 300 $B --workload $1 --batch $2 > $O/fixup_$1_bs$2_${v}_$rep.json 2> $O/fixup_$1_bs$2_${v}_$rep.err
   python -c "
import json;d=json.loads(open('$O/fixup_$1_bs$2_${v}_$rep.json').read().strip().splitlines()[-1]);print('$1 bs$2 FIXUP=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], 1000.0*$2/d['one_batch_at_a_time']['value'])" || tail -5 $O/fixup_$1_bs$2_${v}_$rep.err
  done
 done
done
