#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv3x3_conv1x1" > $O/pytest_b2b.txt 2>&1; tail -25 $O/pytest_b2b.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "b2b" >> $O/pytest_b2b.txt 2>&1; tail -25 $O/pytest_b2b.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
 for v in 0 1; do
  PPYOLO_HIP_B2B=$v timeout 300 $B > $O/b2b_${v}_$rep.json 2> $O/b2b_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/b2b_${v}_$rep.json').read().strip().splitlines()[-1]);print('B2B=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/b2b_${v}_$rep.err
 done
done
