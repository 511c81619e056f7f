#!/bin/bash
# conv_b2b.hip, second version (column halves, dedicated patches, pooled rows): op tests, A/B, per-layer report
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv3x3_conv1x1" > $O/pytest_b2b.txt 2>&1; tail -25 $O/pytest_b2b.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "b2b" >> $O/pytest_b2b.txt 2>&1; tail -8 $O/pytest_b2b.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
 for v in 0 1; do
  PPYOLO_HIP_B2B=$v timeout 300 $B > $O/b2b3_${v}_$rep.json 2> $O/b2b3_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/b2b3_${v}_$rep.json').read().strip().splitlines()[-1]);print('B2B=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/b2b3_${v}_$rep.err
 done
done
for v in 0 1; do
PPYOLO_HIP_B2B=$v timeout 300 python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling --layer-report $O/layers_b2b3_$v.json > /dev/null 2>&1
python - <<PY
import json
rows=json.load(open('$O/layers_b2b3_$v.json'))
print('B2B=$v')
for r in rows[:16]:
    print('  %3d %-46s cfg %-4s %.4f ms %6.1f TF' % (r['i'], r['key'], r['cfg'], r['ms'], r['tflops']))
PY
done
timeout 300 python tools/experiments/r05_b2b_phases.py 2>&1 | tee $O/b2b_phases5.txt
