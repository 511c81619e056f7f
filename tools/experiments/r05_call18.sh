#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
for v in 0 1; do
PPYOLO_HIP_B2B=$v python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling --layer-report $O/layers_b2b_$v.json > /dev/null 2>&1
python - <<PY
import json
rows=json.load(open('$O/layers_b2b_$v.json'))
print('B2B=$v')
for r in rows[:16]:
    print('  %3d %-46s cfg %-4s %.4f ms %6.1f TF' % (r['i'], r['key'], r['cfg'], r['ms'], r['tflops']))
PY
done
