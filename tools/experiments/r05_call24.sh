#!/bin/bash
# dcn_fused: XCD-contiguous order over (split, tile); tests, traffic, A/B against the build without it is call 23's PANEL=1 numbers
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -m gpu -k "dcn or full_size_batch or b2b" > $O/pytest_dcnmap.txt 2>&1; tail -3 $O/pytest_dcnmap.txt
timeout 600 python tools/pmc_layers.py --out $O/pmc_layers_dcnmap.txt > /dev/null 2>&1; grep "dcnf\|total" $O/pmc_layers_dcnmap.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
  timeout 300 $B --layer-report $O/layers_dcnmap.json > $O/dcnmap_$rep.json 2> $O/dcnmap_$rep.err
  python -c "
import json;d=json.loads(open('$O/dcnmap_$rep.json').read().strip().splitlines()[-1]);print('dcnmap', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/dcnmap_$rep.err
done
python - <<PY
import json
for r in json.load(open('$O/layers_dcnmap.json')):
    if r['key'].startswith('dcnf'): print(r['i'], r['key'], r['cfg'], r['splitk'], r['ms'])
PY
