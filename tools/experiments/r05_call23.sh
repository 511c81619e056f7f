#!/bin/bash
# tile order in column panels where the weights outweigh the activations (conv_shared.h ppy_panel_n / ppy_tile_of): tests, traffic, A/B
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "config or conv or presplit or split" > $O/pytest_panel.txt 2>&1; tail -3 $O/pytest_panel.txt
timeout 600 python tools/pmc_layers.py --out $O/pmc_layers_panel.txt > /dev/null 2>&1; tail -3 $O/pmc_layers_panel.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2 3; do
 for v in 0 1; do
  PPY_TILE_PANEL=$v timeout 300 $B --layer-report $O/layers_panel_$v.json > $O/panel_${v}_$rep.json 2> $O/panel_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/panel_${v}_$rep.json').read().strip().splitlines()[-1]);print('PPY_TILE_PANEL=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/panel_${v}_$rep.err
 done
done
python - <<PY
import json
a = json.load(open('$O/layers_panel_0.json')); b = json.load(open('$O/layers_panel_1.json'))
for r0, r1 in zip(a, b):
    if abs(r0['ms'] - r1['ms']) > 0.0025:
        print('%3d %-60s cfg %-4s %.4f -> %.4f ms' % (r0['i'], r0['key'][:60], r0['cfg'], r0['ms'], r1['ms']))
print('sum', sum(r['ms'] for r in a), sum(r['ms'] for r in b))
PY
