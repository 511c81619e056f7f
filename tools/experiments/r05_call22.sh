#!/bin/bash
# conv_stream.hip, C = 128 form: activations requested TWO tiles ahead.  Kernel tests, the layer table, and the stage-3 expand layers
# (C128 -> K512 + shortcut, on a general tile in the table) forced onto the stream kernel through PPYOLO_HIP_TUNE_CACHE.
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py -q -x -m gpu -k "stream or expand or all_configs or bit_identical or statistics" > $O/pytest_stream2.txt 2>&1; tail -4 $O/pytest_stream2.txt
python - <<PY
import json
for c in (94, 95):
    json.dump({'conv:N8:H76:W76:C128:K512:R1:s1:f:g': [c, 1, 0.0], 'conv:N8:H76:W76:C128:K512:R1:s1:f': [c, 1, 0.0]}, open('/tmp/tune_%d.json' % c, 'w'))
PY
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for rep in 1 2; do
 for v in base 94 95; do
  if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=/tmp/tune_$v.json; fi
  timeout 300 $B --layer-report $O/layers_stream2_$v.json > $O/stream2_${v}_$rep.json 2> $O/stream2_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/stream2_${v}_$rep.json').read().strip().splitlines()[-1]);print('table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/stream2_${v}_$rep.err
 done
done
unset PPYOLO_HIP_TUNE_CACHE
python - <<PY
import json
for v in ('base', '94', '95'):
    rows = json.load(open('$O/layers_stream2_%s.json' % v))
    print(v, [(r['i'], r['cfg'], r['ms']) for r in rows if 'C128:K256:R1' in r['key'] or 'C128:K512:R1' in r['key']])
PY
