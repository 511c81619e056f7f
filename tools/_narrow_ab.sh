timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "narrow or random_shapes or specialised or ws_" 2>&1 | tail -5
B="--no-cpu-baseline --no-alt-math --no-host-input --no-worst-case --no-pmc"
for r in 1 2; do
python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base   two %.1f one %s' % (d['value'], d.get('one_batch_at_a_time',{}).get('value')))"
python bench.py $B --autotune --tune-match K27 --verbose-tune --save-tuning gpurun_out/tuned_k27.json 2>gpurun_out/narrow_tune_$r.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('narrow two %.1f one %s' % (d['value'], d.get('one_batch_at_a_time',{}).get('value')))"
done
grep -n "K27" gpurun_out/narrow_tune_1.log | head -20
