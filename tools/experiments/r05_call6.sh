#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
python -m pytest tests/test_preprocess.py -q -x -m gpu > $O/pytest_pre.txt 2>&1; tail -3 $O/pytest_pre.txt
python tools/energy_tune.py --out $O/tuned_energy.json > $O/energy_tune.txt 2>&1
tail -4 $O/energy_tune.txt
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling --power-trace"
for rep in 1 2 3; do
  timeout 300 $B > $O/et_base_$rep.json 2> $O/et_base_$rep.err
  PPYOLO_HIP_TUNE_CACHE=$PWD/$O/tuned_energy_time_only.json timeout 300 $B > $O/et_time_$rep.json 2> $O/et_time_$rep.err
  PPYOLO_HIP_TUNE_CACHE=$PWD/$O/tuned_energy.json timeout 300 $B > $O/et_energy_$rep.json 2> $O/et_energy_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/et_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-44s value %8.1f sustained %8.1f one-lane %8.1f conv_ms %.3f power %s' % (f, d['value'], d['sustained']['value'], (d.get('one_batch_at_a_time') or {}).get('value', 0), d['roofline']['kernel_ms_per_step'], d['sustained'].get('power')))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
# preprocess A/B inside the bench's own leg
for v in 0 1; do PPY_PRE_PIXEL=$v python bench.py --no-cpu-baseline --no-alt-math --no-pmc --no-worst-case --no-batch-scaling > $O/pre_$v.json 2> $O/pre_$v.err; python -c "
import json;d=json.loads(open('$O/pre_$v.json').read().strip().splitlines()[-1]);print('PPY_PRE_PIXEL=$v', d['roofline_other'].get('preprocess'))"; done
