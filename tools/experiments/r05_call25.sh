#!/bin/bash
# the 19x19 / 38x38 layers re-measured under the new tile orders; A/B of the resulting table entries through PPYOLO_HIP_TUNE_CACHE
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
timeout 1500 $B --autotune --tune-match "H19|H38" --save-tuning $O/tuned_h19_h38.json > $O/retune.json 2> $O/retune.err; tail -2 $O/retune.err
python - <<PY
import json
new = json.load(open('$O/tuned_h19_h38.json')); old = json.load(open('pytorch-ppyolo_amd/ppyolo_hip/tuned_gfx950_f16x2.json'))
ch = {k: v for k, v in new.items() if ('H19' in k or 'H38' in k)}
json.dump(ch, open('$O/tuned_h19_h38_only.json', 'w'), indent=0, sort_keys=True)
for k, v in sorted(ch.items()):
    o = old.get(k)
    if o is None or o[:2] != v[:2]:
        print(k, o, '->', v)
PY
for rep in 1 2 3; do
 for v in base new; do
  if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$O/tuned_h19_h38_only.json; fi
  timeout 300 $B > $O/retune_${v}_$rep.json 2> $O/retune_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/retune_${v}_$rep.json').read().strip().splitlines()[-1]);print('table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/retune_${v}_$rep.err
 done
done
