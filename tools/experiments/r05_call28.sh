#!/bin/bash
# in-situ re-ranking of every layer's front-runners (HipExecutor.insitu_tune), then A/B of the resulting table
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
S=$(date +%s)
timeout 2400 $B --autotune --insitu-tune --insitu-topk 20 --save-tuning $O/tuned_insitu20.json > $O/insitu.json 2> $O/insitu.err; echo "tune seconds $(( $(date +%s) - S ))"; grep -c insitu_tune $O/insitu.err; grep "layers changed" $O/insitu.err
python - <<PY
import json
new = json.load(open('$O/tuned_insitu20.json')); old = json.load(open('pytorch-ppyolo_amd/ppyolo_hip/tuned_gfx950_f16x2.json'))
n = 0
for k, v in sorted(new.items()):
    o = old.get(k)
    if o is None or o[:2] != v[:2]:
        n += 1
        print(k, o, '->', v)
print(n, 'entries differ')
PY
for rep in 1 2 3; do
 for v in base new; do
  if [ $v = base ]; then unset PPYOLO_HIP_TUNE_CACHE; else export PPYOLO_HIP_TUNE_CACHE=$O/tuned_insitu20.json; fi
  timeout 300 $B > $O/insitu20_${v}_$rep.json 2> $O/insitu20_${v}_$rep.err
  python -c "
import json;d=json.loads(open('$O/insitu20_${v}_$rep.json').read().strip().splitlines()[-1]);print('table=$v', d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])" || tail -5 $O/insitu20_${v}_$rep.err
 done
done
