#!/bin/bash
# Round 5, second GPU call: CU-masked lanes, alternating A/B of several partitions (whole-XCD masks are ignored by the runtime:
# gpurun_out/r05/cu_mask_probe_xcd.txt, so every partition here is a share of EVERY XCD).
O=gpurun_out/r05
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
run() {  # tag depth spec
  PPYOLO_HIP_LANE_CUS="$3" timeout 300 $B --in-flight $2 > $O/ab_$1.json 2> $O/ab_$1.err
}
for rep in 1 2; do
  run base_$rep 2 ''
  run half_$rep 2 'm256:0-127|m256:128-255'
  run halfse_$rep 2 'm32:0-15|m32:16-31'
  run asym_$rep 2 'all|m256:0-127'
  run over_$rep 2 'm256:0-191|m256:64-255'
done
run d3_third 3 'm256:0-95|m256:80-175|m256:160-255'
run d4_quarter 4 'm256:0-63|m256:64-127|m256:128-191|m256:192-255'
run d3_base 3 ''
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-44s value %8.1f sustained %8.1f one-lane %8.1f' % (f, d['value'], d['sustained']['value'], (d.get('one_batch_at_a_time') or {}).get('value', 0)))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
