#!/bin/bash
# Round 5, first GPU call: what does a stream's CU mask select on MI355X, does a hipGraph keep it, first A/B of masked lanes.
# The stages that empty whole XCDs come LAST, each in its own process under `timeout`.
O=gpurun_out/r05
mkdir -p $O
P=tools/probes/bin/cu_mask_probe
SAFE='h:ff00ff,ff00ff00,ff00ff,ff00ff00,ff00ff,ff00ff00,ff00ff,ff00ff00|h:ff00ff00,ff00ff,ff00ff00,ff00ff,ff00ff00,ff00ff,ff00ff00,ff00ff'
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case"
{
timeout 90 $P map
timeout 90 $P half
timeout 90 $P pair
} > $O/cu_mask_probe.txt 2>&1
for rep in 1 2; do
  $B > $O/bench_base_$rep.json 2> $O/bench_base_$rep.err
  PPYOLO_HIP_LANE_CUS="$SAFE" $B > $O/bench_safehalf_$rep.json 2> $O/bench_safehalf_$rep.err
done
{
timeout 60 $P half8
timeout 60 $P xcd
timeout 60 $P pairx
} > $O/cu_mask_probe_xcd.txt 2>&1
echo "rc $?" >> $O/cu_mask_probe_xcd.txt
# if the device survived: whole-XCD lanes under either bit order
rocm-smi --showuse > $O/smi_after.txt 2>&1
for spec in 'm8:0-3|m8:4-7' 'm256:0-127|m256:128-255'; do
  tag=$(echo "$spec" | tr -c 'a-z0-9' '_')
  PPYOLO_HIP_LANE_CUS="$spec" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-60s value %8.1f sustained %8.1f one-lane %8.1f' % (f, d['value'], d['sustained']['value'], d['one_batch_at_a_time']['value']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
