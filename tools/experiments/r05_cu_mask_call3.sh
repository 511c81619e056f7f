#!/bin/bash
# Round 5, third GPU call: the tile table re-measured on HALF of every XCD (what a CU-masked lane would want), then the A/B again
# with power / clock samples, and the kernel timeline of the masked pair.
O=gpurun_out/r05
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
HALF='m256:0-127|m256:128-255'
( time timeout 1500 $B --in-flight 1 --autotune --tune-cu-mask m256:0-127 --save-tuning $O/tuned_half.json ) > $O/tune_half.json 2> $O/tune_half.err
for rep in 1 2; do
  timeout 300 $B --power-trace > $O/pw_base_$rep.json 2> $O/pw_base_$rep.err
  PPYOLO_HIP_LANE_CUS="$HALF" timeout 300 $B --power-trace > $O/pw_half_$rep.json 2> $O/pw_half_$rep.err
  PPYOLO_HIP_LANE_CUS="$HALF" PPYOLO_HIP_TUNE_CACHE=$PWD/$O/tuned_half.json timeout 300 $B --power-trace > $O/pw_halftuned_$rep.json 2> $O/pw_halftuned_$rep.err
done
timeout 300 $B --in-flight 1 --power-trace > $O/pw_one.json 2> $O/pw_one.err
# kernel timelines: the masked pair (tuned table) and the unmasked pair
REPO=$PWD
export TMPDIR=/tmp
for tag in half base; do
  SCR=/tmp/tl_$tag; rm -rf $SCR; mkdir -p $SCR
  if [ $tag = half ]; then export PPYOLO_HIP_LANE_CUS="$HALF" PPYOLO_HIP_TUNE_CACHE=$REPO/$O/tuned_half.json; else unset PPYOLO_HIP_LANE_CUS PPYOLO_HIP_TUNE_CACHE; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $SCR -o t -- python $REPO/bench.py --steps 240 --warmup 10 --min-seconds 0 \
      --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-worst-case --no-batch-scaling > $REPO/$O/tl_$tag.log 2>&1 )
  python tools/two_lane_timeline.py $SCR $O/two_lane_timeline_$tag.txt --skip 40 --count 160
done
unset PPYOLO_HIP_LANE_CUS PPYOLO_HIP_TUNE_CACHE
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/pw_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-44s value %8.1f sustained %8.1f one-lane %8.1f power %s' % (f, d['value'], d['sustained']['value'], (d.get('one_batch_at_a_time') or {}).get('value', 0), d['sustained'].get('power')))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json', '.err')).read()[-300:])
PY
tail -3 $O/tune_half.err
