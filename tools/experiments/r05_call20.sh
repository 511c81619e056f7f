#!/bin/bash
# batch 1, one image at a time (the reference's demo case): independent branches of a step on a second captured stream
# (PPYOLO_HIP_STREAMS=2, opt-in since round 2) -- what is it worth where every launch underfills the chip?
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --batch 1 --in-flight 1 --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case --no-batch-scaling"
for wl in r50vd_608 r18vd_320; do
 for rep in 1 2; do
  for v in 1 2; do
   PPYOLO_HIP_STREAMS=$v timeout 300 $B --workload $wl > $O/bs1_streams_${wl}_${v}_$rep.json 2> $O/bs1_streams_${wl}_${v}_$rep.err
   python -c "
import json;d=json.loads(open('$O/bs1_streams_${wl}_${v}_$rep.json').read().strip().splitlines()[-1]);print('$wl STREAMS=$v', d['value'], d['ms_per_step'])" || tail -3 $O/bs1_streams_${wl}_${v}_$rep.err
  done
 done
done
