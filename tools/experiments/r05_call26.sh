#!/bin/bash
# batch 1, one image at a time: how much of the step is kernels and how much is the gap between dependent graph nodes?
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
for wl in r50vd_608 r18vd_320; do
  SCR=/tmp/bs1_$wl; rm -rf $SCR; mkdir -p $SCR
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $SCR -o t -- python $REPO/bench.py --workload $wl --batch 1 --in-flight 1 --steps 240 --warmup 10 --min-seconds 0 \
     --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-worst-case --no-batch-scaling > $REPO/$O/bs1_trace_$wl.log 2>&1)
  python tools/two_lane_timeline.py $SCR $O/bs1_timeline_$wl.txt --skip 40 --count 160; head -30 $O/bs1_timeline_$wl.txt
done
