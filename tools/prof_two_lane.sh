#!/bin/bash
# rocprofv3 kernel trace of bench.py's two-lane timed loop (hipGraph replay on two streams) -> profiles/<tag>_two_lane_timeline.txt
# usage (GPU box): tools/prof_two_lane.sh r04
TAG=${1:-r04}
REPO=$PWD
SCR=/tmp/two_lane_$TAG
rm -rf $SCR; mkdir -p $SCR $REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $SCR -o t -- python $REPO/bench.py --steps 240 --warmup 10 --min-seconds 0 \
    --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $REPO/gpurun_out/two_lane_$TAG.log 2>&1
cd $REPO
python tools/two_lane_timeline.py $SCR gpurun_out/${TAG}_two_lane_timeline.txt --skip 40 --count 160
