#!/bin/bash
# GPU session: training-side tests, the training line with / without the weight preparation table, wgrad9 slice targets, trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
(time python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_train_loop.py -m gpu -x -q) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
tr() { python bench.py --train --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]);print('train $*',d['value'],d['ms_per_step'],d['roofline']['achieved'])"; }
PPYOLO_HIP_TRAIN_PREP=0 tr PREP=0
tr default
PPYOLO_HIP_TRAIN_PREP=0 tr PREP=0
tr default
PPY_WGRAD9_WGS=256 tr WGS=256
PPY_WGRAD9_WGS=384 tr WGS=384
PPY_WGRAD9_WGS=768 tr WGS=768
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
python tools/prof_summarize.py /tmp/prof_tr $O/train_trace.txt 7 > /dev/null 2> $O/train_trace.err
grep -A36 "^kernel " $O/train_trace.txt | cut -c1-150 | head -40
