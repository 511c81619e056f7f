#!/bin/bash
# GPU session: the training-side tests, the training line, a kernel trace of the training step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
(time python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_train_loop.py -m gpu -x -q) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
python bench.py --train --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; python -c "
import json;d=json.loads([l for l in open('$O/bench_train.json') if l.startswith('{')][-1]);print('train',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
python tools/prof_summarize.py /tmp/prof_tr $O/train_trace.txt 7 > /dev/null 2> $O/train_trace.err
grep -A40 "^kernel " $O/train_trace.txt | cut -c1-150 | head -44
