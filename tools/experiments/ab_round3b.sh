#!/bin/bash
# GPU session of round 3, second half: full -m gpu suite, the training line, a trace of the training step, stem A/B.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
python bench.py --train --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; tail -c 600 $O/bench_train.json; echo
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --train --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
python tools/prof_summarize.py /tmp/prof_tr $O/train_trace.txt 7 > /dev/null 2> $O/train_trace.err
grep -A45 "^kernel " $O/train_trace.txt | cut -c1-150 | head -60
for v in 1 0 1 0; do PPY_STEM_OLD=$v python bench.py --no-cpu-baseline --no-alt-math --no-host-input 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print('PPY_STEM_OLD=$v',d['value'],d['sustained']['value'],d['one_batch_at_a_time']['value'],d['roofline']['achieved'],d['roofline']['frac'])"; done | tee $O/stem_ab.txt
