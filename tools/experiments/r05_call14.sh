#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_train_step.py -q -x -m gpu -k "prefetched or side_stream or multi_scale" > $O/pytest_prefetch.txt 2>&1; tail -15 $O/pytest_prefetch.txt
for rep in 1 2; do
python bench.py --train --no-cpu-baseline --no-alt-math --no-pmc > $O/train_pipe_$rep.json 2> $O/train_pipe_$rep.err; tail -2 $O/train_pipe_$rep.err
python -c "
import json;d=json.loads(open('$O/train_pipe_$rep.json').read().strip().splitlines()[-1]);print('train', d['value'], d['ms_per_step'], d.get('one_step_at_a_time'), d['loss_first'], d['loss_last'])"
done
