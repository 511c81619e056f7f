#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
python -m pytest tests/test_preprocess.py -q -x -m gpu > $O/pytest_pre.txt 2>&1; tail -2 $O/pytest_pre.txt
for v in 0 1; do PPY_PRE_PIXEL=$v python bench.py --no-cpu-baseline --no-alt-math --no-pmc --no-worst-case --no-batch-scaling > $O/pre_$v.json 2> $O/pre_$v.err; python -c "
import json;d=json.loads(open('$O/pre_$v.json').read().strip().splitlines()[-1]);p=d['roofline_other'].get('preprocess');print('PPY_PRE_PIXEL=$v', p['us_per_launch'], p['frac'], p['from_raw_host_images_per_s'])"; done
python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case > $O/bs_scaling.json 2> $O/bs_scaling.err
python -c "
import json;d=json.loads(open('$O/bs_scaling.json').read().strip().splitlines()[-1])
for r in d['roofline'].get('batch_scaling'): print(r)"
python bench.py --train --no-cpu-baseline > $O/train_base.json 2> $O/train_base.err; python -c "
import json;d=json.loads(open('$O/train_base.json').read().strip().splitlines()[-1]);print('train', d['value'], d['ms_per_step'])"
bash tools/prof_train.sh r05 > $O/prof_train.log 2>&1
head -60 gpurun_out/prof_train_r05/trace.txt | cut -c1-150
