#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.txt 2>&1; tail -4 $O/pytest_gpu_full.txt
for v in 0 1 2 3; do PPY_PRE_VARIANT=$v python bench.py --no-cpu-baseline --no-alt-math --no-pmc --no-worst-case --no-batch-scaling > $O/prev_$v.json 2> $O/prev_$v.err; python -c "
import json;d=json.loads(open('$O/prev_$v.json').read().strip().splitlines()[-1]);p=d['roofline_other'].get('preprocess');print('PPY_PRE_VARIANT=$v', p['us_per_launch'], p['frac'])"; done
