#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.txt 2>&1; tail -4 $O/pytest_gpu_full.txt
python bench.py --train > $O/r05_bench_train_r50vd_608.json 2> $O/train_full.err; tail -2 $O/train_full.err
python -c "
import json;d=json.loads(open('$O/r05_bench_train_r50vd_608.json').read().strip().splitlines()[-1]);print('train', d['value'], d['ms_per_step'], d.get('one_step_at_a_time'), d.get('value_fp32_exact'), d['roofline']['frac'], d['roofline'].get('traffic'), d.get('cpu_baseline',{}).get('value'))"
