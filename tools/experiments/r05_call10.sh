#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.txt 2>&1; tail -5 $O/pytest_gpu_full.txt
python bench.py --train > $O/train_full.json 2> $O/train_full.err; tail -2 $O/train_full.err
python -c "
import json;d=json.loads(open('$O/train_full.json').read().strip().splitlines()[-1]);print('train', d['value'], d['ms_per_step'], d.get('value_fp32_exact'), d.get('value_fp32_exact_note'), d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))"
