#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
python bench.py --no-cpu-baseline --no-host-input --no-alt-math --no-pmc --no-worst-case > $O/bs_scaling.json 2> $O/bs_scaling.err
python -c "
import json;d=json.loads(open('$O/bs_scaling.json').read().strip().splitlines()[-1]);print(d['value'], d['one_batch_at_a_time']);print(json.dumps(d['roofline'].get('batch_scaling'),indent=1))"
for v in 0 1; do PPY_PRE_PIXEL=$v python bench.py --no-cpu-baseline --no-alt-math --no-pmc --no-worst-case --no-batch-scaling > $O/pre_$v.json 2> $O/pre_$v.err; python -c "
import json;d=json.loads(open('$O/pre_$v.json').read().strip().splitlines()[-1]);print('PPY_PRE_PIXEL=$v', d['roofline_other'].get('preprocess'))"; done
