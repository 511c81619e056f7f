#!/bin/bash
# headline line again after the bench fixes (fused launches counted in the PMC traffic and in the per-class table), + batch-1 streams probe
O=gpurun_out/refresh; mkdir -p $O
python bench.py --layer-report $O/r05_layers_r50vd_608_bs8.json > $O/r05_bench_r50vd_608.json 2> $O/bench_r50.err
python bench.py --in-flight 1 --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-worst-case > $O/r05_bench_r50vd_608_one_lane.json 2>/dev/null
for f in $O/r05_bench_r50vd_608.json $O/r05_bench_r50vd_608_one_lane.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
bash tools/prof_run.sh r05 > $O/prof_run.log 2>&1
for n in trace pmc_sq pmc_fetch pmc_write pmc_lds; do cp gpurun_out/prof_r05/$n.txt $O/r05_$n.txt 2>/dev/null; done
