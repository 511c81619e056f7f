#!/bin/bash
# Re-measure the per-shape (tile config, split-K) tables of the three bench workloads on the GPU box.
# usage: tools/retune.sh [fp32|bf16x3]   -> gpurun_out/tuned_<mode>_*.json (merge with tools/merge_tuned.py)
MODE=${1:-bf16x3}
export PPYOLO_HIP_MATH=$MODE
python bench.py --autotune --save-tuning gpurun_out/tuned_${MODE}_r50.json --no-cpu-baseline --no-alt-math > gpurun_out/bench_${MODE}_r50_tune.json 2> gpurun_out/bench_${MODE}_r50_tune.err
python bench.py --workload r18vd_416 --autotune --save-tuning gpurun_out/tuned_${MODE}_r18_416.json --no-cpu-baseline --no-alt-math > gpurun_out/bench_${MODE}_r18_416_tune.json 2>/dev/null
python bench.py --workload r18vd_320 --batch 1 --autotune --save-tuning gpurun_out/tuned_${MODE}_r18_320b1.json --no-cpu-baseline --no-alt-math > gpurun_out/bench_${MODE}_r18_320b1_tune.json 2>/dev/null
python bench.py --workload r18vd_320 --autotune --save-tuning gpurun_out/tuned_${MODE}_r18_320.json --no-cpu-baseline --no-alt-math > gpurun_out/bench_${MODE}_r18_320_tune.json 2>/dev/null
for f in gpurun_out/bench_${MODE}_*_tune.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['roofline']['achieved'], d['roofline']['frac'])"; done
