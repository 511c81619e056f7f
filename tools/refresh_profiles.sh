#!/bin/bash
# Regenerate the measured set under profiles/ on the GPU box (run through gpurun; results land in gpurun_out/refresh/,
# copy them into profiles/ afterwards).  usage: tools/refresh_profiles.sh <tag>
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; mkdir -p $O
python bench.py --layer-report $O/${TAG}_layers_r50vd_608_bs8.json --trace-layers $O/${TAG}_one_lane_layers_trace.txt > $O/${TAG}_bench_r50vd_608.json 2> $O/bench_r50.err
PPYOLO_HIP_MATH=bf16x3 python bench.py --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r50vd_608_math_bf16x3.json 2>/dev/null
PPYOLO_HIP_MATH=fp32 python bench.py --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r50vd_608_math_fp32.json 2>/dev/null
python bench.py --in-flight 1 --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r50vd_608_one_lane.json 2>/dev/null
python bench.py --workload r18vd_416 --no-host-input --no-worst-case --layer-report $O/${TAG}_layers_r18vd_416_bs8.json > $O/${TAG}_bench_r18vd_416.json 2>/dev/null
python bench.py --workload r18vd_320 --no-cpu-baseline --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r18vd_320_bs8.json 2>/dev/null
python bench.py --workload r18vd_320 --batch 1 --no-cpu-baseline --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r18vd_320_bs1.json 2>/dev/null
bash tools/prof_run.sh $TAG > $O/prof_run.log 2>&1
for n in pmc_sq pmc_fetch pmc_write pmc_lds; do cp gpurun_out/prof_$TAG/$n.txt $O/${TAG}_$n.txt 2>/dev/null; done
cp gpurun_out/prof_$TAG/trace.txt $O/${TAG}_one_lane_kernel_trace_stats.txt 2>/dev/null      # (prof_run.sh: one lane, eager, no graph)
for f in $O/${TAG}_bench_*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; done
for spec in "r50vd_608 1" "r50vd_320 1" "r18vd_608 1" "r18vd_416 1"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_$1_bs$2.json 2>/dev/null
done
# ---- round 3 additions ----
python tools/decode_bench.py > $O/${TAG}_decode_bench.txt 2>/dev/null
python tools/fullsize_parity.py > $O/${TAG}_fullsize_parity.txt 2>/dev/null
python tools/train_fullsize_parity.py --tag $TAG --out $O/${TAG}_train_parity.txt > /dev/null 2>&1
python bench.py --train > $O/${TAG}_bench_train_r50vd_608.json 2> $O/bench_train.err
bash tools/prof_train.sh $TAG > $O/prof_train.log 2>&1
cp gpurun_out/prof_train_$TAG/trace.txt $O/${TAG}_train_kernel_trace_stats.txt 2>/dev/null
cp gpurun_out/prof_train_$TAG/${TAG}_train_pmc_traffic.json $O/ 2>/dev/null
cp gpurun_out/prof_train_$TAG/pmc_fetch.txt $O/${TAG}_train_pmc_fetch.txt 2>/dev/null
cp gpurun_out/prof_train_$TAG/pmc_write.txt $O/${TAG}_train_pmc_write.txt 2>/dev/null
bash tools/ab_env.sh PPYOLO_HIP_TUNE_CACHE "tools/probes/r06_pre_kp_entries.json none" 2 --no-pmc --no-kernel-trace --no-batch-scaling > $O/${TAG}_kp_table_ab.txt 2>&1      # round 6: the k-parity entries against the tiles they replaced
# ---- round 4 additions ----
python bench.py --batch 4 --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-worst-case > $O/${TAG}_bench_r50vd_608_bs4.json 2>/dev/null
bash tools/prof_two_lane.sh $TAG > $O/prof_two_lane.log 2>&1
cp gpurun_out/${TAG}_two_lane_timeline.txt $O/ 2>/dev/null
python -m pytest tests/test_gpu_model.py -q -s -k "headline_sizes" 2>&1 | grep -v "^$" | tail -40 > $O/${TAG}_g18_gpu.txt
