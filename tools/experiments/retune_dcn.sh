#!/bin/bash
# Re-measure the (scheme, tile, split-K) choice of the fused DCNv2 kernel for the R50vd bench workloads, all three math modes.
# -> gpurun_out/dcn_tune/tuned_<mode>_<workload>.json (only the 'dcnf:' keys are merged: tools/merge_tuned.py --only dcnf)
mkdir -p gpurun_out/dcn_tune
for MODE in f16x2 bf16x3 fp32; do
  for W in "r50vd_608 8" "r50vd_608 1" "r50vd_320 1"; do
    set -- $W
    PPYOLO_HIP_MATH=$MODE python bench.py --workload $1 --batch $2 --autotune --tune-kinds dcn --verbose-tune \
      --save-tuning gpurun_out/dcn_tune/tuned_${MODE}_$1_b$2.json --no-cpu-baseline --no-alt-math --no-host-input --steps 20 \
      2>/dev/null | grep "^autotune" | sed "s/^/$MODE $1 b$2: /"
  done
done
