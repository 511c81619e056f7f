#!/bin/bash
# Re-measure the tile choice of the 1x1 layers with C = $1 (64 or 128; f16x2 mode), where the streaming kernel of csrc/conv_stream.hip is a
# candidate (the last two conv cfg ids), for the bench workloads.  -> gpurun_out/stream_tune/tuned_<workload>_b<batch>.json
# (merge: tools/merge_tuned.py f16x2 --match ":C64:,:R1:" gpurun_out/stream_tune/*.json)
CC=${1:-64}
RR=${2:-1}            # (MATCH=":R1:" overrides the key filter: e.g. every 1x1 layer, where the specialised-wave tiles of csrc/conv_ws.hip can win)
# (retune_stream.sh 32 3: the 3x3 stem layers, where the patch kernel of csrc/conv_patch.hip is a candidate)
mkdir -p gpurun_out/stream_tune
for W in ${WORKLOADS:-r50vd_608:8 r50vd_608:1 r50vd_320:1 r18vd_416:8 r18vd_416:1 r18vd_320:8 r18vd_320:1 r18vd_608:1}; do
  set -- ${W/:/ }
  python bench.py --workload $1 --batch $2 --autotune --tune-kinds conv --tune-match "${MATCH:-:C$CC:,:R$RR:}" --verbose-tune \
    --save-tuning gpurun_out/stream_tune/tuned_c${CC}_$1_b$2.json --no-cpu-baseline --no-alt-math --no-host-input --steps 20 \
    2>&1 | grep "^autotune\|\"value\"" | cut -c1-400 | sed "s/^/$1 b$2: /"
done
