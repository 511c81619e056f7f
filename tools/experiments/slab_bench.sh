#!/bin/bash
# Slab-reuse variants (cfg 67..84) against the plain f16x2 tiles (40..66) on the six big 3x3 layer shapes of R50vd-608 bs 8.
# usage (GPU box): tools/slab_bench.sh > gpurun_out/slab_bench.txt
PLAIN=40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57
SLAB=67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84
for shp in 8,76,76,128,256,3,1 8,38,38,256,512,3,1 8,19,19,512,1024,3,1 8,152,152,64,64,3,1 8,76,76,128,128,3,1 8,38,38,256,256,3,1 8,304,304,32,64,3,1; do
  for sp in 1 2; do
    python tools/conv_bench.py $shp $PLAIN,$SLAB $sp 2>&1 | grep -v amdgpu.ids | sort -t: -k2 -n | awk '{print}' | sed 's/dbg=0 //' | sort -k4 -n | head -8
    echo
  done
done
