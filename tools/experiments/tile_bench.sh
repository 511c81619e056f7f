#!/bin/bash
# The 96 / 192-row f16x2 tiles (cfg 85..93) against the best plain tiles on the layer shapes of R50vd-608 bs 8 whose grids
# quantise badly.  usage (GPU box): tools/tile_bench.sh > gpurun_out/tile_bench.txt
OLD=40,41,42,43,44,45,46,47,49,50,51,53,54,55,56,58,60,63
NEW=85,86,87,88,89,90,91,92,93
for shp in 8,76,76,128,256,3,1 8,38,38,256,512,3,1 8,19,19,512,1024,3,1 8,76,76,128,128,3,1 8,38,38,256,256,3,1 8,38,38,1024,256,1,1 8,38,38,256,1024,1,1,1 8,76,76,512,128,1,1 8,76,76,128,512,1,1,1 8,19,19,2048,512,1,1 8,19,19,512,2048,1,1,1 8,152,152,64,256,1,1,1 8,152,152,256,64,1,1 8,76,76,640,128,1,1 8,38,38,1280,256,1,1; do
  for sp in 1 2 4; do
    python tools/conv_bench.py $shp $OLD,$NEW $sp 2>&1 | grep "TF" | sed 's/dbg=0 //' | sort -k4 -n | head -4
  done
  echo
done
