python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "tile_config or random_shapes or bf16x3" -s 2>&1 | tail -8
for shp in "8,38,38,256,512,3,1" "8,76,76,128,256,3,1" "8,19,19,512,1024,3,1" "8,38,38,1024,256,1,1" "8,152,152,64,64,3,1" "8,38,38,256,1024,1,1,1" "8,76,76,128,512,1,1,1"; do
  python tools/conv_bench.py $shp 19,31,32,33,34,35,36,37,38 1 2>&1 | grep -v "^$"
done
