#!/bin/bash
# Where does the time of the patch kernel (csrc/conv_patch.hip) go?  Rebuilds ONLY that translation unit with PPY_PATCH_ABL = 0..5,
# links a side library next to the product one and times one layer.  usage (GPU box): tools/patch_ablate.sh "N,H,W,32,K,3,1"
D=pytorch-ppyolo_amd/ppyolo_hip
mkdir -p gpurun_out/abl
for abl in ${ABLS:-0 1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DPPY_PATCH_ABL=$abl \
     -c $D/csrc/conv_patch.hip -o gpurun_out/abl/conv_patch.o 2>/dev/null
  OBJS=$(ls $D/lib/obj/*.o | grep -v conv_patch.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS gpurun_out/abl/conv_patch.o -o gpurun_out/abl/lib_abl.so
  echo "== PPY_PATCH_ABL=$abl"
  PPYOLO_HIP_LIB=$PWD/gpurun_out/abl/lib_abl.so python tools/conv_bench.py "$1" $(python -c "
import sys; sys.path.insert(0,'pytorch-ppyolo_amd')
from ppyolo_hip import ops; print(ops.patch_first_cfg())") 2>&1 | grep -v amdgpu.ids
done
rm -f gpurun_out/abl/lib_abl.so gpurun_out/abl/conv_patch.o
