#!/bin/bash
# Rebuild libppyolo_hip.so with an ablation switch (PPY_X3_ABL in csrc/conv_x3.hip: 1..9) and time one layer: where does the 16-bit-MFMA kernel's time go?
# usage (on the GPU box): tools/x3_ablate.sh "N,H,W,C,K,R,stride" cfg[,cfg..]
for abl in ${ABLS:-0 1 2}; do
  PPY_EXTRA_HIPCC_FLAGS="-DPPY_X3_ABL=$abl" python pytorch-ppyolo_amd/ppyolo_hip/build.py > /dev/null 2>&1
  echo "== PPY_X3_ABL=$abl"
  PPY_EXTRA_HIPCC_FLAGS="-DPPY_X3_ABL=$abl" python tools/conv_bench.py "$1" "$2" 1
done
