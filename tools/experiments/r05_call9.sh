#!/bin/bash
O=gpurun_out/r05
python tools/experiments/r05_quantisation_probe.py 2>&1 | grep -v amdgpu.ids
python tools/experiments/r05_train_copy_sites.py 2>&1 | grep -v amdgpu.ids | tee $O/train_copy_sites.txt
