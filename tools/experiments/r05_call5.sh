#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
python tools/energy_layers.py --out $O/energy_layers.json > $O/energy_layers.txt 2>&1
tail -8 $O/energy_layers.txt
python -m pytest tests/test_gpu_model.py tests/test_weights_io.py tests/test_gpu_train_step.py -q -x -m gpu -k "not full_size" > $O/pytest_call5.txt 2>&1
tail -3 $O/pytest_call5.txt
python tools/train_fullsize_parity.py --tag r05 --out $O/train_parity.txt > $O/train_parity.log 2>&1
grep -n "locked\|LeakyReLU elements" $O/train_parity.txt; tail -3 $O/train_parity.log
