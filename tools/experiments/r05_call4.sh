#!/bin/bash
O=gpurun_out/r05
mkdir -p $O
python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "cu_masked or in_flight" > $O/pytest_masked.txt 2>&1
tail -3 $O/pytest_masked.txt
python tools/energy_layers.py --out $O/energy_layers.json > $O/energy_layers.txt 2>&1
tail -12 $O/energy_layers.txt
python tools/train_fullsize_parity.py --tag r05 --out $O/train_parity.txt > $O/train_parity.log 2>&1
grep -n "worst\|bias-gradient" $O/train_parity.txt
