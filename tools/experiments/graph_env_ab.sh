#!/bin/bash
# A/B of HIP-runtime environment knobs on the one-lane and two-lane step (hipGraph replay).  Every run under its own timeout, one
# line per run appended to gpurun_out/graph_env_ab.txt (a knob that hangs the replay must not take the others with it).
OUT=gpurun_out/graph_env_ab.txt; mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; if env "$@" timeout 150 python bench.py --no-cpu-baseline --no-alt-math --no-host-input --steps 50 > /tmp/ab.json 2>/tmp/ab.err; then python -c "
import json;d=json.load(open('/tmp/ab.json'));print('%-42s two lanes %.1f (sustained %.1f)  one lane %.1f' % ('$label',d['value'],d['sustained']['value'],d['one_batch_at_a_time']['value']))" >> $OUT; else echo "$label: failed / timed out (rc $?)" >> $OUT; fi; tail -1 $OUT; }
run "default" X=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "default (again)" X=1
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1" DEBUG_HIP_GRAPH_BATCH_SIZE=1
run "DEBUG_HIP_GRAPH_BATCH_SIZE=256" DEBUG_HIP_GRAPH_BATCH_SIZE=256
