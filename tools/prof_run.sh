#!/bin/bash
# Profile bench.py with rocprofv3 on the GPU box: kernel-trace stats + separate PMC passes
# (never combined with sys/runtime traces).  usage: tools/prof_run.sh <tag> [bench args]
TAG=${1:-r01}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
rm -rf $SCR; mkdir -p $OUT $SCR
export TMPDIR=/tmp
STEPS=5
BENCH="python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-alt-math --no-host-input --no-pmc --no-kernel-trace --no-batch-scaling --no-worst-case --no-graph --in-flight 1 --min-seconds 0 $*"   # (one lane: the per-kernel durations bench.py reports are solo durations)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
# (counter passes profile the clean --pmc-child run -- plan passes only -- so that per-pass sums are not inflated by the side legs'
#  extra launches of single layers; bench.py's own PMC leg does the same)
run_pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $SCR/$name -o pmc -- $BENCH --pmc-child > $OUT/$name.log 2>&1; }
run_pmc pmc_sq SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pmc pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM
cd $REPO
python tools/prof_summarize.py $SCR/trace $OUT/trace.txt $((STEPS+2)) > /dev/null 2>$OUT/trace.err || true
for n in pmc_sq pmc_fetch pmc_write pmc_lds; do      # (--pmc-child: 1 pass while the plan is built + 3 more)
  python tools/prof_summarize.py $SCR/$n $OUT/$n.txt 4 > /dev/null 2>$OUT/$n.err || true
done
tail -3 $OUT/trace.log
ls -la $OUT; du -sh $SCR
