#!/bin/bash
# rocprofv3 on the TRAINING step (bench.py --train, R50vd-608, 8 images, freeze_at 5): kernel-trace stats + separate PMC passes
# for the HBM-side traffic (never combined with sys / runtime traces).  usage: tools/prof_train.sh <tag>
# -> gpurun_out/prof_train_<tag>/{trace,pmc_fetch,pmc_write}.txt and <tag>_train_pmc_traffic.json (copy into profiles/)
TAG=${1:-r03}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/prof_train_$TAG
SCR=/tmp/prof_train_$TAG
rm -rf $SCR; mkdir -p $OUT $SCR
export TMPDIR=/tmp
STEPS=5
BENCH="python $REPO/bench.py --train --steps $STEPS --warmup 2 --min-seconds 0 --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
run_pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $SCR/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1; }
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
cd $REPO
for n in trace pmc_fetch pmc_write; do
  python tools/prof_summarize.py $SCR/$n $OUT/$n.txt $((STEPS+2)) > /dev/null 2>$OUT/$n.err || true
done
python - "$OUT" "$TAG" $((STEPS+2)) <<'PY'
import json, re, sys, time
out, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
def table(path, counter):
    res, cols = {}, None
    for line in open(path):
        if line.startswith('kernel ') and counter in line:
            cols = line.split()[2:]
            continue
        if cols is None:
            continue
        if line.startswith('--') or not line.strip():
            break
        m = re.match(r'(.{92}) +(\d+) (.*)$', line.rstrip('\n'))
        if m:
            res[m.group(1).strip()] = (int(m.group(2)), float(m.group(3).split()[cols.index(counter)]))
    return res
fetch, write = table(out + '/pmc_fetch.txt', 'FETCH_SIZE'), table(out + '/pmc_write.txt', 'WRITE_SIZE')
conv = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_wgrad', 'dcn_fused', 'splitk_reduce', 'wgrad_combine')
is_conv = lambda k: any(n in k for n in conv)
f_kb = sum(v for k, (_, v) in fetch.items() if is_conv(k)) / steps
w_kb = sum(v for k, (_, v) in write.items() if is_conv(k)) / steps
launches = sum(c for k, (c, _) in fetch.items() if is_conv(k) and 'splitk_reduce' not in k and 'wgrad_combine' not in k) / steps
all_f = sum(v for _, v in fetch.values()) / steps
all_w = sum(v for _, v in write.values()) / steps
rec = dict(source='gpurun_out/prof_train_%s/pmc_fetch.txt, pmc_write.txt (rocprofv3 --pmc, separate passes; bench.py --train)' % tag,
           measured=time.strftime('%Y-%m-%d') + ' (' + tag + ')', steps_profiled=steps, conv_launches_per_step=launches,
           fetch_size_kb_per_step=f_kb, write_size_kb_per_step=w_kb, gfx950_fetch_correction=2.0,
           hbm_bytes_per_step=(2.0 * f_kb + w_kb) * 1024.0, whole_step_hbm_bytes=(2.0 * all_f + all_w) * 1024.0,
           note='convolution launches of the training step: forward (frozen backbone + head), data gradients, weight gradients (+ their '
                'split-K / slice combines); FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); whole_step_hbm_bytes = every kernel')
json.dump(rec, open(out + '/%s_train_pmc_traffic.json' % tag, 'w'), indent=1)
print(json.dumps(rec, indent=1))
PY
tail -3 $OUT/trace.log
ls -la $OUT
