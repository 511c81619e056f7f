#!/bin/bash
# effective shader clock during the conv kernel = GRBM_GUI_ACTIVE (per XCD) / kernel duration
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/clk -o c -- python /root/repo/tools/conv_bench.py $1 $2 $3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/clk/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.OrderedDict()
for r in rows:
    if 'conv_igemm' not in r['Kernel_Name']: continue
    e = d.setdefault(r['Dispatch_Id'], {'dur': float(r['End_Timestamp']) - float(r['Start_Timestamp'])})
    e[r['Counter_Name']] = float(r['Counter_Value'])
vals = list(d.values())[-20:]
for v in vals[-5:]:
    clk = v['GRBM_GUI_ACTIVE'] / 8 / v['dur']      # cycles per ns = GHz
    print('dur %.1f us  GRBM/8 %.0f cyc  -> %.2f GHz   MFMA busy %.1f %% of SIMD-cycles' % (
        v['dur'] / 1e3, v['GRBM_GUI_ACTIVE'] / 8, clk, 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8)))
PY
