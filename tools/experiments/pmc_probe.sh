#!/bin/bash
# usage: pmc_probe.sh "<shape>" <cfg> <split> COUNTER...   -> per-dispatch mean of each counter for the conv kernel
export TMPDIR=/tmp
SH=$1; CFG=$2; SP=$3; shift 3
cd /tmp && rm -rf /tmp/pmcp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcp -o c -- python /root/repo/tools/conv_bench.py $SH $CFG $SP > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pmcp/**/*counter_collection.csv', recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'conv_igemm' not in r['Kernel_Name']: continue
    e = d.setdefault(r['Dispatch_Id'], {'dur_us': (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3})
    e[r['Counter_Name']] = float(r['Counter_Value'])
vals = list(d.values())[-10:]
keys = sorted(vals[0].keys())
print('  '.join('%s=%.4g' % (k, sum(v[k] for v in vals) / len(vals)) for k in keys))
PY
