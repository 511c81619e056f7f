#!/bin/bash
# Kernel-trace of the hipGraph replay: busy time vs wall time per step (inter-kernel gaps).  usage: tools/graph_gaps.sh [bench args]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gg && rocprofv3 --kernel-trace --output-format csv -d /tmp/gg -o t -- python /root/repo/bench.py --no-cpu-baseline --no-alt-math --no-host-input --min-seconds 0 --steps 20 --warmup 5 "$@" > /tmp/gg.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/gg/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# take the last 20 "steps": find repeating pattern by the nms_finish kernel as step delimiter
ends = [i for i, r in enumerate(rows) if 'nms_finish' in r['Kernel_Name']]
if len(ends) > 12:
    a, b = ends[-11], ends[-1]
    seg = rows[a + 1:b + 1]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    wall = int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])
    n = 10
    gaps = [int(seg[i + 1]['Start_Timestamp']) - int(seg[i]['End_Timestamp']) for i in range(len(seg) - 1)]
    print('per step: %d kernels, busy %.3f ms, wall %.3f ms, idle %.3f ms (%.1f%%), mean gap %.2f us, max gap %.1f us' % (
        len(seg) // n, busy / n / 1e6, wall / n / 1e6, (wall - busy) / n / 1e6, 100.0 * (wall - busy) / wall,
        sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3))
PY
