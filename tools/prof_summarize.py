#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace / counter collection) into small text tables
that can be committed under profiles/.  usage: prof_summarize.py <dir> <out.txt> [steps]"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    if len(name) > 90:
        name = name[:87] + '...'
    return name


def main():
    d, out = sys.argv[1], sys.argv[2]
    lines = []
    for path in sorted(glob.glob(os.path.join(d, '**', '*.csv'), recursive=True)):
        base = os.path.basename(path)
        with open(path) as fh:
            rd = csv.DictReader(fh)
            rows = list(rd)
        if not rows:
            continue
        cols = rows[0].keys()
        lines.append('== %s  (%d rows)  columns: %s' % (os.path.relpath(path, d), len(rows), ','.join(cols)))
        if 'Counter_Name' in cols:
            agg = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(set)
            for r in rows:
                k = short(r['Kernel_Name'])
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                calls[k].add(r['Dispatch_Id'])
            names = sorted({c for v in agg.values() for c in v})
            lines.append('%-92s %7s ' % ('kernel', 'calls') + ' '.join('%22s' % n for n in names))
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
                lines.append('%-92s %7d ' % (k, len(calls[k])) + ' '.join('%22.6g' % v.get(n, 0) for n in names))
            # per-dispatch detail for the conv kernels of the LAST step (dispatch order = plan order)
            conv = [r for r in rows if 'conv_igemm' in r['Kernel_Name']]
            if conv:
                per = OrderedDict()
                for r in conv:
                    per.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
                    per[r['Dispatch_Id']]['_grid'] = r.get('Grid_Size', '')
                    per[r['Dispatch_Id']]['_name'] = short(r['Kernel_Name'])[18:60]
                ids = list(per.keys())
                n_per_step = len(ids) // max(1, int(sys.argv[3]) if len(sys.argv) > 3 else 7)
                lines.append('-- per-dispatch (last step, conv_igemm only): dispatch, template, grid, ' + ' '.join(names))
                for i in ids[-n_per_step:]:
                    v = per[i]
                    lines.append('%8s %-44s %10s ' % (i, v['_name'], v['_grid']) + ' '.join('%16.6g' % v.get(n, 0) for n in names))
        elif 'Start_Timestamp' in cols and 'Kernel_Name' in cols:
            tot = defaultdict(float)
            cnt = defaultdict(int)
            for r in rows:
                k = short(r['Kernel_Name'])
                tot[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                cnt[k] += 1
            all_ns = sum(tot.values())
            lines.append('%-92s %8s %12s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
            for k in sorted(tot, key=lambda k: -tot[k]):
                lines.append('%-92s %8d %12.1f %12.2f %6.2f%%' % (k, cnt[k], tot[k] / 1e3, tot[k] / cnt[k] / 1e3,
                                                                 100 * tot[k] / all_ns))
        else:
            for r in rows[:40]:
                lines.append('  ' + ' | '.join('%s' % v for v in r.values()))
        lines.append('')
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:200]))


if __name__ == '__main__':
    main()
