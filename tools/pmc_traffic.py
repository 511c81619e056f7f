#!/usr/bin/env python
"""Derive the per-step HBM traffic of the conv kernels from the two PMC summaries written by
tools/prof_run.sh (FETCH_SIZE pass, WRITE_SIZE pass) -> profiles/<tag>_pmc_traffic.json, which
bench.py quotes as roofline.traffic.   usage: pmc_traffic.py <tag>   (reads profiles/<tag>_pmc_*.txt)"""
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FETCH_CORRECTION = 2.0     # gfx950: FETCH_SIZE counts half of a 16 B/lane streaming read (MI355X_MICROARCH.md)


def table(path, counter):
    """{kernel: (calls, value)} from the aggregated table of a prof_summarize.py text file."""
    out, cols = {}, None
    for line in open(path):
        if line.startswith('kernel ') and counter in line:
            cols = line.split()[2:]
            continue
        if cols is None:
            continue
        if line.startswith('--') or not line.strip():
            break
        m = re.match(r'(.{92}) +(\d+) (.*)$', line.rstrip('\n'))
        if not m:
            continue
        vals = m.group(3).split()
        out[m.group(1).strip()] = (int(m.group(2)), float(vals[cols.index(counter)]))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    p = lambda n: os.path.join(ROOT, 'profiles', '%s_%s.txt' % (tag, n))
    fetch, write = table(p('pmc_fetch'), 'FETCH_SIZE'), table(p('pmc_write'), 'WRITE_SIZE')
    runs = [c for k, (c, _) in fetch.items() if 'stem_conv' in k][0]        # one stem launch per pass
    main_k = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_b2b', 'dcn_fused')      # the plan's conv / DCN ops: tiles (incl. specialised waves), streaming 1x1, stem patch, fused DCNv2 (as bench.py's own PMC leg counts them)
    is_conv = lambda k: any(n in k for n in main_k) or 'splitk_reduce' in k
    f_kb = sum(v for k, (_, v) in fetch.items() if is_conv(k)) / runs
    w_kb = sum(v for k, (_, v) in write.items() if is_conv(k)) / runs
    launches = sum(c for k, (c, _) in fetch.items() if any(n in k for n in main_k)) / runs
    rec = {
        'source': 'profiles/%s_pmc_fetch.txt, profiles/%s_pmc_write.txt (rocprofv3 --pmc, separate passes)' % (tag, tag),
        'measured': time.strftime('%Y-%m-%d') + ' (' + tag + ')',
        'conv_runs_profiled': runs,
        'conv_launches_per_step': launches,
        'fetch_size_kb_per_step': f_kb,
        'write_size_kb_per_step': w_kb,
        'gfx950_fetch_correction': FETCH_CORRECTION,
        'hbm_bytes_per_step': (FETCH_CORRECTION * f_kb + w_kb) * 1024.0,
        'note': 'FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of a 16 B/lane streaming read '
                '(MI355X_MICROARCH.md, HBM section) and the conv operand loads are 16 B/lane buffer_load..lds, hence x2. '
                'Counters sit on the fabric side of L2, so Infinity-Cache hits are included. Split-K partial sums '
                '(conv + combine kernel) are included.',
    }
    out = os.path.join(ROOT, 'profiles', '%s_pmc_traffic.json' % tag)
    with open(out, 'w') as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
