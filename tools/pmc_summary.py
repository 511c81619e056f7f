#!/usr/bin/env python
"""Counter-based figures of the north star from the rocprofv3 summaries of tools/prof_run.sh (profiles/<tag>_pmc_sq.txt, _pmc_fetch.txt,
_pmc_write.txt, _kernel_trace_stats.txt) -> profiles/<tag>_pmc_summary.txt:
  * MFMA utilisation of the convolution kernels = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs).  SQ_VALU_MFMA_BUSY_CYCLES
    counts 32 per v_mfma_f32_32x32x16 wave-instruction over all SIMDs (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the
    8 XCDs, so elapsed cycles = GRBM_GUI_ACTIVE / 8 -- at the clock the board actually ran, not the nominal 2.4 GHz;
  * achieved HBM-side GB/s of the decode / Matrix-NMS kernels = (2 x FETCH_SIZE + WRITE_SIZE) KB per call / average duration
    (gfx950: FETCH_SIZE reports half of a 16 B/lane streaming read).
usage: pmc_summary.py <tag>"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path, first_col='kernel'):
    """[(kernel, calls, {column: value})] of the first aggregated table of a prof_summarize.py file."""
    rows, cols = [], None
    for line in open(path):
        if line.startswith(first_col + ' ') and cols is None:
            cols = line.split()[2:]
            continue
        if cols is None:
            continue
        if line.startswith('--') or not line.strip():
            break
        m = re.match(r'(.{92}) +(\d+) (.*)$', line.rstrip('\n'))
        if m:
            vals = m.group(3).split()
            rows.append((m.group(1).strip(), int(m.group(2)), dict(zip(cols, [float(v) for v in vals]))))
    return rows


def durations(path):
    """{kernel: (calls, avg_us)} from the 'kernel calls total_us avg_us pct' table."""
    out, on = {}, False
    for line in open(path):
        if line.startswith('kernel ') and 'avg_us' in line:
            on = True
            continue
        if on:
            m = re.match(r'(.{92}) +(\d+) +([\d.]+) +([\d.]+) +', line)
            if not m:
                if not line.strip():
                    on = False
                continue
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
    p = lambda n: os.path.join(ROOT, 'profiles', '%s_%s.txt' % (tag, n))
    sq = table(p('pmc_sq'))
    conv = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_b2b', 'dcn_fused')
    lines = ['MFMA utilisation from counters (profiles/%s_pmc_sq.txt): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)' % tag,
             '%-92s %6s %12s %9s' % ('kernel', 'calls', 'avg cycles', 'MFMA busy')]
    tb = tc = 0.0
    for k, calls, v in sq:
        if not any(c in k for c in conv) or v.get('GRBM_GUI_ACTIVE', 0) <= 0:
            continue
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0
        tb += v['SQ_VALU_MFMA_BUSY_CYCLES']
        tc += cyc
        lines.append('%-92s %6d %12.0f %8.1f%%' % (k, calls, cyc / calls, 100.0 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)))
    lines.append('%-92s %6s %12s %8.1f%%   <- all convolution / DCN launches of a step, time-weighted' % ('conv path', '', '', 100.0 * tb / (tc * 1024.0)))
    lines.append('(x 3 MFMA products per fp32 multiply-add in the f16x2 scheme: the share of the matrix pipe\'s cycles spent on useful fp32 work is a third of this)')
    lines.append('')
    fetch = {k: (c, v) for k, c, v in table(p('pmc_fetch'))}
    write = {k: (c, v) for k, c, v in table(p('pmc_write'))}
    # durations of the same kernels from the counter pass's own kernel trace (second table of the pmc file)
    dur = durations(p('pmc_fetch'))
    lines.append('achieved HBM-side bandwidth of the decode / Matrix-NMS kernels from counters: (2 x FETCH_SIZE + WRITE_SIZE) / average duration')
    lines.append('%-92s %6s %10s %10s %10s' % ('kernel', 'calls', 'KB / call', 'avg us', 'GB/s'))
    for k in fetch:
        if not any(n in k for n in ('yolo_decode', 'nms_')):
            continue
        c, v = fetch[k]
        w = write.get(k, (c, {'WRITE_SIZE': 0.0}))[1].get('WRITE_SIZE', 0.0)
        kb = (2.0 * v['FETCH_SIZE'] + w) / c
        us = dur.get(k, (0, 0.0))[1]
        lines.append('%-92s %6d %10.1f %10.2f %10.1f' % (k, c, kb, us, kb * 1024.0 / (us * 1e-6) / 1e9 if us > 0 else 0.0))
    out = os.path.join(ROOT, 'profiles', '%s_pmc_summary.txt' % tag)
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
