"""Config (5) at its own workload: R50vd 608x608, 8 images, freeze_at = 5, DropBlock masks shared -- the HIP training step against
the fp32 and the float64 training oracle (tests/train_parity_util.py).  Writes profiles/<tag>_train_parity.txt.

    python tools/train_fullsize_parity.py [--tag r03] [--quick]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', default='r03')
    ap.add_argument('--out', default=None)
    ap.add_argument('--quick', action='store_true', help='only the 320-px cases')
    a = ap.parse_args()
    from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
    import train_parity_util as tp
    cases = [(PPYOLO_2x_Config, 320, 4, 5, True), (PPYOLO_2x_Config, 320, 4, 3, False), (PPYOLO_r18vd_Config, 416, 8, 5, True)]
    if not a.quick:
        cases.insert(0, (PPYOLO_2x_Config, 608, 8, 5, True))
    lines = []
    for cfgc, S, N, fa, drop in cases:
        res = tp.three_way(cfgc, S, N, fa, drop)
        lines.append(cfgc.__name__)
        lines += tp.summarize(res)
        print('\n'.join(lines[-12:]), flush=True)
    out = a.out or os.path.join(ROOT, 'gpurun_out', '%s_train_parity.txt' % a.tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
