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


def bias_gradient_probe(ts, res):
    """Round 5 (verdict r4 item 4): the output convolutions' bias gradients are the tensors furthest from float64 relative to the
    reference (whole step).  A bias gradient is the per-channel sum of d loss / d output over all pixels -- here the HIP path's OWN
    d loss / d output summed three ways: ppy_channel_sum_f32 (what the step stored), torch's fp32 sum, a float64 sum.  If the first
    is as close to the third as the second is, the summation is not the cause and the distance is inherited from dout itself."""
    import torch
    L = []
    grads = ts.grads()
    for i, o in enumerate(ts.outs):
        k = 'head.yolo_output_convs.%d.conv.bias' % i
        if k not in grads or getattr(o, 'g', None) is None:
            continue
        dout = o.g.dense_nchw()                                   # [N, K, H, W] fp32, the HIP path's own
        s64 = dout.double().sum(dim=(0, 2, 3))
        s32 = dout.sum(dim=(0, 2, 3))
        den = s64.norm().clamp_min(1e-300)
        e_hip = float((grads[k].double().reshape(-1) - s64).norm() / den)
        e_t32 = float((s32.double() - s64).norm() / den)
        cancel = float(dout.double().abs().sum(dim=(0, 2, 3)).norm() / den)
        L.append('  bias-gradient probe %s: ppy_channel_sum vs float64 sum of the SAME dout %.2e | torch fp32 sum %.2e | cancellation '
                 '(sum|dout| / |sum dout|) %.1f | this tensor vs the float64 oracle (whole step) %.2e, reference fp32 %.2e'
                 % (k, e_hip, e_t32, cancel, res['grads'][k][0], res['grads'][k][1]))
    return L


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
        res = tp.three_way(cfgc, S, N, fa, drop, want_model=True)
        ts = res.pop('_ts')
        res.pop('_model')
        lines.append(cfgc.__name__)
        lines += tp.summarize(res)
        lines += bias_gradient_probe(ts, res)
        print('\n'.join(lines[-12:]), flush=True)
    out = a.out or os.path.join(ROOT, 'gpurun_out', '%s_train_parity.txt' % a.tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
