#!/usr/bin/env python
"""Soak test of the training step's second stream (train.py:_wgrad) and of the layers that run without raw tensors: the same
batch stepped `rounds` times from the same initial weights in two fresh TrainStep objects -- one with every round-4 overlap /
fusion switched off, one with the defaults -- must give the same loss sequence where the kernels are the same (bit for bit with
PPYOLO_HIP_TRAIN_BN_EPILOGUE=1; the default also moves the C = 128 layers to another kernel) and finite losses throughout; the
allocator gets the chance to hand a weight-gradient operand out again before the side stream has read it hundreds of times.
usage: train_soak.py [rounds] [size] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')]
import torch  # noqa: E402
from conftest import build_model  # noqa: E402
from config import PPYOLO_2x_Config  # noqa: E402
from ppyolo_hip import synth  # noqa: E402
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth  # noqa: E402


def run(rounds, S, N, env):
    os.environ.update(env)
    from ppyolo_hip.train import TrainStep
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg, 0, 'cuda')
    hc = cfg.head
    x = synth.synth_images(N, S, seed=77).cuda()
    bb, cc, ss = synth_ground_truth(N, 50)
    targets = [torch.from_numpy(t).cuda() for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
    gt = torch.from_numpy(bb).cuda()
    ts = TrainStep(model, cfg)
    out = []
    for i in range(rounds):
        out.append(ts.step(x, gt, targets, 1e-4))
        if i % 7 == 3:          # churn the caching allocator between steps
            junk = [torch.empty(int(3e6) * (k + 1), device='cuda') for k in range(4)]
            del junk
    torch.cuda.synchronize()
    return torch.stack(out).sum(1).cpu()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    a = run(rounds, S, N, {'PPYOLO_HIP_TRAIN_WGRAD_STREAM': '0', 'PPYOLO_HIP_TRAIN_BN_EPILOGUE': '0'})
    b = run(rounds, S, N, {'PPYOLO_HIP_TRAIN_WGRAD_STREAM': '1', 'PPYOLO_HIP_TRAIN_BN_EPILOGUE': '1'})
    c = run(rounds, S, N, {'PPYOLO_HIP_TRAIN_WGRAD_STREAM': '1', 'PPYOLO_HIP_TRAIN_BN_EPILOGUE': '2'})
    d = run(rounds, S, N, {'PPYOLO_HIP_TRAIN_WGRAD_STREAM': '0', 'PPYOLO_HIP_TRAIN_BN_EPILOGUE': '0', 'PPYOLO_HIP_TRAIN_MATH': 'bf16x3'})
    os.environ.pop('PPYOLO_HIP_TRAIN_MATH')
    print('training soak: %d steps of %d images %dx%d, loss %.4f -> %.4f' % (rounds, N, S, S, float(a[0]), float(a[-1])))
    print('  in-line / two-tensor form vs second stream + BatchNorm epilogue (same kernels): %s (%d of %d losses differ)'
          % ('EQUAL' if torch.equal(a, b) else 'DIFFERENT', int((a != b).sum()), rounds))
    print('  defaults (C = 128 layers on the streaming kernel too): finite %s, last loss %.4f (%.2e relative to the in-line form)'
          % (bool(torch.isfinite(c).all()), float(c[-1]), abs(float(c[-1] - a[-1])) / float(a[-1])))
    marks = [m for m in (0, 1, 4, 9, 24, 49, 99, rounds - 1) if m < rounds]
    print('  how two fp32-grade evaluations of the same step drift apart (training-mode BatchNorm + SGD amplify rounding): relative loss difference at step')
    print('    ' + '  '.join('%d' % (m + 1) for m in marks))
    print('    defaults vs in-line:           ' + '  '.join('%.1e' % (abs(float(c[m] - a[m])) / float(a[m])) for m in marks))
    print('    bf16x3 vs f16x2 (both in-line): ' + '  '.join('%.1e' % (abs(float(d[m] - a[m])) / float(a[m])) for m in marks))
    ok = torch.equal(a, b) and bool(torch.isfinite(c).all()) and bool(torch.isfinite(a).all())
    print('PASS' if ok else 'FAIL')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
