"""Round 6, call d2 follow-up: the permuted-batch leg of test_full_size_batch_properties[PPYOLO_2x_Config-608] -- image 1 at batch
position 6 lands 116 px from its row at position 1.  A real difference or two near-tied rows trading places?  Prints the rows that differ.
usage: python tools/experiments/r06_perm_rows.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tools', 'experiments')):
    sys.path.insert(0, p)
from ppyolo_hip import synth      # noqa: E402
from r06_kp2_bisect import build      # noqa: E402


def main():
    model = build()
    N, S = 8, 608
    x = synth.synth_images(N, S)
    ims = torch.tensor([[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]] * 2)
    d0, c0, k0 = [t.cpu() for t in model.forward_padded(x.cuda(), ims.cuda())]
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4])
    d1, c1, k1 = [t.cpu() for t in model.forward_padded(x[perm].cuda(), ims[perm].cuda())]
    for j, i in enumerate(perm.tolist()):
        a, b = d0[i][:int(c0[i])], d1[j][:int(c1[j])]
        ka, kb = k0[i][:int(c0[i])], k1[j][:int(c1[j])]
        same_set = sorted(ka.tolist()) == sorted(kb.tolist())
        bad = [r for r in range(min(len(a), len(b))) if int(ka[r]) != int(kb[r])]
        print('image %d (position %d): %d / %d rows, keep sets equal: %s, rows with another keep index: %s' % (i, j, len(a), len(b), same_set, bad))
        for r in bad:
            print('   row %3d  original: keep %8d label %2d score %.9f box %s' % (r, int(ka[r]), int(a[r, 0]), float(a[r, 1]), [round(float(v), 3) for v in a[r, 2:]]))
            print('            permuted: keep %8d label %2d score %.9f box %s' % (int(kb[r]), int(b[r, 0]), float(b[r, 1]), [round(float(v), 3) for v in b[r, 2:]]))
        if not bad:
            print('   max box gap %.3e, max score gap %.3e' % (float((a[:, 2:] - b[:, 2:]).abs().max()), float((a[:, 1] - b[:, 1]).abs().max())))


if __name__ == '__main__':
    main()
