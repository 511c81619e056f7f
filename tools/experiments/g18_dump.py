#!/usr/bin/env python
"""GPU box: run the HIP path on the g18 workloads (R50vd-608 / r18vd-416, batch 8, both im_size sets) and dump what a
box-error post-mortem needs -> gpurun_out/r04_g18_dump.npz: rows, keep indices, and for ONE image the raw head outputs and
the decoded boxes (so that the decode arithmetic can be separated from the network's logit noise offline)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')]
from conftest import build_model          # noqa: E402
from config import PPYOLO_2x_Config        # noqa: E402
from ppyolo_hip import synth               # noqa: E402

IMG = int(os.environ.get('G18_IMG', '7'))
out = {}
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g18_r50vd_608.npz'))
for math in ('f16x2', 'fp32'):
    os.environ['PPYOLO_HIP_MATH'] = math
    model, _ = build_model(PPYOLO_2x_Config(), 0, 'cuda')
    x = synth.synth_images(8, 608).cuda()
    for k in ('a', 'b'):
        ims = torch.from_numpy(g['im_size_' + k]).cuda()
        dets, cnt, keep = model.forward_padded(x, ims)
        torch.cuda.synchronize()
        out['%s_%s_dets' % (math, k)] = dets.cpu().numpy()
        out['%s_%s_cnt' % (math, k)] = cnt.cpu().numpy()
        out['%s_%s_keep' % (math, k)] = keep.cpu().numpy()
        ex = model._plans.executor(x)
        out['%s_%s_boxes%d' % (math, k, IMG)] = ex.boxes[IMG].cpu().numpy()
    for lv, a in enumerate(ex.plan.head_outs):
        out['%s_head%d_img%d' % (math, lv, IMG)] = ex.view(a).dense().permute(0, 3, 1, 2)[IMG].cpu().numpy()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'r04_g18_dump.npz'), **out)
print('dumped', {k: v.shape for k, v in out.items()})
