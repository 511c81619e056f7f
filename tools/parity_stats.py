#!/usr/bin/env python
"""Print deviation statistics of the HIP path vs the fp32 oracle and vs an fp64 run of the
oracle ("exact" arithmetic), next to the fp32 oracle's own deviation from fp64.  Test
infrastructure (imports oracle/); run on the GPU box:  python tools/parity_stats.py r50 608 2"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from conftest import build_model  # noqa: E402
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config  # noqa: E402
from oracle import ppyolo_oracle as orc  # noqa: E402
from ppyolo_hip import synth  # noqa: E402


def main():
    which, S, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    cfg = {'r50': PPYOLO_2x_Config, 'r18': PPYOLO_r18vd_Config}[which]()
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, S)
    ims = torch.tensor([[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]] * 2)[:N]
    hip = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    r32 = orc.ppyolo_forward(sd, cfg, x, ims)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    r64 = orc.ppyolo_forward(sd64, cfg, x.double(), ims.double())
    # raw head outputs too
    from ppyolo_hip.runtime import build_plan
    ex = model._plans.executor(x.cuda())
    f32, o32 = orc.backbone_and_head(sd, cfg, x)
    f64, o64 = orc.backbone_and_head(sd64, cfg, x.double())
    for i, a in enumerate(ex.plan.head_outs):
        h = ex.view(a).dense().permute(0, 3, 1, 2).cpu().double()
        print('head out %d: |hip-f64| max %.3e rms %.3e   |ref32-f64| max %.3e rms %.3e   |hip-ref32| max %.3e'
              % (i, (h - o64[i]).abs().max(), (h - o64[i]).pow(2).mean().sqrt(), (o32[i].double() - o64[i]).abs().max(),
                 (o32[i].double() - o64[i]).pow(2).mean().sqrt(), (h - o32[i].double()).abs().max()))
    for i in range(N):
        a, b, c = hip[i].double(), r32[i].double(), r64[i]
        same = a.shape == b.shape == c.shape and torch.equal(a[:, 0], b[:, 0]) and torch.equal(b[:, 0], c[:, 0])
        print('image %d: rows %s/%s/%s labels-equal %s' % (i, a.shape[0], b.shape[0], c.shape[0], same))
        if same:
            for nm, sl in (('score', slice(1, 2)), ('box', slice(2, 6))):
                print('   %-5s |hip-ref32| %.3e  |hip-f64| %.3e  |ref32-f64| %.3e' % (
                    nm, (a[:, sl] - b[:, sl]).abs().max(), (a[:, sl] - c[:, sl]).abs().max(),
                    (b[:, sl] - c[:, sl]).abs().max()))


if __name__ == '__main__':
    main()
