#!/usr/bin/env python
"""Parity at the headline configuration, measured: EVERY image of a full-size batch (R50vd-608 bs 8 by default), in all three
PPYOLO_HIP_MATH modes, against (a) the fp32 CPU oracle = the reference's arithmetic and (b) a float64 run of the oracle =
the exact answer.  Rows are matched by Matrix-NMS keep index (candidate = box*80 + class), so a pair of near-tied rows that
swaps places is still compared detection by detection.  Test infrastructure (imports oracle/).  On the GPU box:

    python tools/fullsize_parity.py r50 608 8 > profiles/r02_fullsize_parity.txt

The table this prints is what tests/test_gpu_model.py::test_full_size_parity_all_images asserts against.
"""
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

IMS = [[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]]


def match_rows(a, ka, b, kb):
    """rows of `a` and `b` ([K,6] each, keep indices ka / kb) that are the same detection -> (ia, ib) index lists"""
    pos = {int(k): j for j, k in enumerate(kb)}
    ia, ib = [], []
    for i, k in enumerate(ka):
        if int(k) in pos:
            ia.append(i)
            ib.append(pos[int(k)])
    return ia, ib


def compare(a, ka, b, kb):
    """-> dict(n_a, n_b, matched, order_same, score, box, box_rel) for two detection sets of one image"""
    if a[0, 0] < 0 or b[0, 0] < 0:
        both = bool(a[0, 0] < 0 and b[0, 0] < 0)
        return dict(n_a=0 if a[0, 0] < 0 else a.shape[0], n_b=0 if b[0, 0] < 0 else b.shape[0], matched=0,
                    order_same=both, score=0.0, box=0.0, box_rel=0.0, labels=both)
    ia, ib = match_rows(a, ka, b, kb)
    A, B = a[ia].double(), b[ib].double()
    size = torch.maximum((B[:, 4] - B[:, 2]).abs(), (B[:, 5] - B[:, 3]).abs()).clamp_min(1.0)
    d = (A[:, 2:] - B[:, 2:]).abs()
    return dict(n_a=a.shape[0], n_b=b.shape[0], matched=len(ia), order_same=(ia == ib and a.shape[0] == b.shape[0]),
                score=float((A[:, 1] - B[:, 1]).abs().max()) if ia else 0.0, box=float(d.max()) if ia else 0.0,
                box_rel=float((d.max(dim=1).values / size).max()) if ia else 0.0, labels=bool(torch.equal(A[:, 0], B[:, 0])))


def hip_run(cfg, x, ims, mode):
    os.environ['PPYOLO_HIP_MATH'] = mode
    model, sd = build_model(cfg, 0, 'cuda')
    dets, cnt, keep = model.forward_padded(x.cuda(), ims.cuda())
    torch.cuda.synchronize()
    preds, keeps = [], []
    for i in range(x.shape[0]):
        k = max(int(cnt[i]), 1)
        preds.append(dets[i, :k].cpu().clone())
        keeps.append(keep[i, :k].cpu().clone())
    ex = model._plans.executor(x.cuda())
    heads = [ex.view(a).dense().permute(0, 3, 1, 2).cpu().double() for a in ex.plan.head_outs]
    return sd, preds, keeps, heads


def main():
    which, S, N = (sys.argv[1:] + ['r50', '608', '8'])[:3]
    S, N = int(S), int(N)
    cfg = {'r50': PPYOLO_2x_Config, 'r18': PPYOLO_r18vd_Config}[which]()
    x = synth.synth_images(N, S)
    ims = torch.tensor((IMS * ((N + 3) // 4))[:N])
    _, sd = build_model(cfg, 0, 'cpu')
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    def oracle(sdx, xx, imx):           # one forward: raw head outputs + detections with keep indices
        _, outs = orc.backbone_and_head(sdx, cfg, xx)
        with torch.no_grad():
            boxes, scores = orc.decode_all(outs, cfg.head, imx)
            nms = dict(cfg.nms_cfg)
            nms.pop('nms_type')
            return outs, [orc.matrix_nms(boxes[i], scores[i], return_index=True, **nms) for i in range(boxes.shape[0])]
    o32, r32 = oracle(sd, x, ims)
    o64, r64 = oracle(sd64, x.double(), ims.double())
    print('# %s %dx%d batch %d, image sizes %s; box errors in px of the ORIGINAL image; box_rel = box error / max(box side, 1 px)'
          % (which, S, S, N, [tuple(int(v) for v in r) for r in ims.tolist()]))
    print('# reference fp32 (CPU oracle) vs float64 oracle:')
    ref_img = []
    for i in range(N):
        c = compare(r32[i][0], r32[i][1], r64[i][0], r64[i][1])
        ref_img.append(c)
        print('ref32-f64  image %d: rows %3d/%3d matched %3d order_same %-5s labels %-5s score %.3e box %.3e px box_rel %.3e'
              % (i, c['n_a'], c['n_b'], c['matched'], c['order_same'], c['labels'], c['score'], c['box'], c['box_rel']))
    for lv in range(len(o32)):
        e = (o32[lv].double() - o64[lv])
        print('ref32-f64  head level %d: rms %.3e max %.3e' % (lv, e.pow(2).mean().sqrt(), e.abs().max()))
    for mode in ('f16x2', 'bf16x3', 'fp32'):
        _, preds, keeps, heads = hip_run(cfg, x, ims, mode)
        print('# PPYOLO_HIP_MATH=%s' % mode)
        worst = dict(s32=0.0, b32=0.0, r32=0.0, s64=0.0, b64=0.0, r64=0.0)
        for i in range(N):
            c32 = compare(preds[i], keeps[i], r32[i][0], r32[i][1])
            c64 = compare(preds[i], keeps[i], r64[i][0], r64[i][1])
            for tag, c in (('hip-ref32', c32), ('hip-f64  ', c64)):
                print('%s image %d: rows %3d/%3d matched %3d order_same %-5s labels %-5s score %.3e box %.3e px box_rel %.3e'
                      % (tag, i, c['n_a'], c['n_b'], c['matched'], c['order_same'], c['labels'], c['score'], c['box'], c['box_rel']))
            worst['s32'] = max(worst['s32'], c32['score']); worst['b32'] = max(worst['b32'], c32['box']); worst['r32'] = max(worst['r32'], c32['box_rel'])
            worst['s64'] = max(worst['s64'], c64['score']); worst['b64'] = max(worst['b64'], c64['box']); worst['r64'] = max(worst['r64'], c64['box_rel'])
        for lv in range(len(heads)):
            e = heads[lv] - o64[lv]
            e32 = heads[lv] - o32[lv].double()
            er = o32[lv].double() - o64[lv]
            print('%-6s head level %d: |hip-f64| rms %.3e max %.3e   |ref32-f64| rms %.3e max %.3e   |hip-ref32| rms %.3e max %.3e'
                  % (mode, lv, e.pow(2).mean().sqrt(), e.abs().max(), er.pow(2).mean().sqrt(), er.abs().max(),
                     e32.pow(2).mean().sqrt(), e32.abs().max()))
        rs = max(c['score'] for c in ref_img)
        rb = max(c['box'] for c in ref_img)
        rr = max(c['box_rel'] for c in ref_img)
        print('SUMMARY %-6s max over %d images: hip-ref32 score %.3e box %.3e px (rel %.3e) | hip-f64 score %.3e box %.3e px (rel %.3e) '
              '| ref32-f64 score %.3e box %.3e px (rel %.3e)' % (mode, N, worst['s32'], worst['b32'], worst['r32'], worst['s64'],
                                                                worst['b64'], worst['r64'], rs, rb, rr))


if __name__ == '__main__':
    main()
