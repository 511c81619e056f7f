#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE implementation.

Build-container only: imports miemie2013/Pytorch-PPYOLO from /root/reference (which
never travels to the GPU box) with the recipe of SURVEY.md section 8c, feeds it this
repo's deterministic synthetic weights / inputs, and stores inputs + expected outputs
as small fixtures.  The fixtures are DATA (tensors); no reference source is stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [--only g5]

Every fixture whose result depends on a sort asserts that the sorted keys are
pairwise distinct (torch's CPU argsort is not stable; SURVEY.md section 7 hard part 1).
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth = _load('ppy_synth', os.path.join(ROOT, 'pytorch-ppyolo_amd', 'ppyolo_hip', 'synth.py'))

# --- reference import recipe ------------------------------------
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self          # reference model/head.py:43 hard-codes .cuda()
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config, select_backbone, select_head  # noqa: E402
from model.ppyolo import PPYOLO                                                          # noqa: E402
from model.custom_layers import Conv2dUnit, DCNv2, DCNv2_Slow, CoordConv, SPP           # noqa: E402
from model.head import yolo_box, get_iou_aware_score                                     # noqa: E402
from model.matrix_nms import matrix_nms                                                  # noqa: E402


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **conv)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def gen(seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return g


# ---------------------------------------------------------------------------------------
def g1_conv_units():
    """Conv2dUnit (reference model/custom_layers.py:65-253) at tiny shapes."""
    cases = [  # cin, cout, k, stride, act, bn, bias, N, H, W
        (32, 64, 1, 1, 'relu', 1, False, 2, 9, 11),
        (32, 32, 3, 1, 'leaky', 1, False, 2, 10, 7),
        (64, 32, 3, 2, 'relu', 1, False, 2, 11, 9),
        (64, 96, 3, 2, None, 1, False, 1, 12, 12),
        (32, 27, 3, 1, None, 0, True, 2, 6, 6),
        (64, 258, 1, 1, None, 0, True, 2, 5, 5),
        (96, 160, 1, 1, 'leaky', 1, False, 1, 19, 19),
        (3, 32, 3, 2, 'relu', 1, False, 2, 16, 16),
        (3, 32, 3, 2, 'relu', 1, False, 1, 15, 18),
    ]
    g = gen(101)
    out = {}
    for i, (ci, co, k, s, act, bn, bias, N, H, W) in enumerate(cases):
        m = Conv2dUnit(ci, co, k, stride=s, bias_attr=bias, bn=bn, act=act)
        m.eval()
        with torch.no_grad():
            m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * (2.0 / (ci * k * k)) ** 0.5)
            if bias:
                m.conv.bias.copy_(torch.randn(co, generator=g) * 0.5)
            if bn:
                m.bn.weight.copy_(torch.rand(co, generator=g) + 0.5)
                m.bn.bias.copy_(torch.randn(co, generator=g) * 0.2)
                m.bn.running_mean.copy_(torch.randn(co, generator=g) * 0.2)
                m.bn.running_var.copy_(torch.rand(co, generator=g) + 0.5)
            x = torch.randn(N, ci, H, W, generator=g)
            y = m(x)
        p = 'c%d_' % i
        out[p + 'meta'] = np.array([ci, co, k, s, {'relu': 1, 'leaky': 2, None: 0}[act], bn, int(bias)])
        out[p + 'x'] = x
        out[p + 'w'] = m.conv.weight
        if bias:
            out[p + 'b'] = m.conv.bias
        if bn:
            out[p + 'bn'] = torch.stack([m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var])
        out[p + 'y'] = y
    out['ncases'] = np.array(len(cases))
    save('g1_conv_units', **out)


def g2_dcn():
    """DCNv2 (reference model/custom_layers.py:486-677), cross-checked against the
    python-loop DCNv2_Slow (:346-482) which has explicit per-corner bounds tests."""
    g = gen(202)
    out = {}
    cases = [(32, 64, 1, 3, 9, 9), (32, 32, 2, 3, 10, 10), (64, 32, 1, 2, 7, 5)]
    for i, (ci, co, s, N, H, W) in enumerate(cases):
        m = DCNv2(ci, co, filter_size=3, stride=s, padding=1, bias_attr=False)
        m.eval()
        with torch.no_grad():
            # offsets: std ~3 px plus a few far-out-of-range taps
            m.conv_offset.weight.copy_(torch.randn(m.conv_offset.weight.shape, generator=g) * (3.0 / (ci * 9) ** 0.5))
            b = torch.randn(27, generator=g) * 1.0
            b[3] = 12.3
            b[10] = -11.7
            m.conv_offset.bias.copy_(b)
            m.dcn_weight.copy_(torch.randn(m.dcn_weight.shape, generator=g) * (1.0 / (ci * 9)) ** 0.5)
            x = torch.randn(N, ci, H, W, generator=g)
            om = m.conv_offset(x)
            y = m(x)
            slow = DCNv2_Slow(ci, co, filter_size=3, stride=s, padding=1, bias_attr=False)
            slow.load_state_dict(m.state_dict())
            ys = slow(x)
        err = (y - ys).abs().max().item()
        assert err < 5e-5, err
        p = 'd%d_' % i
        out[p + 'meta'] = np.array([ci, co, s])
        out[p + 'x'] = x
        out[p + 'w_off'] = m.conv_offset.weight
        out[p + 'b_off'] = m.conv_offset.bias
        out[p + 'w_dcn'] = m.dcn_weight
        out[p + 'offset_mask'] = om
        out[p + 'y'] = y
        out[p + 'y_slow'] = ys
    out['ncases'] = np.array(len(cases))
    save('g2_dcnv2', **out)


def g3_coord_spp():
    g = gen(303)
    out = {}
    x = torch.randn(2, 32, 5, 7, generator=g)
    out['coord_x'] = x
    out['coord_y'] = CoordConv(True)(x)
    for i, (N, C, H, W) in enumerate([(2, 32, 19, 19), (1, 32, 10, 10), (1, 64, 13, 13)]):
        x = torch.randn(N, C, H, W, generator=g)
        out['spp%d_x' % i] = x
        out['spp%d_y' % i] = SPP()(x)
    save('g3_coord_spp', **out)


def g4_decode():
    """get_iou_aware_score + yolo_box (reference model/head.py:21-141)."""
    g = gen(404)
    cfg = PPYOLO_2x_Config()
    anchors = np.array(cfg.head['anchors'], dtype=np.float32)
    out = {}
    im_size = torch.tensor([[480., 640.], [375., 500.], [1080., 1920.]])
    for i, (S, stride, mask, iou_aware) in enumerate([(5, 32, [6, 7, 8], True), (7, 16, [3, 4, 5], True),
                                                      (6, 8, [0, 1, 2], False)]):
        nch = 3 * (80 + (6 if iou_aware else 5))
        o = torch.randn(3, nch, S, S, generator=g) * 2.0
        # push some boxes across every image border so both clips fire
        o[:, :, 0, 0] += 1.5
        with torch.no_grad():
            t = get_iou_aware_score(o, 3, 80, 0.4) if iou_aware else o
            boxes, scores = yolo_box(t, anchors[mask], stride, 80, 1.05, im_size, True, 0.01)
        out['l%d_meta' % i] = np.array([S, stride, int(iou_aware)] + mask)
        out['l%d_out' % i] = o
        out['l%d_boxes' % i] = boxes
        out['l%d_scores' % i] = scores
    out['im_size'] = im_size
    out['anchors'] = anchors
    save('g4_decode', **out)


def _nms_case(boxes, scores, **kw):
    cfg = dict(score_threshold=0.01, post_threshold=0.01, nms_top_k=500, keep_top_k=100,
               use_gaussian=False, gaussian_sigma=2.)
    cfg.update(kw)
    # tie-freeness at the first sort site
    s = scores[scores > cfg['score_threshold']]
    assert len(torch.unique(s)) == len(s), 'score ties in NMS fixture'
    with torch.no_grad():
        pred = matrix_nms(boxes, scores, **cfg)
    if pred[0, 0] >= 0:
        assert len(torch.unique(pred[:, 1])) == pred.shape[0], 'ties after decay'
    return pred, cfg


def _rand_boxes(g, n, w=640., h=480., smin=8., smax=200.):
    cx = torch.rand(n, generator=g) * w
    cy = torch.rand(n, generator=g) * h
    bw = torch.rand(n, generator=g) * (smax - smin) + smin
    bh = torch.rand(n, generator=g) * (smax - smin) + smin
    b = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, w)
    b[:, 1::2] = b[:, 1::2].clamp(0, h)
    return b


def _distinct_scores(g, shape, lo, hi):
    n = int(np.prod(shape))
    # distinct by construction: a random permutation of an arithmetic grid, float32-exact
    grid = torch.linspace(lo, hi, n, dtype=torch.float64)
    s = grid[torch.randperm(n, generator=g)].float().reshape(shape)
    assert len(torch.unique(s)) == n
    return s


def g5_matrix_nms():
    """matrix_nms (reference model/matrix_nms.py:102-151) on synthetic tie-free data."""
    g = gen(505)
    out = {}
    cases = []
    # 0: many candidates (>> nms_top_k), 5 classes, clustered boxes so decay matters
    M, C = 900, 5
    base = _rand_boxes(g, 60)
    boxes = base[torch.randint(0, 60, (M,), generator=g)] + torch.randn(M, 4, generator=g) * 6.0
    scores = _distinct_scores(g, (M, C), 0.0005, 0.95)
    scores[torch.rand(M, C, generator=g) < 0.6] = 0.001
    cases.append((boxes, scores, {}))
    # 1: few candidates (< keep_top_k)
    boxes = _rand_boxes(g, 40)
    scores = _distinct_scores(g, (40, 3), 0.0, 0.03)
    cases.append((boxes, scores, {}))
    # 2: nothing above threshold -> sentinel
    boxes = _rand_boxes(g, 30)
    scores = _distinct_scores(g, (30, 4), 0.0, 0.0099)
    cases.append((boxes, scores, {}))
    # 3: exact duplicate boxes of the same class (1 - comp == 0 -> inf / NaN paths)
    boxes = _rand_boxes(g, 50)
    boxes[10] = boxes[3]
    boxes[20] = boxes[3]
    boxes[21] = boxes[7]
    scores = _distinct_scores(g, (50, 2), 0.02, 0.9)
    cases.append((boxes, scores, {}))
    # 4: zero-area boxes (0/0 IoU -> NaN poisons every decay -> sentinel)
    boxes = _rand_boxes(g, 30)
    boxes[5] = torch.tensor([100., 100., 100., 150.])
    boxes[9] = torch.tensor([300., 200., 300., 200.])
    scores = _distinct_scores(g, (30, 2), 0.02, 0.9)
    cases.append((boxes, scores, {}))
    # 5: one zero-area box only (NaN only where it meets another zero-area box: none)
    boxes = _rand_boxes(g, 30)
    boxes[5] = torch.tensor([100., 100., 100., 150.])
    scores = _distinct_scores(g, (30, 2), 0.02, 0.9)
    cases.append((boxes, scores, {}))
    # 6: gaussian kernel, smaller top-k
    M, C = 400, 3
    base = _rand_boxes(g, 25)
    boxes = base[torch.randint(0, 25, (M,), generator=g)] + torch.randn(M, 4, generator=g) * 4.0
    scores = _distinct_scores(g, (M, C), 0.001, 0.8)
    cases.append((boxes, scores, dict(use_gaussian=True, nms_top_k=200, keep_top_k=50)))
    # 7: heavy suppression: post_threshold removes most
    M, C = 300, 1
    base = _rand_boxes(g, 6)
    boxes = base[torch.randint(0, 6, (M,), generator=g)] + torch.randn(M, 4, generator=g) * 2.0
    scores = _distinct_scores(g, (M, C), 0.011, 0.3)
    cases.append((boxes, scores, dict(post_threshold=0.05)))
    # 8: 80 classes, PP-YOLO-sized box count
    M, C = 2535, 80
    base = _rand_boxes(g, 200)
    boxes = base[torch.randint(0, 200, (M,), generator=g)] + torch.randn(M, 4, generator=g) * 5.0
    scores = _distinct_scores(g, (M, C), 0.0, 0.012)
    hot = torch.rand(M, C, generator=g) < 0.004
    scores[hot] = _distinct_scores(g, (int(hot.sum()),), 0.02, 0.97)
    cases.append((boxes, scores, {}))
    for i, (b, s, kw) in enumerate(cases):
        pred, cfg = _nms_case(b, s, **kw)
        out['n%d_boxes' % i] = b
        out['n%d_scores' % i] = s
        out['n%d_pred' % i] = pred
        out['n%d_cfg' % i] = np.array([cfg['score_threshold'], cfg['post_threshold'], cfg['nms_top_k'],
                                       cfg['keep_top_k'], float(cfg['use_gaussian']), cfg['gaussian_sigma']],
                                      dtype=np.float64)
        print('   nms case %d -> %s rows (first label %.0f)' % (i, tuple(pred.shape), pred[0, 0]))
    out['ncases'] = np.array(len(cases))
    save('g5_matrix_nms', **out)


def build_ref(cfg, seed=0):
    bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
    hd = select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, seed=seed), strict=True)
    m.eval()
    hd.set_dropblock(is_test=True)
    return m, shapes


def _assert_tie_free(model, x, im_size, cfg):
    """Both sort sites of matrix_nms (reference model/matrix_nms.py:120, :140)."""
    with torch.no_grad():
        outs = model.head._get_outputs(model.backbone(x))
        from model.head import get_iou_aware_score as gi, yolo_box as yb
        bs, ss = [], []
        for i, o in enumerate(outs):
            if model.head.iou_aware:
                o = gi(o, 3, 80, model.head.iou_aware_factor)
            b, s = yb(o, model.head._anchors[model.head.anchor_masks[i]], model.head.downsample[i], 80,
                      model.head.scale_x_y, im_size, True, 0.01)
            bs.append(b)
            ss.append(s)
        scores = torch.cat(ss, 1)
    for n in range(scores.shape[0]):
        s = scores[n][scores[n] > 0.01]
        top = torch.sort(s, descending=True)[0][:520]
        assert len(torch.unique(top)) == len(top), 'tie among top candidates of image %d' % n
    return int((scores > 0.01).sum())


def g6_g7_models():
    out_shapes = {}
    for tag, C, S, N, seed in (('r18vd_64', PPYOLO_r18vd_Config, 64, 2, 0),
                               ('r50vd_96', PPYOLO_2x_Config, 96, 2, 0)):
        cfg = C()
        m, shapes = build_ref(cfg, seed)
        x = synth.synth_images(N, S, seed=1234)
        with torch.no_grad():
            feats = m.backbone(x)
            outs = m.head._get_outputs(feats)
        arrs = {'meta': np.array([S, N, seed, 1234])}
        for i, f in enumerate(feats):
            arrs['feat%d' % i] = f
        for i, o in enumerate(outs):
            arrs['out%d' % i] = o
        save('g6_' + tag, **arrs)
    for tag, C, S, N, seed in (('r18vd_320', PPYOLO_r18vd_Config, 320, 1, 0),
                               ('r50vd_160', PPYOLO_2x_Config, 160, 2, 0)):
        cfg = C()
        m, shapes = build_ref(cfg, seed)
        x = synth.synth_images(N, S, seed=1234)
        im_size = torch.tensor([[480., 640.], [375., 500.]])[:N]
        ncand = _assert_tie_free(m, x, im_size, cfg)
        with torch.no_grad():
            preds = m(x, im_size)
            outs = m.head._get_outputs(m.backbone(x))
        arrs = {'meta': np.array([S, N, seed, 1234]), 'im_size': im_size, 'ncand': np.array(ncand)}
        for i, p in enumerate(preds):
            assert len(torch.unique(p[:, 1])) == p.shape[0]
            arrs['pred%d' % i] = p
        for i, o in enumerate(outs):
            arrs['out%d' % i] = o
        save('g7_' + tag, **arrs)
        print('   %s: %d candidates, rows %s' % (tag, ncand, [tuple(p.shape) for p in preds]))


def g8_preprocess():
    """NormalizeImage + Permute of the reference's Decode.process_image (model/decode_np.py:136-137), run by the
    reference's own classes on uint8 images that are ALREADY S x S (the cv2.resize in front cannot run here: cv2 is not
    installed, a constants-only stub lets tools/transform.py import), plus the resize scale factors ResizeImage computes."""
    import types
    stub = types.ModuleType('cv2')
    for i, k in enumerate(('INTER_NEAREST', 'INTER_LINEAR', 'INTER_CUBIC', 'INTER_AREA', 'INTER_LANCZOS4')):
        setattr(stub, k, i)
    stub.COLOR_BGR2RGB = 4
    sys.modules.setdefault('cv2', stub)
    tr = _load('ref_transform', os.path.join(REF, 'tools', 'transform.py'))
    cfg = PPYOLO_2x_Config()
    norm = tr.NormalizeImage(**cfg.normalizeImage)
    perm = tr.Permute(**cfg.permute)
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(64, 64, 3)).astype(np.uint8)
    img[:4, :64, :] = np.arange(256, dtype=np.uint8).reshape(4, 64, 1)         # every grey level in every channel
    sample = perm(norm({'image': img.copy()}, None), None)
    out = sample['image']
    assert out.dtype == np.float32 and out.shape == (3, 64, 64)
    save('g8_preprocess', image=img, normalized_chw=out, mean=np.array(cfg.normalizeImage['mean']),
         std=np.array(cfg.normalizeImage['std']), to_rgb=np.array(int(cfg.decodeImage['to_rgb'])),
         interp=np.array(cfg.resizeImage['interp']), target_size=np.array(cfg.test_cfg['target_size']))


def _cv2_stub():
    import types
    stub = types.ModuleType('cv2')
    for i, k in enumerate(('INTER_NEAREST', 'INTER_LINEAR', 'INTER_CUBIC', 'INTER_AREA', 'INTER_LANCZOS4')):
        setattr(stub, k, i)
    stub.COLOR_BGR2RGB = 4
    sys.modules.setdefault('cv2', stub)


def g9_decode_harness():
    """The host harness on top of the model -- the reference's Decode.predict / detect_image / detect_batch
    (model/decode_np.py:41-96, :142-150) on numpy inputs, with the constants-only cv2 stub (nothing is drawn)."""
    _cv2_stub()
    from model.decode_np import Decode
    cfg = PPYOLO_r18vd_Config()
    m, _ = build_ref(cfg, 0)
    dec = Decode(m, ['c%d' % i for i in range(80)], False, cfg, for_test=True)
    x = synth.synth_images(2, 320, seed=1234).numpy()
    ims = np.array([[480, 640], [375, 500]], dtype=np.int32)
    with torch.no_grad():
        image, boxes, scores, classes = dec.detect_image(None, x[:1], ims[:1], draw_image=False)
        _, bb, ss, cc = dec.detect_batch([None, None], x, ims, draw_image=False)
    assert boxes.dtype == np.float32 and scores.dtype == np.float32 and classes.dtype == np.int32
    arrs = dict(meta=np.array([320, 2, 0, 1234]), im_size=ims, boxes=boxes, scores=scores, classes=classes)
    for i in range(2):
        arrs['b_boxes%d' % i], arrs['b_scores%d' % i], arrs['b_classes%d' % i] = bb[i], ss[i], cc[i]
    save('g9_decode_harness', **arrs)


def g10_coco_records():
    """The eval harness's on-disk record format: the reference's own writer, tools/cocotools.py:159-191
    (`multi_thread_write_json`), on detection arrays of the types `Decode.detect_batch` returns -- the g9 harness output
    plus hand-made rows on the rounding edges (x.x5 in float32, -0.0 from the x0 clip, sub-pixel and 4-digit boxes)."""
    import json
    import tempfile
    _cv2_stub()
    coco = _load('ref_cocotools', os.path.join(REF, 'tools', 'cocotools.py'))
    g9 = np.load(os.path.join(OUT, 'g9_decode_harness.npz'))
    edge = np.array([[-0.0, 0.05, 10.05, 20.15], [0.25, 0.35, 0.45, 0.55], [1234.5678, 987.65, 1919.95, 1079.949],
                     [3.14159, 2.71828, 3.14159, 2.71828], [100.04999, 100.05, 100.15, 100.25], [7.5, 8.5, 9.5, 10.5]],
                    dtype=np.float32)
    cases = [(g9['b_boxes0'], g9['b_scores0'], g9['b_classes0'], 139, '000000000139.jpg'),
             (g9['b_boxes1'], g9['b_scores1'], g9['b_classes1'], 285, '000000000285.jpg'),
             (edge, np.array([0.99999994, 0.5, 0.010000001, 1e-8, 0.123456789, 0.75], dtype=np.float32),
              np.array([0, 11, 79, 24, 60, 1], dtype=np.int32), 724, 'edge.case.png'),
             (np.array([]), np.array([]), np.array([]), 7, 'empty.jpg')]
    arrs = dict(ncases=np.array(len(cases)), clsid2catid=np.array([coco.clsid2catid[i] for i in range(80)]))
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, 'bbox'))
        for j, (b, sc, cl, im_id, name) in enumerate(cases):
            coco.multi_thread_write_json(0, [None], [b], [sc], [cl], [im_id], [name], coco.clsid2catid, False, td)
            text = open(os.path.join(td, 'bbox', name.split('.')[0] + '.json')).read()
            json.loads(text)
            arrs.update({'boxes%d' % j: b, 'scores%d' % j: sc, 'classes%d' % j: cl, 'im_id%d' % j: np.array(im_id),
                         'name%d' % j: np.frombuffer(name.encode(), dtype=np.uint8),
                         'json%d' % j: np.frombuffer(text.encode(), dtype=np.uint8)})
    save('g10_coco_records', **arrs)


def g11_state_dict_layout():
    """Checkpoint layout: key names (in order), shapes and dtypes of the REFERENCE modules' state_dict
    -- what `torch.save(ppyolo.state_dict(), ...)` (reference 1_ppyolo_2x_2pytorch.py:321) puts into a .pt file."""
    arrs = {}
    for tag, C in (('r18vd', PPYOLO_r18vd_Config), ('r50vd', PPYOLO_2x_Config)):
        m, _ = build_ref(C(), 0)
        sd = m.state_dict()
        keys = list(sd.keys())
        arrs[tag + '_keys'] = np.frombuffer(''.join(keys).encode(), dtype=np.uint8)
        arrs[tag + '_keylens'] = np.array([len(k) for k in keys])
        arrs[tag + '_shapes'] = np.array([d for k in keys for d in sd[k].shape], dtype=np.int64)
        arrs[tag + '_ranks'] = np.array([sd[k].dim() for k in keys])
        arrs[tag + '_dtypes'] = np.array([str(sd[k].dtype).ljust(16).encode() for k in keys], dtype='S16')
    save('g11_state_dict_layout', **arrs)


def build_ref_train(cfg, seed=0):
    """The model exactly as reference train.py:238-264 builds it: losses, Head(is_train=True), backbone.freeze(); left in
    nn.Module's default training mode (train.py never calls .eval() before the loop)."""
    from config import select_loss
    bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
    iou_loss = select_loss(cfg.iou_loss_type)(**cfg.iou_loss)
    iou_aware = select_loss(cfg.iou_aware_loss_type)(**cfg.iou_aware_loss) if cfg.head['iou_aware'] else None
    yolo_loss = select_loss(cfg.yolo_loss_type)(iou_loss=iou_loss, iou_aware_loss=iou_aware, **cfg.yolo_loss)
    hd = select_head(cfg.head_type)(yolo_loss=yolo_loss, is_train=True, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, seed=seed), strict=True)
    bb.freeze()
    return m


def synth_gt(N, S, cfg, seed):
    """Deterministic ground truth in the reader's format (normalised cx, cy, w, h; 50 slots per image, zero padded) and
    the YOLO targets made from it by the REFERENCE's own Gt2YoloTarget (tools/transform.py:1211-1316)."""
    _cv2_stub()
    from tools.transform import Gt2YoloTarget
    rng = np.random.RandomState(seed)
    samples = []
    for n in range(N):
        k = 3 + n
        gt_bbox = np.zeros((50, 4), np.float32)
        gt_class = np.zeros((50,), np.int32)
        gt_score = np.zeros((50,), np.float32)
        wh = rng.uniform(0.08, 0.7, size=(k, 2))
        c = rng.uniform(0.15, 0.85, size=(k, 2))
        gt_bbox[:k] = np.concatenate([c, wh], 1)
        gt_class[:k] = rng.randint(0, 80, size=k)
        gt_score[:k] = np.where(rng.rand(k) < 0.3, rng.uniform(0.3, 0.9, size=k), 1.0)       # mixup leaves scores < 1
        samples.append(dict(image=np.zeros((3, S, S), np.float32), gt_bbox=gt_bbox, gt_class=gt_class, gt_score=gt_score))
    op = Gt2YoloTarget(anchors=cfg.head['anchors'], anchor_masks=cfg.head['anchor_masks'],
                       downsample_ratios=cfg.head['downsample'], num_classes=80)
    samples = op(samples)
    L = len(cfg.head['anchor_masks'])
    targets = [np.stack([smp['target%d' % i] for smp in samples]) for i in range(L)]
    return (np.stack([smp['gt_bbox'] for smp in samples]), np.stack([smp['gt_class'] for smp in samples]),
            np.stack([smp['gt_score'] for smp in samples]), targets)


def grad_digest(g):
    """[sum, sum|.|, l2] in float64 + 64 strided samples: pins a gradient tensor without storing all of it."""
    d = g.detach().double().reshape(-1)
    step = max(1, d.numel() // 64)
    return np.array([d.sum().item(), d.abs().sum().item(), d.pow(2).sum().sqrt().item()]), d[::step][:64].float().numpy()


def g12_train_step():
    """Config (5), SURVEY 8f rank 2: ONE training forward + backward of the reference (train.py:416-443 up to
    all_loss.backward()): the loss terms, the raw head outputs of the training-mode forward (BatchNorm on batch
    statistics everywhere -- backbone.freeze() only stops gradients --, DropBlock drawing from torch's global RNG),
    d loss / d head outputs, digests of every parameter gradient, and the BatchNorm running statistics after the step."""
    for tag, C, S, N, seed in (('r18vd_96', PPYOLO_r18vd_Config, 96, 2, 0), ('r50vd_96', PPYOLO_2x_Config, 96, 2, 0)):
        cfg = C()
        m = build_ref_train(cfg, seed)
        assert m.training
        x = synth.synth_images(N, S, seed=1234)
        gt_bbox, gt_class, gt_score, targets = synth_gt(N, S, cfg, seed=77)
        assert all(t[:, :, 5].sum() > 0 for t in targets[-2:])
        torch.manual_seed(4321)                                  # DropBlock's torch.rand
        feats = m.backbone(x)
        outs = m.head._get_outputs(feats)
        for o in outs:
            o.retain_grad()
        hd = m.head
        losses = hd.yolo_loss(outs, torch.from_numpy(gt_bbox), torch.from_numpy(gt_class), torch.from_numpy(gt_score),
                              [torch.from_numpy(t) for t in targets], hd.anchors, hd.anchor_masks, hd.mask_anchors, hd.num_classes)
        all_loss = 0.0
        for k in losses:
            all_loss = all_loss + losses[k]
        all_loss.backward()
        arrs = dict(meta=np.array([S, N, seed, 1234, 4321]), gt_bbox=gt_bbox, gt_class=gt_class, gt_score=gt_score,
                    loss_names=np.array(list(losses.keys())), loss_values=np.array([float(losses[k]) for k in losses], np.float32),
                    all_loss=np.array(float(all_loss), np.float32))
        for i, t in enumerate(targets):
            arrs['target%d' % i] = t
        for i, o in enumerate(outs):
            arrs['out%d' % i] = o.detach()
            arrs['dout%d' % i] = o.grad
        names, digs, samples = [], [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                assert not p.requires_grad and k.startswith('backbone.'), k
                continue
            a, b = grad_digest(p.grad)
            names.append(k); digs.append(a); samples.append(np.pad(b, (0, 64 - len(b))))
        assert all(n.startswith('head.') for n in names)
        arrs.update(grad_names=np.array(names), grad_digest=np.stack(digs), grad_samples=np.stack(samples))
        sd = m.state_dict()
        for k in ('backbone.stage1_conv1_1.bn.running_mean', 'backbone.stage1_conv1_1.bn.running_var',
                  'head.yolo_output_convs.0.conv.bias'):
            arrs['after.' + k] = sd[k]
        bnk = [k for k in sd if k.startswith('head.') and k.endswith('running_var')][-1]
        arrs['after_name'] = np.array(bnk)
        arrs['after_value'] = sd[bnk]
        # full gradients of the small tensors: the three output convolutions
        for i in range(len(outs)):
            arrs['gw_out%d' % i] = getattr(m.head.yolo_output_convs[i].conv.weight, 'grad')
            arrs['gb_out%d' % i] = getattr(m.head.yolo_output_convs[i].conv.bias, 'grad')
        print(tag, {k: float(losses[k]) for k in losses}, 'trainable tensors', len(names))
        save('g12_train_' + tag, **arrs)


def g13_ema():
    """ExponentialMovingAverage of the trainable parameters (reference model/EMA.py:16-44; train.py:283-286, :443-444): the
    reference's own class on a small module through four updates -- the warm-up of the decay, min(0.9998, (1+t)/(10+t)), and
    numpy's float32 arithmetic of `decay * old + (1 - decay) * new`."""
    from model.EMA import ExponentialMovingAverage
    g = gen(5)
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.randn(7, 5, generator=g))
    m.b = torch.nn.Parameter(torch.randn(11, generator=g) * 1e-3)
    m.frozen = torch.nn.Parameter(torch.randn(3, generator=g), requires_grad=False)
    ema = ExponentialMovingAverage(m, 0.9998)
    ema.register()
    arrs = dict(a0=m.a.detach().clone(), b0=m.b.detach().clone())
    decays = []
    for t in range(4):
        with torch.no_grad():
            m.a.add_(torch.randn(7, 5, generator=g) * 0.1)
            m.b.mul_(1.5)
            m.frozen.add_(1.0)
        decays.append(ema.update())
        arrs['a_param%d' % t], arrs['b_param%d' % t] = m.a.detach().clone(), m.b.detach().clone()
        arrs['a_shadow%d' % t], arrs['b_shadow%d' % t] = ema._shadow['a'].copy(), ema._shadow['b'].copy()
        assert ema._shadow['a'].dtype == np.float32 and 'frozen' not in ema._shadow
    arrs['decays'] = np.array(decays, dtype=np.float64)
    save('g13_ema', **arrs)


def g14_train_loop():
    """The reference's training LOOP through its public surface (train.py:264-286, :416-444): backbone.freeze(),
    model.add_param_group -> torch.optim.SGD(momentum, per-group weight decay), ExponentialMovingAverage, then three times
    `losses = model(images, None, False, gt_bbox, gt_class, gt_score, targets); sum(losses).backward(); optimizer.step();
    ema.update()` -- the loss terms of every iteration, every head parameter after the third, the EMA shadows, and the
    parameter groups of both configurations.  DropBlock is switched to its test mode (its masks come from torch's global
    RNG; the masks themselves are pinned by g12), everything else is the reference's training mode."""
    from model.EMA import ExponentialMovingAverage
    arrs = {}
    for tag, C in (('r18vd', PPYOLO_r18vd_Config), ('r50vd', PPYOLO_2x_Config)):
        cfg = C()
        m = build_ref_train(cfg, 0)
        groups = []
        m.add_param_group(groups, cfg.learningRate['base_lr'], cfg.optimizerBuilder['regularizer']['factor'])
        names = {id(q): k for k, q in m.named_parameters()}
        arrs[tag + '.group_names'] = np.array([names[id(g['params'][0])] for g in groups])
        arrs[tag + '.group_lr_wd'] = np.array([[g['lr'], g['base_lr'], g['weight_decay']] for g in groups], np.float64)
        assert all(len(g['params']) == 1 for g in groups)
    cfg = PPYOLO_r18vd_Config()
    S, N, base_lr, wd, mom = 96, 2, 0.01, 0.0005, 0.9
    m = build_ref_train(cfg, 0)
    m.head.set_dropblock(is_test=True)
    groups = []
    m.add_param_group(groups, base_lr, wd)
    opt = torch.optim.SGD(groups, lr=base_lr, momentum=mom, weight_decay=wd)
    ema = ExponentialMovingAverage(m, 0.9998)
    ema.register()
    gt_bbox, gt_class, gt_score, targets = synth_gt(N, S, cfg, seed=77)
    arrs.update(meta=np.array([S, N, 0, 77]), hyper=np.array([base_lr, wd, mom, 0.9998]), gt_bbox=gt_bbox, gt_class=gt_class,
                gt_score=gt_score)
    for i, t in enumerate(targets):
        arrs['target%d' % i] = t
    T = torch.from_numpy
    for it in range(3):
        x = synth.synth_images(N, S, seed=1234 + it)
        losses = m(x, None, False, T(gt_bbox), T(gt_class), T(gt_score), [T(t) for t in targets])
        all_loss = 0.0
        for k in losses:
            all_loss = all_loss + losses[k]
        lr = base_lr * (1.0 - 0.25 * it)                            # the loop rescales every group's lr each iteration (:437-439)
        for g in opt.param_groups:
            g['lr'] = lr * g['base_lr'] / base_lr
        opt.zero_grad()
        all_loss.backward()
        opt.step()
        ema.update()
        arrs['loss_names'] = np.array(list(losses.keys()))
        arrs['losses%d' % it] = np.array([float(losses[k]) for k in losses], np.float32)
    sd = m.state_dict()
    heads = [k for k, q in m.named_parameters() if q.requires_grad]
    arrs['param_names'] = np.array(heads)
    digs, smp, sdig, ssmp = [], [], [], []
    for k in heads:
        a, b = grad_digest(sd[k])
        digs.append(a); smp.append(np.pad(b, (0, 64 - len(b))))
        a, b = grad_digest(torch.from_numpy(ema._shadow[k]))
        sdig.append(a); ssmp.append(np.pad(b, (0, 64 - len(b))))
    arrs.update(param_digest=np.stack(digs), param_samples=np.stack(smp), shadow_digest=np.stack(sdig), shadow_samples=np.stack(ssmp))
    m0 = build_ref_train(cfg, 0).state_dict()
    arrs['update_l2'] = np.array([float((sd[k].double() - m0[k].double()).norm()) for k in heads])
    for k in ('head.yolo_output_convs.1.conv.bias', 'head.detection_blocks.0.layers.2.bn.weight'):
        arrs['after.' + k] = sd[k]
    save('g14_train_loop', **arrs)


def g15_dcn_backward():
    """Backward of the reference's DCNv2 module (model/custom_layers.py:486-677) by torch autograd, the semantics the
    reference trains with when freeze_at < 5: for a random upstream gradient dy, the gradients of the input (whole: sampling
    path + conv_offset path), of the raw conv_offset output (18 offsets + 9 mask logits), and of the three parameter tensors.
    Offsets of ~2 px with a few taps pushed far outside (clamped: zero offset gradient there), mask logits on both tails."""
    g = gen(1515)
    out = {}
    cases = [(32, 64, 1, 2, 9, 9), (64, 32, 2, 3, 10, 12), (32, 32, 1, 1, 6, 5)]
    for i, (ci, co, s, N, H, W) in enumerate(cases):
        m = DCNv2(ci, co, filter_size=3, stride=s, padding=1, bias_attr=False)
        with torch.no_grad():
            m.conv_offset.weight.copy_(torch.randn(m.conv_offset.weight.shape, generator=g) * (2.0 / (ci * 9) ** 0.5))
            b = torch.randn(27, generator=g) * 1.0
            b[3] = 12.3
            b[10] = -11.7
            b[20] = 4.0
            b[25] = -5.0
            m.conv_offset.bias.copy_(b)
            m.dcn_weight.copy_(torch.randn(m.dcn_weight.shape, generator=g) * (1.0 / (ci * 9)) ** 0.5)
        x = torch.randn(N, ci, H, W, generator=g, requires_grad=True)
        kept = {}
        def keep(mod, inp, o):
            o.retain_grad()
            kept['om'] = o
        h = m.conv_offset.register_forward_hook(keep)
        y = m(x)
        h.remove()
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        p = 'b%d_' % i
        out[p + 'meta'] = np.array([ci, co, s])
        out[p + 'x'], out[p + 'dy'], out[p + 'y'] = x.detach(), dy, y.detach()
        out[p + 'w_off'], out[p + 'b_off'], out[p + 'w_dcn'] = m.conv_offset.weight.detach(), m.conv_offset.bias.detach(), m.dcn_weight.detach()
        out[p + 'offset_mask'] = kept['om'].detach()
        out[p + 'dx'] = x.grad
        out[p + 'd_offset_mask'] = kept['om'].grad
        out[p + 'dw_off'], out[p + 'db_off'], out[p + 'dw_dcn'] = m.conv_offset.weight.grad, m.conv_offset.bias.grad, m.dcn_weight.grad
        assert float(kept['om'].grad[:, :18].abs().max()) > 0 and float((kept['om'].grad[:, 3] == 0).float().mean()) > 0.5      # pushed out: clamped
    out['ncases'] = np.array(len(cases))
    save('g15_dcn_backward', **out)


def g16_train_step_backbone():
    """One training forward + backward of the reference with backbone stages training too (freeze_at = 3: stages 4 and 5,
    i.e. the three DCNv2 bottlenecks, the strided 3x3 of stage4_0, both avg-pool shortcuts; r18vd with freeze_at = 2, and 0 = the
    whole network incl. the stem and its max-pool): as g12
    -- loss terms, digests of every parameter gradient -- with DropBlock in test mode (masks: g12)."""
    for tag, C, S, N, fa in (('r50vd_128_fa3', PPYOLO_2x_Config, 128, 2, 3), ('r50vd_320_fa3', PPYOLO_2x_Config, 320, 2, 3), ('r18vd_96_fa2', PPYOLO_r18vd_Config, 96, 2, 2),
                             ('r18vd_96_fa0', PPYOLO_r18vd_Config, 96, 2, 0)):
        cfg = C()
        cfg.backbone['freeze_at'] = fa
        m = build_ref_train(cfg, 0)
        m.head.set_dropblock(is_test=True)
        x = synth.synth_images(N, S, seed=1234)
        gt_bbox, gt_class, gt_score, targets = synth_gt(N, S, cfg, seed=77)
        T = torch.from_numpy
        losses = m(x, None, False, T(gt_bbox), T(gt_class), T(gt_score), [T(t) for t in targets])
        all_loss = 0.0
        for k in losses:
            all_loss = all_loss + losses[k]
        all_loss.backward()
        arrs = dict(meta=np.array([S, N, 0, 1234, fa]), gt_bbox=gt_bbox, gt_class=gt_class, gt_score=gt_score,
                    loss_names=np.array(list(losses.keys())), loss_values=np.array([float(losses[k].detach()) for k in losses], np.float32))
        for i, t in enumerate(targets):
            arrs['target%d' % i] = t
        names, digs, samples = [], [], []
        for k, q in m.named_parameters():
            stage = int(k[len('backbone.stage')]) if k.startswith('backbone.stage') else 6
            assert (q.grad is not None) == (stage > fa), k
            if q.grad is None:
                continue
            a, b = grad_digest(q.grad)
            names.append(k); digs.append(a); samples.append(np.pad(b, (0, 64 - len(b))))
        arrs.update(grad_names=np.array(names), grad_digest=np.stack(digs), grad_samples=np.stack(samples))
        save('g16_train_' + tag, **arrs)


def g17_paddle_names():
    """SURVEY 8f rank 3, the optional Paddle checkpoint mapping: which PaddleDetection variable feeds which state_dict entry, taken
    from the reference's converter scripts (1_ppyolo_2x_2pytorch.py:63-317, 1_ppyolo_r18vd_2pytorch.py) by RUNNING them on a
    recording dict: `fluid.io.load_program_state` (paddle is not in this image; the scripts use it for exactly this one call,
    which returns {variable name: ndarray}) is replaced by a dict that hands out a one-element array carrying the index of the
    requested name, so that after the script every parameter / buffer it filled holds the index of its source.  Only the names
    are pinned here -- no Paddle arithmetic is involved in the conversion."""
    import runpy
    import types

    class Recorder(dict):
        def __init__(self):
            self.names = []

        def __getitem__(self, k):
            self.names.append(k)
            return np.full((1,), len(self.names) - 1, np.float32)
    out = {}
    for tag, script, C in (('r50vd', '1_ppyolo_2x_2pytorch.py', PPYOLO_2x_Config), ('r18vd', '1_ppyolo_r18vd_2pytorch.py', PPYOLO_r18vd_Config)):
        rec = Recorder()
        paddle, fluid = types.ModuleType('paddle'), types.ModuleType('paddle.fluid')
        fluid.io = types.SimpleNamespace(load_program_state=lambda path: rec)
        paddle.fluid = fluid
        sys.modules['paddle'], sys.modules['paddle.fluid'] = paddle, fluid
        real_save, cwd = torch.save, os.getcwd()
        kept = {}
        torch.save = lambda obj, path: kept.__setitem__('sd', obj)
        os.chdir('/tmp')
        try:
            runpy.run_path(os.path.join(REF, script), run_name='__main__')
        finally:
            torch.save = real_save
            os.chdir(cwd)
            del sys.modules['paddle'], sys.modules['paddle.fluid']
        sd = kept['sd']
        keys, names = [], []
        for k, v in sd.items():
            if k.endswith('num_batches_tracked'):
                continue
            assert v.numel() == 1, (k, tuple(v.shape))          # every entry was filled by the script
            keys.append(k)
            names.append(rec.names[int(v.reshape(-1)[0])])
        assert len(set(names)) == len(names) == len(rec.names)
        out[tag + '.keys'], out[tag + '.paddle'] = np.array(keys), np.array(names)
    save('g17_paddle_names', **out)


_G18_IMS = {'a': [[480., 640.]] * 8,                                                     # BASELINE.md's im_size
            'b': [[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]] * 2}            # tests/test_gpu_model.py::_FULL_IMS
# Evaluations of the SAME reference network that are all "the reference": the first is the primary fixture, the others measure
# the reference against ITSELF.  On the build box MKLDNN picks the same blocking for 2..16 threads (bit-identical results, measured),
# so the ensemble is: 8 threads; 1 thread; ATen's native convolution (torch.backends.mkldnn off); the images one at a time;
# and the reference's own modules in float64 (= the exact answer).
_G18_RUNS = ('t8', 't1', 'native', 'bs1', 'f64')


class _RecordNms:
    """The reference's matrix_nms (model/matrix_nms.py:102-151) hands back rows, not indices, and full-size batches contain
    bit-identical boxes (dozens of boxes clipped to the whole image), so rows cannot be matched back by value.  This records
    what the reference itself computes on the way -- the two torch.argsort results and the decayed scores _matrix_nms returns
    -- from which the flat index (box * C + class) of every returned row follows by the reference's own indexing."""

    def __enter__(self):
        import model.matrix_nms as mm
        self.mm, self.real_sort, self.real_inner = mm, torch.argsort, mm._matrix_nms
        self.sorts, self.decayed = [], []

        def argsort(*a, **k):
            r = self.real_sort(*a, **k)
            self.sorts.append(r)
            return r

        def inner(*a, **k):
            r = self.real_inner(*a, **k)
            self.decayed.append(r)
            return r
        torch.argsort, mm._matrix_nms = argsort, inner
        return self

    def __exit__(self, *exc):
        torch.argsort, self.mm._matrix_nms = self.real_sort, self.real_inner

    def keep_index(self, pred, boxes, scores, nms_cfg, C=80):
        if pred[0, 0] < 0:
            return np.zeros((0,), np.int64)
        assert len(self.sorts) == 2 and len(self.decayed) == 1
        cand = (scores > nms_cfg['score_threshold']).nonzero()                    # matrix_nms.py:110, :115
        first = self.sorts[0][:nms_cfg['nms_top_k']] if nms_cfg['nms_top_k'] > 0 else self.sorts[0]
        kept = cand[first][self.decayed[0] >= nms_cfg['post_threshold']]          # :131
        rows = kept[self.sorts[1][:nms_cfg['keep_top_k']]]                        # :139-141
        assert torch.equal(rows[:, 1].float(), pred[:, 0]) and torch.equal(boxes[rows[:, 0]], pred[:, 2:6])
        return (rows[:, 0] * C + rows[:, 1]).numpy().astype(np.int64)


def g18_headline_sizes():
    """BASELINE.json configs[1] / configs[2] at their OWN sizes, by the reference itself (model/ppyolo.py:19-22 ->
    model/head.py:424-469): r18vd-416 and R50vd-608, 8 images, with (a) BASELINE's im_size = (480, 640) and (b) the mixed
    original sizes of the GPU tests.  Evaluated five ways (_G18_RUNS): the summation order of the reference's own CPU
    convolutions changes with the thread count / backend / batch, which gives the distance of the reference from ITSELF
    (two correct fp32 evaluations of one network), and its float64 run gives its distance from the exact answer.  Stored:
    detections + keep indices of every run, the spread, and a strided sample of the raw head outputs (the full tensors are
    62 MB) of the primary and the float64 run."""
    import copy
    from model.head import get_iou_aware_score as gi, yolo_box as yb
    for tag, C, S in (('r18vd_416', PPYOLO_r18vd_Config, 416), ('r50vd_608', PPYOLO_2x_Config, 608)):
        cfg = C()
        m, _ = build_ref(cfg, 0)
        m64 = copy.deepcopy(m).double()
        x = synth.synth_images(8, S)
        arrs = {'meta': np.array([S, 8, 0, 0]), 'runs': np.array(_G18_RUNS)}
        for k, v in _G18_IMS.items():
            arrs['im_size_' + k] = np.array(v, np.float32)
        nms = {k_: v_ for k_, v_ in cfg.nms_cfg.items() if k_ != 'nms_type'}

        def network(run):
            mm, xx = (m64, x.double()) if run == 'f64' else (m, x)
            with torch.no_grad():
                if run == 'bs1':
                    per = [mm.backbone(xx[i:i + 1]) for i in range(8)]
                    feats = [torch.cat([p[l] for p in per]) for l in range(len(per[0]))]
                    outs = [torch.cat(o) for o in zip(*[mm.head._get_outputs(p) for p in per])]
                else:
                    feats = mm.backbone(xx)
                    outs = mm.head._get_outputs(feats)
            return mm, feats, outs

        heads = {}
        for run in _G18_RUNS:
            torch.set_num_threads(1 if run == 't1' else 8)          # in force for everything this run computes
            torch.backends.mkldnn.enabled = run != 'native'
            mm, feats, outs = network(run)
            heads[run] = outs
            with torch.no_grad():
                for k, v in _G18_IMS.items():
                    ims = torch.tensor(v, dtype=outs[0].dtype)
                    if run == _G18_RUNS[0]:
                        arrs['ncand_' + k] = np.array(_assert_tie_free(m, x, ims, cfg))
                    bs, ss = [], []
                    for i, o in enumerate(outs):
                        if mm.head.iou_aware:
                            o = gi(o, 3, 80, mm.head.iou_aware_factor)
                        b_, s_ = yb(o, mm.head._anchors[mm.head.anchor_masks[i]], mm.head.downsample[i], 80, mm.head.scale_x_y,
                                    ims, True, cfg.nms_cfg['score_threshold'])
                        bs.append(b_)
                        ss.append(s_)
                    boxes, scores = torch.cat(bs, 1), torch.cat(ss, 1)
                    whole = mm.head.get_prediction(feats, ims)                  # the reference's own path, untouched
                    for i in range(8):
                        with _RecordNms() as rec:
                            p = matrix_nms(boxes[i], scores[i], **nms)
                        assert torch.equal(p, whole[i]), (run, k, i, p.shape, whole[i].shape, (p - whole[i]).abs().max() if p.shape == whole[i].shape else None)
                        assert len(torch.unique(p[:, 1])) == p.shape[0]
                        arrs['%s_%s_pred%d' % (run, k, i)] = p
                        arrs['%s_%s_keep%d' % (run, k, i)] = rec.keep_index(p, boxes[i], scores[i], nms)
        torch.set_num_threads(8)
        torch.backends.mkldnn.enabled = True
        # strided samples of the raw head outputs (primary and float64) + the reference-vs-reference distance per level
        r0 = _G18_RUNS[0]
        for lv, o in enumerate(heads[r0]):
            flat = o.reshape(-1)
            idx = torch.arange(0, flat.numel(), max(1, flat.numel() // 8192))[:8192]
            arrs['out%d_idx' % lv], arrs['out%d_val' % lv] = idx.numpy(), flat[idx]
            arrs['out%d_val64' % lv] = heads['f64'][lv].reshape(-1)[idx]
            arrs['out%d_shape' % lv] = np.array(o.shape)
            arrs['out%d_rms' % lv] = np.array(float(o.double().pow(2).mean().sqrt()))
            arrs['out%d_spread_max' % lv] = np.array([float((heads[r][lv].double() - o.double()).abs().max()) for r in _G18_RUNS])
            arrs['out%d_spread_rms' % lv] = np.array([float((heads[r][lv].double() - o.double()).pow(2).mean().sqrt()) for r in _G18_RUNS])
            print('   %s head level %d: reference runs %s vs %s: max %s rms %s' % (
                tag, lv, _G18_RUNS, r0, ['%.2e' % v for v in arrs['out%d_spread_max' % lv]], ['%.2e' % v for v in arrs['out%d_spread_rms' % lv]]))
        # the reference against itself on the detections: rows matched by keep index
        for k in _G18_IMS:
            sp = np.zeros((len(_G18_RUNS), 5))       # unmatched rows, rows out of place, max |score|, max |box| px, max |box| / side
            for j, run in enumerate(_G18_RUNS):
                for i in range(8):
                    a, ka = arrs['%s_%s_pred%d' % (r0, k, i)], arrs['%s_%s_keep%d' % (r0, k, i)]
                    b, kb = arrs['%s_%s_pred%d' % (run, k, i)], arrs['%s_%s_keep%d' % (run, k, i)]
                    pos = {int(q): r for r, q in enumerate(kb)}
                    ia = [r for r, q in enumerate(ka) if int(q) in pos]
                    ib = [pos[int(ka[r])] for r in ia]
                    A, B = a[ia].double(), b[ib].double()
                    side = torch.maximum(A[:, 4] - A[:, 2], A[:, 5] - A[:, 3]).clamp_min(1.0)
                    d = (A[:, 2:] - B[:, 2:]).abs().max(dim=1).values
                    sp[j] = np.maximum(sp[j], [max(len(ka), len(kb)) - len(ia), sum(1 for p_, q_ in zip(ia, ib) if p_ != q_),
                                               float((A[:, 1] - B[:, 1]).abs().max()), float(d.max()), float((d / side).max())])
            arrs['spread_' + k] = sp
            print('   %s im_size %s: reference runs %s vs %s: rows unmatched / out of place / score / box px / box of side\n%s'
                  % (tag, k, _G18_RUNS, r0, sp))
        save('g18_' + tag, **arrs)


ALL = dict(g1=g1_conv_units, g2=g2_dcn, g3=g3_coord_spp, g4=g4_decode, g5=g5_matrix_nms, g67=g6_g7_models,
           g8=g8_preprocess, g9=g9_decode_harness, g10=g10_coco_records, g11=g11_state_dict_layout, g12=g12_train_step, g13=g13_ema, g14=g14_train_loop, g15=g15_dcn_backward, g16=g16_train_step_backbone, g17=g17_paddle_names, g18=g18_headline_sizes)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    for k, fn in ALL.items():
        if a.only is None or a.only == k:
            fn()
