"""Pin the CPU oracle (oracle/ppyolo_oracle.py) against fixtures produced by the
reference itself (tools/make_goldens.py).  Same ATen ops in the same order -> the
oracle is expected to be BIT-EXACT on this torch build; a tiny tolerance is allowed
only where the oracle deliberately restructures the arithmetic (none today)."""
import numpy as np
import pytest
import torch

from oracle import ppyolo_oracle as orc
from ppyolo_hip import synth
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config

ACTS = {0: None, 1: 'relu', 2: 'leaky'}


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_conv_units(golden):
    g = golden('g1_conv_units')
    for i in range(int(g['ncases'])):
        p = 'c%d_' % i
        ci, co, k, s, act, bn, bias = [int(v) for v in g[p + 'meta']]
        sd = {'u.conv.weight': T(g[p + 'w'])}
        if bias:
            sd['u.conv.bias'] = T(g[p + 'b'])
        if bn:
            b = T(g[p + 'bn'])
            sd.update({'u.bn.weight': b[0], 'u.bn.bias': b[1], 'u.bn.running_mean': b[2],
                       'u.bn.running_var': b[3]})
        y = orc.conv_unit(sd, 'u', T(g[p + 'x']), s, ACTS[act])
        assert torch.equal(y, T(g[p + 'y'])), i


def test_g2_dcnv2(golden):
    g = golden('g2_dcnv2')
    for i in range(int(g['ncases'])):
        p = 'd%d_' % i
        ci, co, s = [int(v) for v in g[p + 'meta']]
        sd = {'d.dcn_weight': T(g[p + 'w_dcn']), 'd.conv_offset.weight': T(g[p + 'w_off']),
              'd.conv_offset.bias': T(g[p + 'b_off'])}
        y = orc.dcnv2(sd, 'd', T(g[p + 'x']), stride=s)
        assert torch.equal(y, T(g[p + 'y'])), i
        # second oracle: explicit per-corner bounds checks (reference DCNv2_Slow)
        assert (y - T(g[p + 'y_slow'])).abs().max() < 5e-5


def test_g15_dcnv2_backward(golden):
    """torch autograd through the oracle's DCNv2 = the reference's own gradients (golden g15: autograd through the
    reference's module): the oracle is the checker of ppy_dcnv2_backward_f32 on the GPU."""
    g = golden('g15_dcn_backward')
    for i in range(int(g['ncases'])):
        p = 'b%d_' % i
        ci, co, s = [int(v) for v in g[p + 'meta']]
        sd = {'d.dcn_weight': T(g[p + 'w_dcn']).requires_grad_(), 'd.conv_offset.weight': T(g[p + 'w_off']).requires_grad_(),
              'd.conv_offset.bias': T(g[p + 'b_off']).requires_grad_()}
        x = T(g[p + 'x']).requires_grad_()
        y = orc.dcnv2(sd, 'd', x, stride=s)
        assert torch.equal(y.detach(), T(g[p + 'y'])), i
        y.backward(T(g[p + 'dy']))
        for got, key in ((x.grad, 'dx'), (sd['d.dcn_weight'].grad, 'dw_dcn'), (sd['d.conv_offset.weight'].grad, 'dw_off'),
                         (sd['d.conv_offset.bias'].grad, 'db_off')):
            want = T(g[p + key])
            assert (got - want).abs().max() <= 2e-6 * want.abs().max(), (i, key)


def test_g3_coord_spp(golden):
    g = golden('g3_coord_spp')
    assert torch.equal(orc.coord_concat(T(g['coord_x'])), T(g['coord_y']))
    for i in range(3):
        assert torch.equal(orc.spp(T(g['spp%d_x' % i])), T(g['spp%d_y' % i]))


def test_g4_decode(golden):
    g = golden('g4_decode')
    anchors = g['anchors']
    im_size = T(g['im_size'])
    for i in range(3):
        meta = [int(v) for v in g['l%d_meta' % i]]
        S, stride, iou_aware, mask = meta[0], meta[1], meta[2], meta[3:]
        o = T(g['l%d_out' % i])
        if iou_aware:
            o = orc.iou_aware_score(o, 3, 80, 0.4)
        b, s = orc.yolo_box(o, anchors[mask], stride, 80, 1.05, im_size, True)
        assert torch.equal(b, T(g['l%d_boxes' % i]))
        assert torch.equal(s, T(g['l%d_scores' % i]))
        # -0.0 from the `x0 * 0` clip is preserved
        assert np.array_equal(np.signbit(b.numpy()), np.signbit(g['l%d_boxes' % i]))


def test_g5_matrix_nms(golden):
    g = golden('g5_matrix_nms')
    for i in range(int(g['ncases'])):
        c = g['n%d_cfg' % i]
        pred = orc.matrix_nms(T(g['n%d_boxes' % i]), T(g['n%d_scores' % i]), float(np.float32(c[0])),
                              float(np.float32(c[1])), int(c[2]), int(c[3]), bool(c[4]), float(c[5]))
        ref = T(g['n%d_pred' % i])
        assert pred.shape == ref.shape, i
        assert torch.equal(pred, ref), i


@pytest.mark.parametrize('tag,cfgc', [('r18vd_64', PPYOLO_r18vd_Config), ('r50vd_96', PPYOLO_2x_Config)])
def test_g6_backbone_head(golden, tag, cfgc, model_shapes):
    g = golden('g6_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    sd = synth.synth_state_dict(model_shapes(cfg), seed=seed)
    feats, outs = orc.backbone_and_head(sd, cfg, synth.synth_images(N, S, seed=iseed))
    for i, f in enumerate(feats):
        assert torch.equal(f, T(g['feat%d' % i]))
    for i, o in enumerate(outs):
        assert torch.equal(o, T(g['out%d' % i]))


@pytest.mark.parametrize('tag,cfgc', [('r18vd_320', PPYOLO_r18vd_Config), ('r50vd_160', PPYOLO_2x_Config)])
def test_g7_end_to_end(golden, tag, cfgc, model_shapes):
    g = golden('g7_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    sd = synth.synth_state_dict(model_shapes(cfg), seed=seed)
    preds = orc.ppyolo_forward(sd, cfg, synth.synth_images(N, S, seed=iseed), T(g['im_size']))
    for i, p in enumerate(preds):
        ref = T(g['pred%d' % i])
        assert p.shape == ref.shape
        assert torch.equal(p, ref)


def test_g9_decode_harness(golden, model_shapes):
    """The reference's Decode.detect_image / detect_batch (numpy in, (boxes, scores, classes) out) = the oracle's
    forward split the same way (model/decode_np.py:41-96)."""
    g = golden('g9_decode_harness')
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = PPYOLO_r18vd_Config()
    sd = synth.synth_state_dict(model_shapes(cfg), seed=seed)
    x = synth.synth_images(N, S, seed=iseed)
    ims = T(g['im_size']).float()
    one = orc.ppyolo_forward(sd, cfg, x[:1], ims[:1])[0].numpy()
    assert np.array_equal(one[:, 2:], g['boxes']) and np.array_equal(one[:, 1], g['scores'])
    assert np.array_equal(one[:, 0].astype(np.int32), g['classes'])
    both = orc.ppyolo_forward(sd, cfg, x, ims)
    for i, p in enumerate(both):
        p = p.numpy()
        assert np.array_equal(p[:, 2:], g['b_boxes%d' % i]) and np.array_equal(p[:, 1], g['b_scores%d' % i])
        assert np.array_equal(p[:, 0].astype(np.int32), g['b_classes%d' % i])


@pytest.mark.parametrize('tag,cfgc', [('r18vd_416', PPYOLO_r18vd_Config), ('r50vd_608', PPYOLO_2x_Config)])
def test_g18_headline_sizes(golden, tag, cfgc, model_shapes):
    """BASELINE.json configs[1] / configs[2] at their own sizes (batch 8): the oracle -- fp32 and float64 -- against rows AND
    keep indices the reference itself produced there (tools/make_goldens.py g18: the reference's matrix_nms instrumented for
    its indices).  Bit for bit at the fixture's thread count; the fixture also carries the reference's distance from ITSELF
    under other summation orders (1 thread, ATen's native convolution, one image at a time), checked to be self-consistent."""
    g = golden('g18_' + tag)
    S, N = int(g['meta'][0]), int(g['meta'][1])
    cfg = cfgc()
    sd = synth.synth_state_dict(model_shapes(cfg), seed=0)
    x = synth.synth_images(N, S)
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        feats, outs = orc.backbone_and_head(sd, cfg, x)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        _, outs64 = orc.backbone_and_head(sd64, cfg, x.double())
    finally:
        torch.set_num_threads(threads)
    for lv, o in enumerate(outs):
        idx = T(g['out%d_idx' % lv])
        assert tuple(o.shape) == tuple(g['out%d_shape' % lv])
        assert torch.equal(o.reshape(-1)[idx], T(g['out%d_val' % lv])), 'head level %d' % lv
        # (the reference's DCNv2 keeps fp32 pieces -- the sampling grid's arange -- inside its float64 run, the oracle's float64
        # run is float64 throughout: 5e-8 apart at R50vd, bit-equal at r18vd; the fp32 runs are 3e-6 rms from either)
        assert (outs64[lv].reshape(-1)[idx] - T(g['out%d_val64' % lv])).abs().max() <= 2e-7, 'head level %d (float64)' % lv
    nms = dict(cfg.nms_cfg)
    nms.pop('nms_type')
    for k in ('a', 'b'):
        for run, oo in (('t8', outs), ('f64', outs64)):
            with torch.no_grad():
                boxes, scores = orc.decode_all(oo, cfg.head, T(g['im_size_' + k]).to(oo[0].dtype))
            for i in range(N):
                rows, keep = orc.matrix_nms(boxes[i], scores[i], return_index=True, **nms)
                ref = T(g['%s_%s_pred%d' % (run, k, i)])
                assert torch.equal(rows, ref) if run == 't8' else (rows.shape == ref.shape and (rows - ref).abs().max() <= 1e-4), (k, run, i)
                assert np.array_equal(np.asarray(keep), g['%s_%s_keep%d' % (run, k, i)]), (k, run, i)
        sp = g['spread_' + k]                       # rows: runs; columns: unmatched, out of place, score, box px, box / side
        assert sp.shape == (len(g['runs']), 5) and not sp[0].any() and sp[:, 0].max() == 0
        assert sp[:, 2].max() <= 1e-6 and sp[:, 4].max() <= 2e-5      # the reference agrees with itself to ~1e-5 of a box side
