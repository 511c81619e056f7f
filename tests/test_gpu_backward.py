"""Backward kernels of the training step: ppy_conv2d_dgrad_f32 / ppy_conv2d_wgrad_f32 through
the C ABI against torch autograd of F.conv2d on the CPU (the reference's backward IS torch autograd: train.py:441), on the
head's layer shapes, on ragged shapes, and on the gradients the REFERENCE produced in its own training step (golden g12:
the three output convolutions' weight gradients and d loss / d head outputs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _ref(x, w, dy, stride, pad):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    y.backward(dy)
    return x.grad, w.grad


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


CASES = [  # N, H, W, C, K, R, stride: head layer shapes at small maps + ragged ones
    (2, 19, 19, 512, 1024, 3, 1), (2, 19, 19, 1024, 512, 1, 1), (2, 19, 19, 1024, 258, 1, 1), (1, 38, 38, 256, 512, 3, 1),
    (2, 12, 12, 128, 258, 1, 1), (3, 7, 5, 64, 96, 3, 1), (1, 9, 9, 32, 40, 3, 1), (2, 6, 6, 96, 255, 1, 1), (1, 1, 1, 64, 64, 1, 1),
    (2, 11, 3, 160, 33, 3, 1), (5, 2, 2, 32, 130, 3, 1),
]


@pytest.mark.parametrize('case', CASES)
def test_dgrad_and_wgrad_vs_autograd(case):
    from ppyolo_hip import ops
    N, H, W, C, K, R, stride = case
    pad = (R - 1) // 2
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(N, K, Ho, Wo, generator=g) * torch.exp(torch.randn(N, K, Ho, Wo, generator=g))
    dx_ref, dw_ref = _ref(x, w, dy, stride, pad)
    xd, dyd = nhwc(x).cuda(), nhwc(dy).cuda()
    wk = w.permute(0, 2, 3, 1).contiguous().cuda()
    dx = torch.full((N, H, W, C), 7.0).cuda()
    dw = torch.full((K, R, R, C), 7.0).cuda()
    ops.conv2d_dgrad(ops.View(dyd), wk, ops.View(dx), stride, pad)
    ops.conv2d_wgrad(ops.View(xd), ops.View(dyd), dw, stride, pad)
    torch.cuda.synchronize()
    e_dx, e_dw = _rel(dx.cpu().permute(0, 3, 1, 2), dx_ref), _rel(dw.cpu().permute(0, 3, 1, 2), dw_ref)
    assert e_dx <= 2e-5 and e_dw <= 2e-5, (case, e_dx, e_dw)
    # run-to-run identical (fixed summation order of the pixel slices)
    dw2 = torch.zeros_like(dw)
    ops.conv2d_wgrad(ops.View(xd), ops.View(dyd), dw2, stride, pad)
    assert torch.equal(dw, dw2)


def test_wgrad_strided_and_channel_slices():
    """wgrad with stride 2 and operands that are channel slices of wider buffers (ld > C), as concat buffers are."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(7)
    N, H, W, C, K, R, stride, pad = 2, 13, 10, 64, 48, 3, 2, 1
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) * 0.05
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(N, K, Ho, Wo, generator=g)
    _, dw_ref = _ref(x, w, dy, stride, pad)
    xwide = torch.randn(N, H, W, C + 32, generator=g).cuda()
    xwide[..., 16:16 + C] = nhwc(x).cuda()
    dywide = torch.randn(N, Ho, Wo, K + 16, generator=g).cuda()
    dywide[..., 8:8 + K] = nhwc(dy).cuda()
    dw = torch.zeros(K, R, R, C).cuda()
    ops.conv2d_wgrad(ops.View(xwide, 16, C), ops.View(dywide, 8, K), dw, stride, pad)
    torch.cuda.synchronize()
    assert _rel(dw.cpu().permute(0, 3, 1, 2), dw_ref) <= 2e-5


def test_wgrad_f16x2_with_tracked_maxima():
    """The weight gradient on the f16x2 kernel (two fp16 terms after ONE power-of-two scale per operand from the tracked
    maxima, 3 MFMA products) against float64: not worse than the bf16x3 kernel, on operands with the dynamic range of real
    gradients (a few large values, most tiny) and with images of very different magnitude."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(17)
    for N, H, W, C, K, R, stride in ((4, 19, 19, 256, 512, 3, 1), (2, 38, 38, 288, 128, 1, 1), (3, 13, 10, 64, 48, 3, 2)):
        pad = (R - 1) // 2
        x = torch.randn(N, H, W, C, generator=g) * torch.tensor([1.0, 30.0, 0.02, 5.0][:N]).view(N, 1, 1, 1)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        dy = torch.randn(N, Ho, Wo, K, generator=g) * torch.exp(torch.randn(N, Ho, Wo, K, generator=g) * 3.0) * 1e-4
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (K, C, R, R), dy.permute(0, 3, 1, 2).double(), stride=stride,
                                          padding=pad).permute(0, 2, 3, 1)
        xd, dyd = x.cuda(), dy.cuda()
        err = {}
        for name, kw in (('bf16x3', {}), ('f16x2', dict(amax_x=ops.amax_slots(xd), amax_dy=ops.amax_slots(dyd)))):
            dw = torch.full((K, R, R, C), float('nan'), device='cuda')
            ops.conv2d_wgrad(ops.View(xd), ops.View(dyd), dw, stride, pad, **kw)
            err[name] = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
        print('wgrad N%d %dx%d C%d K%d R%d s%d: max error / max|dw| vs float64: %s' % (N, H, W, C, K, R, stride, err))
        assert err['f16x2'] <= max(2.0 * err['bf16x3'], 2e-6) and err['f16x2'] <= 1e-5


def test_wgrad_nine_tap_kernel_edge_shapes():
    """Round 3: 3x3 / stride 1 weight gradients run with all nine taps in one workgroup (conv_wgrad9_kernel: rows padded to a
    multiple of 4 as virtual pixels, column-shifted quads built in registers).  Against float64 and against the one-tap kernel
    (PPY_WGRAD9=0) on shapes with W % 4 != 0, one-row / one-column maps, a K tail, several 32-channel tiles, channel slices of
    wider buffers, and more pixels than one slice."""
    import os
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(29)
    for (N, H, W, C, K) in ((2, 7, 10, 32, 72), (1, 1, 5, 64, 128), (3, 6, 1, 32, 40), (2, 19, 19, 96, 260), (8, 38, 38, 64, 128)):
        x = torch.randn(N, H, W, C, generator=g) * torch.tensor([1.0, 30.0, 0.02, 5.0, 1.0, 2.0, 0.5, 1.0][:N]).view(N, 1, 1, 1)
        dy = torch.randn(N, H, W, K, generator=g) * torch.exp(torch.randn(N, H, W, K, generator=g) * 2.0) * 1e-3
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (K, C, 3, 3), dy.permute(0, 3, 1, 2).double(), stride=1,
                                          padding=1).permute(0, 2, 3, 1)
        xw = torch.randn(N, H, W, C + 32, generator=g).cuda()
        xw[..., 16:16 + C] = x.cuda()
        Kp = (K + 3) // 4 * 4
        dyw = torch.zeros(N, H, W, Kp + 8).cuda()
        dyw[..., 4:4 + K] = dy.cuda()
        xv, dyv = ops.View(xw, 16, C), ops.View(dyw, 4, K)
        ax, ady = ops.amax_slots(x.cuda()), ops.amax_slots(dy.cuda())
        err = {}
        for form in ('1', '0'):
            os.environ['PPY_WGRAD9'] = form
            try:
                dw = torch.full((K, 3, 3, C), float('nan'), device='cuda')
                ops.conv2d_wgrad(xv, dyv, dw, 1, 1, amax_x=ax, amax_dy=ady)
                torch.cuda.synchronize()
            finally:
                os.environ.pop('PPY_WGRAD9', None)
            err[form] = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
        print('wgrad 3x3 N%d %dx%d C%d K%d: max error / max|dw| vs float64: nine taps %.2e, one tap per workgroup %.2e'
              % (N, H, W, C, K, err['1'], err['0']))
        assert err['1'] <= max(2.0 * err['0'], 2e-6) and err['1'] <= 1e-5


def test_weight_prep_table_is_bit_identical_to_the_per_layer_splits():
    """Round 3: ppy_train_prepare_weights_f16x2 rebuilds the forward planes and the flipped / transposed data-gradient planes
    of ALL trainable weights in three launches; every buffer must equal what the per-layer calls produce (split_weights_f16x2,
    the workspace of ppy_conv2d_dgrad_f32) -- checked through the planes themselves and through a data gradient on them."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(31)
    shapes = [(258, 1, 1, 512), (64, 3, 3, 96), (512, 3, 3, 32), (40, 1, 1, 64), (1024, 3, 3, 544)]
    ws_ = [(torch.randn(K, R, S, C, generator=g) * torch.exp(torch.randn(K, 1, 1, C, generator=g))).cuda().contiguous() for K, R, S, C in shapes]
    ws_[1][:, :, :, 5] = 0.0                                    # an all-zero input channel: its dgrad row scale comes from the clamp
    entries = [dict(key=i, w=w, dgrad=(i != 3)) for i, w in enumerate(ws_)]
    tab = ops.WeightPrepTable(entries)
    tab.build()
    torch.cuda.synchronize()
    for e, w in zip(entries, ws_):
        K, R, S, C = w.shape
        planes, sc = ops.split_weights_f16x2(w, torch.ones(K).cuda())
        assert torch.equal(planes, e['planes']) and torch.equal(sc, e['scale']), ('forward planes', tuple(w.shape))
        if not e['dgrad']:
            continue
        N, H, W = 2, 9, 7
        pad = (R - 1) // 2
        dy = (torch.randn(N, H, W, K, generator=g) * torch.tensor([1.0, 40.0]).view(N, 1, 1, 1)).cuda()
        amax = ops.amax_slots(dy)
        dx0 = torch.full((N, H, W, C), float('nan'), device='cuda')
        dx1 = torch.full((N, H, W, C), float('nan'), device='cuda')
        ops.conv2d_dgrad(ops.View(dy), w, ops.View(dx0), 1, pad, amax_dy=amax)
        ops.conv2d_dgrad_prepared(ops.View(dy), e, ops.View(dx1), pad, torch.ones(C).cuda(), torch.zeros(C).cuda(), None, amax_dy=amax)
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1), ('data gradient', tuple(w.shape))
    # the table follows its masters: an update in place is picked up by the next build()
    ws_[2].mul_(0.5)
    tab.build()
    planes, sc = ops.split_weights_f16x2(ws_[2], torch.ones(512).cuda())
    assert torch.equal(planes, entries[2]['planes']) and torch.equal(sc, entries[2]['scale']) and tab.current()


def test_dgrad_f16x2_with_tracked_maxima():
    """The data gradient on the f16x2 kernels (dy scaled per image by its tracked maximum, the flipped weights per channel)
    against float64: at the level of the bf16x3 path, incl. a 258-channel dy (padded copy) and images of very different scale."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(23)
    for N, H, W, C, K, R in ((4, 19, 19, 512, 258, 1), (2, 38, 38, 256, 512, 3), (3, 16, 12, 96, 64, 3)):
        pad = (R - 1) // 2
        w = torch.randn(K, C, R, R, generator=g) * (1.0 / (C * R * R)) ** 0.5
        dy = torch.randn(N, K, H, W, generator=g) * torch.tensor([1.0, 300.0, 0.003, 20.0][:N]).view(N, 1, 1, 1) * 1e-3
        ref = torch.nn.grad.conv2d_input((N, C, H, W), w.double(), dy.double(), stride=1, padding=pad).permute(0, 2, 3, 1)
        dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
        wk = w.permute(0, 2, 3, 1).contiguous().cuda()
        err = {}
        for name, kw in (('bf16x3', {}), ('f16x2', dict(amax_dy=ops.amax_slots(dyd)))):
            dx = torch.full((N, H, W, C), float('nan'), device='cuda')
            ops.conv2d_dgrad(ops.View(dyd), wk, ops.View(dx), 1, pad, **kw)
            # per image: every image is scaled on its own
            err[name] = max(float((dx[n].cpu().double() - ref[n]).abs().max() / ref[n].abs().max()) for n in range(N))
        print('dgrad N%d %dx%d C%d K%d R%d: max error / max|dx| per image vs float64: %s' % (N, H, W, C, K, R, err))
        assert err['f16x2'] <= max(2.0 * err['bf16x3'], 2e-6) and err['f16x2'] <= 1e-5


def test_dgrad_refuses_stride2():
    from ppyolo_hip import ops
    from ppyolo_hip._lib import PPYoloHipError
    dy = torch.zeros(1, 4, 4, 32).cuda()
    with pytest.raises(PPYoloHipError, match='outside what the kernels implement|unsupported'):
        ops.conv2d_dgrad(ops.View(dy), torch.zeros(32, 3, 3, 32).cuda(), ops.View(torch.zeros(1, 8, 8, 32).cuda()), 2, 1)


@pytest.mark.parametrize('tag', ['r18vd_96', 'r50vd_96'])
def test_head_conv_gradients_of_a_training_step(golden, tag):
    """One whole training step of the oracle on the g12 inputs (its bit-equality with the REFERENCE's step is asserted on
    the build box by tests/test_train_oracle.py; across CPU models the training-mode BatchNorm of 3x3 maps amplifies
    MKLDNN rounding differences, so here the oracle run of THIS box supplies activations, d loss / d outputs and the
    expected gradients): weight gradients of all output convolutions and of the 3x3 tip convolutions, and the input
    gradient of the output convolutions, from the HIP kernels."""
    from conftest import build_model
    from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
    from oracle import ppyolo_oracle as orc, train_oracle as trn
    from ppyolo_hip import ops, synth
    g = golden('g12_train_' + tag)
    S, N, wseed, iseed, rseed = [int(v) for v in g['meta']]
    cfg = {'r18vd_96': PPYOLO_r18vd_Config, 'r50vd_96': PPYOLO_2x_Config}[tag]()
    _, sd = build_model(cfg, wseed, 'cpu')
    x = synth.synth_images(N, S, seed=iseed)
    L = len(cfg.head['anchor_masks'])
    seen = {}
    real_conv2d = F.conv2d

    def spy(inp, weight, bias=None, stride=1, padding=0, *a, **k):        # every convolution of the step: (input, output)
        out = real_conv2d(inp, weight, bias, stride, padding, *a, **k)
        if weight.requires_grad:
            out.retain_grad()
            seen[id(weight)] = (inp.detach(), out, stride if isinstance(stride, int) else stride[0],
                                padding if isinstance(padding, int) else padding[0], weight)
        return out
    orc.F.conv2d = spy
    try:
        r = trn.train_step(sd, cfg, x, torch.from_numpy(g['gt_bbox']), [torch.from_numpy(g['target%d' % i]) for i in range(L)],
                           rng_seed=rseed)
    finally:
        orc.F.conv2d = real_conv2d
    want_loss = float(g['all_loss'])
    assert abs(float(r['all_loss']) - want_loss) <= 2e-2 * want_loss           # same step as the reference's, up to BN noise
    checked = 0
    leaves = {k: v for k, v in r['grads'].items() if k.endswith('conv.weight')}
    by_grad = {v.data_ptr(): k for k, v in leaves.items()}                    # a conv call's weight leaf -> its state_dict key
    assert len(seen) == len(leaves)
    for inp, out, stride, pad, wleaf in seen.values():
        key = by_grad[wleaf.grad.data_ptr()]
        w = sd[key]
        K, Cin, R, _ = w.shape
        assert tuple(out.shape[1:2]) == (K,) and inp.shape[1] == Cin, key
        dy = out.grad
        dw = torch.zeros(K, R, R, Cin).cuda()
        ops.conv2d_wgrad(ops.View(nhwc(inp).cuda()), ops.View(nhwc(dy).cuda()), dw, stride, pad)
        torch.cuda.synchronize()
        e = _rel(dw.cpu().permute(0, 3, 1, 2), leaves[key])
        assert e <= 3e-5, (key, e)
        if 'yolo_output_convs' in key or 'tip_layers' in key:
            dx = torch.zeros(inp.shape[0], inp.shape[2], inp.shape[3], Cin).cuda()
            ops.conv2d_dgrad(ops.View(nhwc(dy).cuda()), w.permute(0, 2, 3, 1).contiguous().cuda(), ops.View(dx), stride, pad)
            torch.cuda.synchronize()
            dx_ref = F.conv_transpose2d(dy, w, None, stride, pad)
            assert _rel(dx.cpu().permute(0, 3, 1, 2), dx_ref) <= 3e-5, key
        checked += 1
    assert checked == len(leaves) >= 7


def test_dcnv2_backward_matches_the_reference(golden):
    """ppy_dcnv2_backward_f32 against golden g15 = torch autograd through the REFERENCE's DCNv2 module: d w, d (raw conv_offset
    output) directly; d x after adding conv_offset's own data gradient (a plain convolution backward of d offset_mask, taken from
    torch here: it is not part of this entry point)."""
    from ppyolo_hip import ops
    g = golden('g15_dcn_backward')
    T = lambda a: torch.from_numpy(np.asarray(a))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    for i in range(int(g['ncases'])):
        p = 'b%d_' % i
        ci, co, s = [int(v) for v in g[p + 'meta']]
        x, om, dy = T(g[p + 'x']), T(g[p + 'offset_mask']), T(g[p + 'dy'])
        N, _, H, W = x.shape
        xh, omh, dyh = nhwc(x).cuda(), nhwc(om).cuda(), nhwc(dy).cuda()
        w = T(g[p + 'w_dcn']).permute(0, 2, 3, 1).contiguous().cuda()
        dx = torch.full_like(xh, float('nan'))
        dom = torch.full_like(omh, float('nan'))
        dw = torch.full_like(w, float('nan'))
        ops.dcnv2_backward(ops.View(xh), w, ops.View(omh), ops.View(dyh), ops.View(dx), ops.View(dom), dw, s, 1)
        rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
        e_w = rel(dw.permute(0, 3, 1, 2), T(g[p + 'dw_dcn']))
        e_om = rel(dom.permute(0, 3, 1, 2), T(g[p + 'd_offset_mask']))
        # the whole d x of the module = sampling path (ours) + conv_offset path
        via_off = torch.nn.grad.conv2d_input(x.shape, T(g[p + 'w_off']), T(g[p + 'd_offset_mask']), stride=s, padding=1)
        e_x = rel(dx.permute(0, 3, 1, 2) + via_off.cuda(), T(g[p + 'dx']))
        print('DCNv2 backward case %d (C %d -> K %d, stride %d): relative max error  d w %.1e  d offset/mask %.1e  d x %.1e'
              % (i, ci, co, s, e_w, e_om, e_x))
        assert e_w <= 2e-5 and e_om <= 2e-5 and e_x <= 2e-5, (i, e_w, e_om, e_x)
        # clamped taps: exactly zero offset gradient, like torch.clamp's backward
        assert torch.equal(dom.permute(0, 3, 1, 2).cpu()[:, :18] == 0, T(g[p + 'd_offset_mask'])[:, :18] == 0)
