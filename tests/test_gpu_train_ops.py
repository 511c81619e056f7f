"""Training-step operators (csrc/train.hip, csrc/yolo_loss.hip) through the C ABI against torch
autograd on the CPU (= what the reference's backward is), the training oracle (oracle/train_oracle.py) and the gradients the
REFERENCE produced in its own step (golden g12: d loss / d head outputs, loss terms)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('shape,act', [((8, 64, 19, 19), 'leaky'), ((2, 260, 3, 3), None), ((3, 100, 7, 5), 'relu'),
                                       ((8, 32, 152, 152), 'relu'), ((1, 512, 1, 2), 'leaky')])
def test_batchnorm_training_forward_and_backward(shape, act):
    from ppyolo_hip import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * torch.exp(torch.randn(1, C, 1, 1, generator=g)) + torch.randn(1, C, 1, 1, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    dy = torch.randn(N, C, H, W, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    y_ref = F.relu(z) if act == 'relu' else (F.leaky_relu(z, 0.1) if act == 'leaky' else z)
    y_ref.backward(dy)
    xd = nhwc(x).cuda()
    mean, invstd = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    rmd, rvd = rm.cuda(), rv.cuda()
    ops.bn_train_stats(ops.View(xd), 1e-5, 0.1, mean, invstd, rmd, rvd)
    y = torch.full((N, H, W, C), 5.0).cuda()
    ops.bn_train_apply(ops.View(xd), mean, invstd, gamma.cuda(), beta.cuda(), ops.View(y), act)
    dx = torch.zeros(N, H, W, C).cuda()
    dgam, dbet = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    ops.bn_train_bwd(ops.View(xd), ops.View(y), ops.View(nhwc(dy).cuda()), mean, invstd, gamma.cuda(), ops.View(dx), dgam, dbet, act)
    torch.cuda.synchronize()
    assert rel(mean, x.mean((0, 2, 3))) <= 2e-6 and rel(invstd, 1 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)) <= 5e-6
    assert rel(rmd, rm_ref) <= 2e-6 and rel(rvd, rv_ref) <= 5e-6
    assert rel(nchw(y), y_ref.detach()) <= 2e-5
    assert rel(dgam, gr.grad) <= 5e-5 and rel(dbet, br.grad) <= 5e-5, (rel(dgam, gr.grad), rel(dbet, br.grad))
    assert rel(nchw(dx), xr.grad) <= 1e-4, rel(nchw(dx), xr.grad)


def test_upsample_spp_act_backward():
    from oracle import ppyolo_oracle as orc
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 7, 9, generator=g, requires_grad=True)
    dy = torch.randn(2, 64, 14, 18, generator=g)
    F.interpolate(x, scale_factor=2, mode='nearest').backward(dy)
    dx = torch.ones(2, 7, 9, 64).cuda()
    ops.upsample2x_bwd(ops.View(nhwc(dy).cuda()), ops.View(dx), accumulate=True)
    torch.cuda.synchronize()
    assert rel(nchw(dx) - 1.0, x.grad) <= 1e-6
    xs = torch.randn(2, 32, 19, 19, generator=g, requires_grad=True)
    xs.data[0, 0, 3, 3] = xs.data[0, 0, 3, 4] = 9.0                 # an exact tie inside one window: first in scan order wins
    dys = torch.randn(2, 128, 19, 19, generator=g)
    orc.spp(xs).backward(dys)
    dxs = torch.zeros(2, 19, 19, 32).cuda()
    ops.spp_bwd(ops.View(nhwc(xs.detach()).cuda()), ops.View(nhwc(dys).cuda()), ops.View(dxs))
    torch.cuda.synchronize()
    assert rel(nchw(dxs), xs.grad) <= 2e-6
    y = torch.randn(2, 5, 5, 32, generator=g)
    d = torch.randn(2, 5, 5, 32, generator=g)
    out = torch.zeros(2, 5, 5, 32).cuda()
    ops.act_bwd(ops.View(d.cuda()), ops.View(y.cuda()), ops.View(out), 'leaky')
    assert torch.equal(out.cpu(), d * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.1)))


def test_spp_backward_fused_and_fallback_forms():
    """Round 3: ppy_spp_bwd_f32 runs as ONE launch when an image's map fits the LDS (argmax maps built there) and as the
    four-launch gather otherwise: both against torch autograd, with a channel count that leaves a partial 16-channel group,
    ties, and a map too large for the fused form (52 x 52)."""
    from oracle import ppyolo_oracle as orc
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(21)
    for (N, C, H, W) in [(2, 24, 19, 19), (1, 40, 13, 10), (1, 16, 52, 52)]:
        xs = torch.randn(N, C, H, W, generator=g)
        xs[0, 0, 3, 3] = xs[0, 0, 3, 4] = 9.0                       # first in scan order wins
        xs[:, 1] = 0.25                                               # a constant channel: every window is one big tie
        xs.requires_grad_(True)
        dys = torch.randn(N, 4 * C, H, W, generator=g)
        orc.spp(xs).backward(dys)
        dxs = torch.full((N, H, W, C + 8), 5.0).cuda()
        ops.spp_bwd(ops.View(nhwc(xs.detach()).cuda()), ops.View(nhwc(dys).cuda()), ops.View(dxs, 0, C))
        torch.cuda.synchronize()
        assert rel(nchw(dxs[..., :C]), xs.grad) <= 2e-6, (N, C, H, W)
        assert bool((dxs[..., C:] == 5.0).all())


@pytest.mark.parametrize('shape', [(1, 13, 11, 40, 0.85), (2, 19, 19, 512, 0.9), (1, 3, 5, 4, 0.5)])
def test_dropblock_mask_odd_shapes(shape):
    """The tiled mask kernel (one hash per element, seeds + halo in LDS) on shapes that leave partial row blocks / channel
    groups: the same bits as nine hashes per element (replicated in numpy)."""
    from ppyolo_hip import ops
    N, H, W, C, keep = shape
    seed = 987654321
    mask, scale = torch.zeros(N, H, W, C).cuda(), torch.zeros(1).cuda()
    ops.dropblock_mask(mask, scale, keep, seed)
    torch.cuda.synchronize()
    ids = np.arange(N * H * W * C, dtype=np.uint64)
    with np.errstate(over='ignore'):
        key = (np.uint64(seed) ^ (ids * np.uint64(0xD1342543DE82EF95))) + np.uint64(0x9E3779B97F4A7C15)
        key = (key ^ (key >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        key = (key ^ (key >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        key = (key ^ (key >> np.uint64(31))) >> np.uint64(32)
    u = ((key >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(N, H, W, C)
    gamma = np.float32(H * H * (1 - keep)) / np.float32(9 * (H - 2) ** 2)
    seeds = torch.from_numpy((u < gamma).astype(np.float32)).permute(0, 3, 1, 2)
    want = 1.0 - F.max_pool2d(seeds, 3, 1, 1)
    assert torch.equal(nchw(mask.cpu()), want)
    if float(want.sum()) > 0:
        assert abs(float(scale) - mask.numel() / float(want.sum())) <= 1e-6 * float(scale)


def test_dropblock_mask_and_apply():
    """mask = 1 - maxpool3x3(u < gamma) with the counter-based generator (replicated here bit for bit), gamma from the
    reference's formula, scale = numel / sum(mask); forward y = x * mask * scale."""
    from ppyolo_hip import ops
    N, H, W, C, keep, seed = 2, 12, 12, 64, 0.9, 12345
    mask, scale = torch.zeros(N, H, W, C).cuda(), torch.zeros(1).cuda()
    ops.dropblock_mask(mask, scale, keep, seed)
    torch.cuda.synchronize()
    ids = np.arange(N * H * W * C, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        key = (np.uint64(seed) ^ (ids * np.uint64(0xD1342543DE82EF95))) + np.uint64(0x9E3779B97F4A7C15)
        key = (key ^ (key >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        key = (key ^ (key >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        key = (key ^ (key >> np.uint64(31))) >> np.uint64(32)
    u = ((key >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(N, H, W, C)
    gamma = np.float32(H * H * (1 - keep)) / np.float32(9 * (H - 2) ** 2)
    seeds = torch.from_numpy((u < gamma).astype(np.float32)).permute(0, 3, 1, 2)
    want = 1.0 - F.max_pool2d(seeds, 3, 1, 1)
    assert torch.equal(nchw(mask.cpu()), want)
    assert abs(float(scale) - mask.numel() / float(want.sum())) <= 1e-6 * float(scale)
    frac = 1.0 - float(want.mean())
    assert 0.04 < frac < 0.2, frac                                         # ~1 - keep_prob of the activations are dropped
    x = torch.randn(N, H, W, C).cuda()
    y = torch.zeros_like(x)
    ops.dropblock_apply(ops.View(x), mask, scale, ops.View(y))
    assert rel(y, x.cpu() * mask.cpu() * float(scale)) <= 1e-6


def test_sgd_momentum_matches_torch():
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(1000, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, weight_decay=0.0005)
    p, v = p0.clone().cuda(), torch.zeros(1000).cuda()
    for step in range(3):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        ops.sgd_momentum(p, gr.cuda(), v, 0.01, 0.9, 0.0005, first_step=(step == 0))
    torch.cuda.synchronize()
    assert rel(p, ref.detach()) <= 1e-6


@pytest.mark.parametrize('tag', ['r18vd_96', 'r50vd_96'])
def test_loss_and_dout_match_the_reference_step(golden, tag):
    """Golden g12: the REFERENCE's head outputs of its training step -> its loss terms and d loss / d outputs."""
    from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
    from ppyolo_hip import ops
    g = golden('g12_train_' + tag)
    cfg = {'r18vd_96': PPYOLO_r18vd_Config, 'r50vd_96': PPYOLO_2x_Config}[tag]()
    hc = cfg.head
    L = len(hc['anchor_masks'])
    loss6 = torch.zeros(6).cuda()
    gt = torch.from_numpy(g['gt_bbox']).cuda()
    for i in range(L):
        out = torch.from_numpy(g['out%d' % i])
        N, nch, S, _ = out.shape
        ld = (nch + 31) // 32 * 32
        ob = torch.zeros(N, S, S, ld).cuda()
        ob[..., :nch] = nhwc(out).cuda()
        db = torch.full((N, S, S, ld), 0.0).cuda()
        anchors = [hc['anchors'][m] for m in hc['anchor_masks'][i]]
        ops.yolov3_loss(ops.View(ob, 0, nch), torch.from_numpy(g['target%d' % i]).cuda(), gt, anchors, 80, hc['downsample'][i],
                        cfg.yolo_loss['scale_x_y'], cfg.yolo_loss['ignore_thresh'], cfg.iou_loss['loss_weight'], hc['iou_aware'],
                        cfg.iou_aware_loss['loss_weight'] if hc['iou_aware'] else 0.0, ops.View(db, 0, nch), loss6, accumulate=i > 0,
                        amax_dout=(amax := ops.amax_slots(device='cuda', N=N)))
        torch.cuda.synchronize()
        # the tracked per-image maximum of the gradient (operand scale of the f16x2 gradient kernels) is exactly max|dout|
        assert torch.equal(amax.view(N, -1).amax(dim=1), db.reshape(N, -1).abs().amax(dim=1))
        want = torch.from_numpy(g['dout%d' % i])
        e = rel(nchw(db[..., :nch]), want)
        assert e <= 2e-5, 'level %d: d loss / d head output off by %.3e of its maximum' % (i, e)
        # and element by element where the gradient is not tiny
        got = nchw(db[..., :nch]).cpu()
        big = want.abs() > 1e-4 * want.abs().max()
        assert ((got[big] - want[big]).abs() / want[big].abs()).max() <= 2e-3
    names = [str(n) for n in g['loss_names']]
    mine = dict(zip(['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware'], loss6.cpu().tolist()))
    for nme, val in zip(names, g['loss_values']):
        assert abs(mine[nme] - float(val)) <= 2e-5 * abs(float(val)), (nme, mine[nme], float(val))


@pytest.mark.parametrize('square', [True, False])
def test_loss_plain_yolov3_branch_and_ignore_mask(square):
    """scale_x_y = 1 (cross-entropy on x / y), gts that overlap predictions above the ignore threshold, random targets:
    against the training oracle's autograd.  square: IouLoss(loss_square=) -- 1 - iou^2 (the configurations) / 1 - iou
    (reference model/iou_losses.py:66-70)."""
    from config import PPYOLO_r18vd_Config
    from oracle import train_oracle as trn
    from ppyolo_hip import ops
    cfg = PPYOLO_r18vd_Config()
    cfg.yolo_loss = dict(cfg.yolo_loss, scale_x_y=1.0, ignore_thresh=0.3)
    cfg.iou_loss = dict(cfg.iou_loss, loss_square=square)
    cfg.head = dict(cfg.head, anchor_masks=[[3, 4, 5]], downsample=[32])
    g = torch.Generator().manual_seed(8)
    N, S, an, C = 3, 6, 3, 80
    out = (torch.randn(N, an * 85, S, S, generator=g) * 0.7).requires_grad_(True)
    tgt = torch.zeros(N, an, 6 + C, S, S)
    gt = torch.zeros(N, 50, 4)
    for n in range(N):
        for j in range(4):
            a, hh, ww = int(torch.randint(0, an, (1,), generator=g)), int(torch.randint(0, S, (1,), generator=g)), int(torch.randint(0, S, (1,), generator=g))
            tgt[n, a, 0:2, hh, ww] = torch.rand(2, generator=g)
            tgt[n, a, 2:4, hh, ww] = torch.randn(2, generator=g) * 0.3
            tgt[n, a, 4, hh, ww] = 1.5
            tgt[n, a, 5, hh, ww] = 0.7 if j == 0 else 1.0
            tgt[n, a, 6 + int(torch.randint(0, C, (1,), generator=g)), hh, ww] = 1.0
            gt[n, j] = torch.tensor([(ww + 0.5) / S, (hh + 0.5) / S, 0.3, 0.25])
    losses = trn.yolov3_loss([out], [tgt], gt, cfg)
    sum(losses.values()).backward()
    ob, db, loss6 = nhwc(out.detach()).cuda(), torch.zeros(N, S, S, an * 85).cuda(), torch.zeros(6).cuda()
    ops.yolov3_loss(ops.View(ob), tgt.cuda(), gt.cuda(), [cfg.head['anchors'][m] for m in (3, 4, 5)], C, 32, 1.0, 0.3,
                    cfg.iou_loss['loss_weight'], False, 0.0, ops.View(db), loss6, iou_loss_square=square)
    torch.cuda.synchronize()
    assert rel(nchw(db), out.grad) <= 2e-5
    for j, nme in enumerate(['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou']):
        assert abs(float(loss6[j]) - float(losses[nme])) <= 2e-5 * abs(float(losses[nme])), nme


def test_pool_backward_and_strided_data_gradient():
    """Backward pieces of a training backbone (freeze_at < 5) against torch autograd: AvgPool2d(2, 2) (odd sizes: floor mode),
    MaxPool2d(3, 2, 1) incl. ties (first maximum in scan order), and the data gradient of a STRIDED convolution as the
    stride-1 data gradient of the zero-inserted output gradient."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(8)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    for N, C, H, W in ((2, 8, 6, 6), (1, 4, 7, 5), (3, 12, 9, 10)):
        x = torch.randn(N, C, H, W, generator=g).requires_grad_()
        y = F.avg_pool2d(x, 2, 2)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        dx = torch.full((N, H, W, C), float('nan'), device='cuda')
        ops.avgpool2x2_bwd(ops.View(nhwc(dy).cuda()), ops.View(dx))
        assert torch.equal(dx.cpu(), nhwc(x.grad)), (N, C, H, W)
        xm = torch.randint(-2, 3, (N, C, H, W), generator=g).float().requires_grad_()        # many exact ties
        ym = F.max_pool2d(xm, 3, 2, 1)
        dym = torch.randn(ym.shape, generator=g)
        ym.backward(dym)
        dxm = torch.full((N, H, W, C), float('nan'), device='cuda')
        ops.maxpool3x3s2_bwd(ops.View(nhwc(xm.detach()).cuda()), ops.View(nhwc(dym).cuda()), ops.View(dxm))
        assert torch.allclose(dxm.cpu(), nhwc(xm.grad), rtol=0, atol=1e-6), (N, C, H, W)
    for N, C, K, H, W, R, s in ((2, 32, 27, 10, 12, 3, 2), (1, 64, 64, 9, 9, 3, 2), (2, 32, 64, 8, 8, 1, 2)):
        pad = (R - 1) // 2
        w = torch.randn(K, C, R, R, generator=g) * 0.1
        Ho, Wo = (H + 2 * pad - R) // s + 1, (W + 2 * pad - R) // s + 1
        dy = torch.randn(N, K, Ho, Wo, generator=g)
        want = torch.nn.grad.conv2d_input((N, C, H, W), w, dy, stride=s, padding=pad)
        H1, W1 = H + 2 * pad - R + 1, W + 2 * pad - R + 1
        Kp = (K + 31) // 32 * 32
        up = torch.zeros(N, H1, W1, Kp, device='cuda')
        ops.zero_insert(ops.View(nhwc(dy).cuda()), ops.View(up, 0, K), s)
        dx = torch.empty(N, H, W, C, device='cuda')
        ops.conv2d_dgrad(ops.View(up, 0, K), w.permute(0, 2, 3, 1).contiguous().cuda(), ops.View(dx), 1, pad)
        err = float((dx.cpu() - nhwc(want)).abs().max() / want.abs().max())
        assert err <= 2e-6, ((N, C, K, H, W, R, s), err)


def test_last_workgroup_finalisation_is_bit_identical_to_the_second_launch():
    """Round 3: the BatchNorm statistics merge and the BatchNorm backward sums finish inside their reduction kernels (the
    workgroup that draws the last ticket of a column block runs the final step: csrc/train.hip, last_ticket) instead of in a
    second 6 us launch.  Same code, same order: every output equals the two-launch form (PPY_BN_FUSE_FINAL=0) bit for bit --
    repeated, so that a counter that did not wrap back to zero would show."""
    import os
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(77)

    def both(fn):
        out = {}
        for form in ('1', '0', '1'):
            os.environ['PPY_BN_FUSE_FINAL'] = form
            try:
                r = fn()
                torch.cuda.synchronize()
            finally:
                os.environ.pop('PPY_BN_FUSE_FINAL', None)
            if form in out:
                assert all(torch.equal(a, b) for a, b in zip(out[form], r)), 'second fused run differs from the first'
            out[form] = r
        assert all(torch.equal(a, b) for a, b in zip(out['1'], out['0']))

    # backward sums: channel counts that leave partial 64- and 16-channel blocks, one- and many-slice maps
    for (N, C, H, W, act) in ((8, 256, 38, 38, 'leaky'), (2, 100, 7, 5, 'relu'), (1, 520, 3, 3, None), (8, 64, 76, 76, 'leaky')):
        x = torch.randn(N, H, W, C, generator=g).cuda()
        dy = torch.randn(N, H, W, C, generator=g).cuda()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()

        def bwd():
            mean, invstd = torch.empty(C).cuda(), torch.empty(C).cuda()
            ops.bn_train_stats(ops.View(x), 1e-5, 0.1, mean, invstd)
            y = torch.empty_like(x)
            ops.bn_train_apply(ops.View(x), mean, invstd, gamma, beta, ops.View(y), act)
            dx, dg, db = torch.empty_like(x), torch.empty(C).cuda(), torch.empty(C).cuda()
            ops.bn_train_bwd(ops.View(x), ops.View(y), ops.View(dy), mean, invstd, gamma, ops.View(dx), dg, db, act)
            return dx, dg, db
        both(bwd)
    # statistics from a convolution's epilogue with more than 256 slices (the two-level merge)
    N, H, W, C, K = 8, 76, 76, 32, 64
    xd = (torch.randn(N, H, W, C, generator=g) + 0.3).cuda()
    wk = (torch.randn(K, 3, 3, C, generator=g) * 0.06).cuda()
    wf = ops.split_weights_f16x2(wk, torch.ones(K).cuda())
    bias = torch.randn(K, generator=g).cuda()

    def stats():
        y = torch.empty(N, H, W, K).cuda()
        part = torch.full((ops.conv2d_bn_partials_bytes(N * H * W, K) // 4,), float('nan')).cuda()
        slices = ops.conv2d_train_fwd(ops.View(xd), wk, wf, bias, ops.View(y), 1, 1, 44, ops.amax_slots(xd), part)
        assert slices > 256
        m, i = torch.empty(K).cuda(), torch.empty(K).cuda()
        rm, rv = torch.full((K,), 0.25).cuda(), torch.full((K,), 2.0).cuda()
        ops.bn_train_stats_merge(part, slices, 1e-5, 0.1, m, i, rm, rv)
        return m, i, rm, rv
    both(stats)


def test_frozen_1x1_layers_without_the_raw_tensor():
    """Round 4: ppy_conv1x1_stats_f32 + ppy_conv1x1_bn_apply_f32 (the streaming kernel run twice: statistics only, then the
    BatchNorm applied to its own accumulators) against ppy_conv2d_train_fwd_f32 + ppy_bn_train_apply_f32 on the same kernel:
    EQUAL partials, EQUAL y, equal tracked maxima -- C = 64 and C = 128, both grids, with / without shortcut, every
    activation, rows that do not fill the last tile, images of very different magnitude, y as a channel slice."""
    from ppyolo_hip import ops
    g = torch.Generator().manual_seed(5200)
    first = ops.stream_first_cfg()
    for N, H, W, C, K, act, use_res, variant, wide in ((3, 24, 20, 64, 256, 'relu', True, 0, False), (2, 9, 7, 128, 128, None, False, 0, False),
                                                      (2, 19, 19, 64, 64, 'leaky', False, 1, True), (4, 13, 11, 128, 512, 'relu', True, 1, False),
                                                      (8, 38, 38, 64, 256, 'relu', True, 0, False)):
        x = torch.randn(N, H, W, C, generator=g) * torch.exp(1.5 * torch.randn(N, 1, 1, 1, generator=g)) + 0.3
        wk = (torch.randn(K, 1, 1, C, generator=g) * (1.0 / C ** 0.5)).cuda()
        bias = torch.zeros(K).cuda()
        gamma, beta = (torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda()
        res = torch.randn(N, H, W, K, generator=g).cuda() if use_res else None
        xd = x.cuda()
        wf = ops.split_weights_f16x2(wk, torch.ones(K).cuda())
        M = N * H * W
        # two tensors: raw output + statistics, then the apply pass
        raw = torch.empty(N, H, W, K).cuda()
        part0 = torch.full((ops.conv2d_bn_partials_bytes(M, K) // 4,), float('nan')).cuda()
        s0 = ops.conv2d_train_fwd(ops.View(xd), wk, wf, bias, ops.View(raw), 1, 0, first + variant, ops.amax_slots(xd), part0)
        m0, i0 = torch.empty(K).cuda(), torch.empty(K).cuda()
        ops.bn_train_stats_merge(part0.clone(), s0, 1e-5, 0.1, m0, i0)
        y0 = torch.empty(N, H, W, K).cuda()
        am0 = ops.amax_slots(N=N, device=y0.device)
        ops.bn_train_apply(ops.View(raw), m0, i0, gamma, beta, ops.View(y0), act, None if res is None else ops.View(res), am0)
        # none
        part1 = torch.full((ops.conv2d_bn_partials_bytes(M, K) // 4,), float('nan')).cuda()
        s1 = ops.conv1x1_stats(ops.View(xd), wf, bias, K, variant, ops.amax_slots(xd), part1)
        assert s1 == s0 and torch.equal(part0[:s0 * K * 3], part1[:s1 * K * 3])
        m1, i1 = torch.empty(K).cuda(), torch.empty(K).cuda()
        ops.bn_train_stats_merge(part1, s1, 1e-5, 0.1, m1, i1)
        assert torch.equal(m0, m1) and torch.equal(i0, i1)
        buf = torch.full((N, H, W, 2 * K if wide else K), 5.0).cuda()
        yv = ops.View(buf, K, K) if wide else ops.View(buf)
        am1 = ops.amax_slots(N=N, device=y0.device)
        ops.conv1x1_bn_apply(ops.View(xd), wf, bias, m1, i1, gamma, beta, yv, act, None if res is None else ops.View(res), variant,
                             ops.amax_slots(xd), am1)
        torch.cuda.synchronize()
        what = 'N%d %dx%d C%d K%d %s res %s variant %d' % (N, H, W, C, K, act, use_res, variant)
        got = buf[..., K:] if wide else buf
        assert torch.equal(got, y0), '%s: %d values differ, max %.3e' % (what, int((got != y0).sum()), float((got - y0).abs().max()))
        if wide:
            assert bool((buf[..., :K] == 5.0).all()), what
        assert torch.equal(am0.view(N, -1).max(dim=1).values, am1.view(N, -1).max(dim=1).values), what + ': tracked maxima'


def test_batchnorm_statistics_from_the_convolution_epilogue():
    """ppy_conv2d_train_fwd_f32 + ppy_bn_train_stats_merge_f32 against the two-kernel form (ppy_conv2d_bn_act_f32, then
    ppy_bn_train_stats_f32 reading y back): y is EQUAL (same tile, same epilogue), mean / invstd / running statistics agree to
    rounding (both are two-pass per slice + Chan merges; the slices differ) and with torch's batch statistics of y -- every
    f16x2 kernel family (tiles incl. slab / 96-row variants, streaming 1x1, stem patch, specialised waves), rows that do not fill the last tile, K = 258 (scalar epilogue, a partial column tile), a
    bias, per-image scales far apart, and a map large enough for the two-level merge (> 256 slices)."""
    from ppyolo_hip import ops
    from ppyolo_hip._lib import PPYoloHipError
    g = torch.Generator().manual_seed(5100)
    wsf = ops.ws_first_cfg()
    for N, H, W, C, K, R, stride, cfgs in ((2, 19, 19, 64, 128, 3, 1, (41, 44, 47, 48, 51, 66, 70, wsf, wsf + 2)), (3, 13, 11, 96, 258, 1, 1, (41, 44, wsf + 1)),
                                           (8, 76, 76, 32, 64, 3, 1, (44, wsf + 3, ops.patch_first_cfg())), (5, 9, 7, 160, 40, 3, 2, (48, wsf)),
                                           (1, 4, 4, 32, 32, 1, 1, (41,)), (3, 21, 37, 32, 32, 3, 1, (ops.patch_first_cfg(), 85)),
                                           (3, 24, 20, 64, 256, 1, 1, (ops.stream_first_cfg(), ops.stream_first_cfg() + 1, 88)),
                                           (2, 9, 7, 128, 128, 1, 1, (ops.stream_first_cfg(),))):
        pad = (R - 1) // 2
        x = torch.randn(N, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(N, 1, 1, 1, generator=g)) + 0.3
        w = torch.randn(K, C, R, R, generator=g) * (1.0 / (R * R * C) ** 0.5)
        bias = torch.randn(K, generator=g).cuda()
        one = torch.ones(K).cuda()
        wk = w.permute(0, 2, 3, 1).contiguous().cuda()
        xd = nhwc(x).cuda()
        wf = ops.split_weights_f16x2(wk, one)
        Ho, Wo = ops.conv_out_hw(H, W, R, R, stride, pad)
        M = N * Ho * Wo
        for cfg in cfgs:
            y0 = torch.full((N, Ho, Wo, K), 9.0).cuda()
            ops.conv2d_bn_act(ops.View(xd), wk, one, bias, ops.View(y0), stride, pad, None, cfg=cfg, splitk=1, w_f16=wf, amax_in=ops.amax_slots(xd))
            m0, i0 = torch.empty(K).cuda(), torch.empty(K).cuda()
            rm0, rv0 = torch.full((K,), 0.25).cuda(), torch.full((K,), 2.0).cuda()
            ops.bn_train_stats(ops.View(y0), 1e-5, 0.1, m0, i0, rm0, rv0)
            y1 = torch.full((N, Ho, Wo, K), 9.0).cuda()
            part = torch.full((ops.conv2d_bn_partials_bytes(M, K) // 4,), float('nan')).cuda()
            slices = ops.conv2d_train_fwd(ops.View(xd), wk, wf, bias, ops.View(y1), stride, pad, cfg, ops.amax_slots(xd), part)
            m1, i1 = torch.empty(K).cuda(), torch.empty(K).cuda()
            rm1, rv1 = torch.full((K,), 0.25).cuda(), torch.full((K,), 2.0).cuda()
            ops.bn_train_stats_merge(part, slices, 1e-5, 0.1, m1, i1, rm1, rv1)
            torch.cuda.synchronize()
            what = 'N%d %dx%d C%d K%d R%d s%d cfg %d (%d slices)' % (N, H, W, C, K, R, stride, cfg, slices)
            assert slices > 0 and torch.equal(y0, y1), what
            yy = y1.reshape(-1, K).double()
            assert rel(m1.double(), yy.mean(0)) < 2e-6, what
            assert rel(i1.double(), 1.0 / torch.sqrt(yy.var(0, unbiased=False) + 1e-5)) < 2e-6, what
            assert rel(m1, m0) < 2e-6 and rel(i1, i0) < 2e-6 and rel(rm1, rm0) < 2e-6 and rel(rv1, rv0) < 2e-6, what
    # another kernel family refuses (the caller then uses the two-kernel form)
    x = torch.randn(1, 8, 8, 64).cuda()
    wk = torch.randn(64, 1, 1, 64).cuda()
    one = torch.ones(64).cuda()
    for cfg in (19, 33):          # exact-fp32 MFMA, bf16x3
        with pytest.raises(PPYoloHipError):
            ops.conv2d_train_fwd(ops.View(x), wk, ops.split_weights_f16x2(wk, one), one, ops.View(torch.empty(1, 8, 8, 64).cuda()), 1, 0,
                                 cfg, ops.amax_slots(x), torch.empty(ops.conv2d_bn_partials_bytes(64, 64) // 4).cuda())
