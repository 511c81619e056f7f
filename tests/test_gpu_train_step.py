"""The whole training step on the GPU (ppyolo_hip/train.py: training-mode forward of the network, YOLOv3Loss, backward through
the head, SGD) against the CPU training oracle (oracle/train_oracle.py, which is bit-equal to the reference's step on the
build box: tests/test_train_oracle.py) on the same inputs and the same DropBlock masks."""
import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from oracle import ppyolo_oracle as orc, train_oracle as trn
from ppyolo_hip import synth

pytestmark = pytest.mark.gpu


def synth_targets(cfg, N, S, seed):
    """Targets in the reference's Gt2YoloTarget layout with a few positive cells per level, and matching gt boxes."""
    g = torch.Generator().manual_seed(seed)
    hc = cfg.head
    gt = torch.zeros(N, 50, 4)
    targets = []
    for i, mask in enumerate(hc['anchor_masks']):
        Sg = S // hc['downsample'][i]
        t = torch.zeros(N, len(mask), 6 + 80, Sg, Sg)
        for n in range(N):
            for j in range(2):
                a, hh, ww = [int(torch.randint(0, m, (1,), generator=g)) for m in (len(mask), Sg, Sg)]
                t[n, a, 0:2, hh, ww] = torch.rand(2, generator=g)
                t[n, a, 2:4, hh, ww] = torch.randn(2, generator=g) * 0.3
                t[n, a, 4, hh, ww] = 2.0 - float(torch.rand(1, generator=g)) * 0.5
                t[n, a, 5, hh, ww] = 1.0 if j else 0.6
                t[n, a, 6 + int(torch.randint(0, 80, (1,), generator=g)), hh, ww] = 1.0
                # small boxes: every prediction stays far below the ignore threshold, so the (discontinuous) ignore mask cannot
                # flip on the 1e-3 differences between two fp32 evaluations of the network (the mask itself is tested in
                # tests/test_gpu_train_ops.py)
                gt[n, 2 * i + j] = torch.tensor([(ww + 0.5) / Sg, (hh + 0.5) / Sg, 0.02 + 0.01 * i, 0.03])
        targets.append(t)
    return gt, targets


def relmax(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('cfgc,S', [(PPYOLO_r18vd_Config, 256)])
def test_train_step_matches_the_oracle(cfgc, S):
    # (R50vd: test_train_step_three_way / test_train_step_full_size below -- its training-mode forward amplifies fp32 rounding
    # 1000x on the way to the head, in the reference as much as here, so it is held to the reference's own distance from float64)
    from ppyolo_hip.train import TrainStep
    cfg = cfgc()
    N = 4
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, S, seed=11)
    gt, targets = synth_targets(cfg, N, S, 5)
    # oracle run, recording the DropBlock masks it drew
    masks, real = [], orc.drop_block_train

    def spy(t, block_size=3, keep_prob=0.9):
        y = real(t, block_size, keep_prob)
        with torch.no_grad():
            keep = (y != 0) | (t == 0)
            masks.append(keep.float())
        return y
    orc.drop_block_train = spy
    try:
        torch.set_num_threads(16)
        r = trn.train_step(sd, cfg, x, gt, targets, rng_seed=99)
    finally:
        orc.drop_block_train = real
    ts = TrainStep(model, cfg)
    # the loss GRADIENT is taken from the oracle: |x - t| terms have kinks, and two fp32 evaluations of the ~80-layer network
    # that differ by 1e-3 disagree on a sign now and then, which would change every upstream gradient by tens of percent
    # (the loss kernel is compared on identical head outputs in tests/test_gpu_train_ops.py, against the reference's own numbers)
    loss6 = ts.forward_backward(x.cuda(), gt.cuda(), [t.cuda() for t in targets], dropblock_masks=masks,
                                inject_douts=[d.clone() for d in r['douts']])
    torch.cuda.synchronize()
    names = ['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware']
    for j, nme in enumerate(names):
        if nme in r['losses']:
            want = float(r['losses'][nme])
            assert abs(float(loss6[j]) - want) <= 2e-3 * abs(want), (nme, float(loss6[j]), want)
    for i, o in enumerate(ts.outs):
        e = relmax(o.dense_nchw(), r['outs'][i].detach())
        print('head output %d: max error %.3e of the maximum' % (i, e))
        # (training-mode BatchNorm over a few dozen samples per channel amplifies rounding differences between two fp32
        # evaluations of the ~60 layers in front; the same inputs through two CPU models differ by as much)
        assert e <= 1e-2, 'head output %d: %.3e' % (i, e)
    grads = ts.grads()
    assert set(grads) == set(r['grads'])
    worst = {}
    for k, g in grads.items():
        worst[k] = relmax(g, r['grads'][k])
    print('max relative gradient error over %d tensors: %.3e (median %.3e)' % (len(worst), max(worst.values()), float(np.median(list(worst.values())))))
    # (median 1e-5 .. 3e-4 depending on the rounding of the BatchNorm statistics; a single LeakyReLU slope that flips on an
    # element within rounding of 0 moves a 256-term sum by 1e-2)
    assert float(np.median(list(worst.values()))) <= 1e-3 and max(worst.values()) <= 3e-2, {k: v for k, v in worst.items() if v > 2e-3}
    # BatchNorm running statistics moved the same way
    for k in ('backbone.stage1_conv1_1.bn.running_mean', 'backbone.stage1_conv1_1.bn.running_var'):
        assert relmax(model.state_dict()[k], r['state'][k]) <= 1e-4


def test_r50_head_backward_strict():
    """The R50vd head (CoordConv, SPP, DropBlock, three levels, two-consumer routes, IoU-aware outputs) forward + backward on
    WELL-CONDITIONED backbone features (random, not through the 53 backbone layers): head outputs, d loss / d outputs and every
    parameter gradient against the oracle at fp32 tolerance."""
    from ppyolo_hip.train import TrainStep, Act
    cfg = PPYOLO_2x_Config()
    N, S = 4, 256
    model, sd = build_model(cfg, 0, 'cuda')
    g = torch.Generator().manual_seed(21)
    feats = [torch.relu(torch.randn(N, c, S // d, S // d, generator=g)) for c, d in ((512, 8), (1024, 16), (2048, 32))]
    gt, targets = synth_targets(cfg, N, S, 5)
    masks, real = [], orc.drop_block_train

    def spy(t, block_size=3, keep_prob=0.9):
        y = real(t, block_size, keep_prob)
        with torch.no_grad():
            masks.append(((y != 0) | (t == 0)).float())
        return y
    orc.drop_block_train = spy
    state = {k: v.clone() for k, v in sd.items()}
    for k in trn.trainable_keys(sd):
        state[k].requires_grad_(True)
    orc.TRAIN_MODE[0] = True
    try:
        torch.manual_seed(3)
        outs = orc.head_outputs(state, feats, cfg.head)
    finally:
        orc.TRAIN_MODE[0] = False
        orc.drop_block_train = real
    for o in outs:
        o.retain_grad()
    losses = trn.yolov3_loss(outs, targets, gt, cfg)
    sum(losses.values()).backward()
    ts = TrainStep(model, cfg)
    ts.tape = []
    ts.masks = list(masks)
    fa = [Act(f.permute(0, 2, 3, 1).contiguous().cuda()) for f in feats]
    loss6 = ts.head_loss_backward(fa, gt.cuda(), [t.cuda() for t in targets])
    torch.cuda.synchronize()
    for j, nme in enumerate(['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware']):
        want = float(losses[nme].detach())
        assert abs(float(loss6[j]) - want) <= 1e-4 * abs(want), (nme, float(loss6[j]), want)
    for i, o in enumerate(ts.outs):
        assert relmax(o.dense_nchw(), outs[i].detach()) <= 2e-5 and relmax(o.g.dense_nchw(), outs[i].grad) <= 1e-4, i
    grads = ts.grads()
    worst = {k: relmax(v, state[k].grad) for k, v in grads.items()}
    print('R50 head, %d tensors: max relative gradient error %.3e' % (len(worst), max(worst.values())))
    # (measured: median 4e-4, max 1e-2 of a tensor's largest entry -- LeakyReLU slopes of the few elements that sit within rounding of 0)
    assert max(worst.values()) <= 3e-2 and float(np.median(list(worst.values()))) <= 2e-3, {k: v for k, v in worst.items() if v > 3e-2}


def test_sgd_step_changes_the_model_and_loss_goes_down():
    """Three steps on one batch: the loss falls, parameters move, sync_to_model makes the inference path see them."""
    from ppyolo_hip.train import TrainStep, lr_at
    cfg = PPYOLO_r18vd_Config()
    N, S = 4, 256
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, S, seed=11).cuda()
    gt, targets = synth_targets(cfg, N, S, 5)
    gt, targets = gt.cuda(), [t.cuda() for t in targets]
    ts = TrainStep(model, cfg)
    w0 = model.state_dict()['head.yolo_output_convs.0.conv.weight'].clone()
    b0 = model.state_dict()['head.yolo_output_convs.0.conv.bias'].clone()
    losses = []
    for it in range(4):
        l6 = ts.step(x, gt, targets, 0.002)
        losses.append(float(l6.sum()))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    ts.sync_to_model()
    sdn = model.state_dict()
    assert not torch.equal(sdn['head.yolo_output_convs.0.conv.weight'], w0) and not torch.equal(sdn['head.yolo_output_convs.0.conv.bias'], b0)
    assert torch.equal(sdn['backbone.stage2_0.conv1.conv.weight'].cpu(), sd['backbone.stage2_0.conv1.conv.weight'])      # frozen
    assert abs(lr_at(2000, cfg) - 0.5 * cfg.learningRate['base_lr']) < 1e-12 and lr_at(10 ** 7, cfg) < cfg.learningRate['base_lr'] * 0.011
    model.eval()
    model.head.set_dropblock(is_test=True)
    out = model(x, synth.synth_im_size(N).cuda())
    assert len(out) == N


def test_weight_gradients_on_the_side_stream_change_nothing(monkeypatch):
    """The head's weight gradients run on a second stream beside the data-gradient chain (train.py:_wgrad): same kernels on the
    same operands, so gradients, parameters and loss after two steps are EQUAL to the in-line form's, bit for bit -- a missing
    wait (an operand overwritten early, the optimizer reading an unfinished gradient) would show here."""
    from ppyolo_hip.train import TrainStep
    cfg = PPYOLO_r18vd_Config()
    N, S = 4, 256
    x = synth.synth_images(N, S, seed=11).cuda()
    gt, targets = synth_targets(cfg, N, S, 5)
    gt, targets = gt.cuda(), [t.cuda() for t in targets]
    got = {}
    for side in ('0', '1', '1'):
        monkeypatch.setenv('PPYOLO_HIP_TRAIN_WGRAD_STREAM', side)
        model, _ = build_model(cfg, 0, 'cuda')
        ts = TrainStep(model, cfg)
        assert ts._wgrad_side == (side == '1')
        losses = [ts.step(x, gt, targets, 0.002).clone() for _ in range(3)]
        torch.cuda.synchronize()
        res = (torch.stack(losses).cpu(), ts.gflat.clone().cpu(), ts.pflat.clone().cpu())
        if side in got:
            for a, b in zip(got[side], res):
                assert torch.equal(a, b), 'the step is not repeatable with the side stream'
        got[side] = res
    for a, b in zip(got['0'], got['1']):
        assert torch.equal(a, b)


@pytest.mark.parametrize('cfgc', [PPYOLO_r18vd_Config, PPYOLO_2x_Config])
def test_prefetched_backbone_changes_nothing(cfgc):
    """Round 5 (train.py prefetch_backbone / step(next_x=)): with the backbone frozen the NEXT batch's backbone forward runs on a third
    stream beside this batch's head, loss and backward.  Four iterations over DIFFERENT batches, plain and pipelined: losses,
    gradients, parameters and every BatchNorm running statistic and step counter are EQUAL bit for bit; a batch tensor modified
    after its prefetch is recomputed; stages that train (freeze_at < 5) ignore next_x."""
    from ppyolo_hip.train import TrainStep
    cfg = cfgc()
    N, S = 2, 256
    xs = [synth.synth_images(N, S, seed=20 + i).cuda() for i in range(4)]
    gts = [synth_targets(cfg, N, S, 5 + i) for i in range(4)]
    gts = [(g.cuda(), [t.cuda() for t in tg]) for g, tg in gts]
    got = {}
    for mode in ('plain', 'pipe', 'pipe'):
        model, _ = build_model(cfg, 0, 'cuda')
        ts = TrainStep(model, cfg)
        losses, used = [], 0
        for i in range(4):
            nxt = xs[i + 1] if (mode == 'pipe' and i + 1 < 4) else None
            had = ts._pref is not None
            losses.append(ts.step(xs[i], gts[i][0], gts[i][1], 0.002, next_x=nxt).clone())
            used += int(had)
        torch.cuda.synchronize()
        assert used == (3 if mode == 'pipe' else 0)
        ts.sync_to_model()
        sd = model.state_dict()
        stats = torch.cat([sd[k].double().reshape(-1).cpu() for k in sorted(sd) if 'running_' in k or 'num_batches' in k])
        res = (torch.stack(losses).cpu(), ts.gflat.clone().cpu(), ts.pflat.clone().cpu(), stats)
        if mode in got:
            for a, b in zip(got[mode], res):
                assert torch.equal(a, b), 'the pipelined loop is not repeatable'
        got[mode] = res
    for a, b in zip(got['plain'], got['pipe']):
        assert torch.equal(a, b), 'the pipelined loop differs from the plain one'
    # a batch modified after its prefetch must not be served from the stale features
    model, _ = build_model(cfg, 0, 'cuda')
    ts = TrainStep(model, cfg)
    xa, xb = xs[0].clone(), xs[1].clone()
    ts.step(xa, gts[0][0], gts[0][1], 0.002, next_x=xb)
    xb.mul_(0.5)
    with pytest.warns(RuntimeWarning, match='prefetched backbone forward was discarded'):
        l_mod = ts.step(xb, gts[1][0], gts[1][1], 0.002).clone()
    model2, _ = build_model(cfg, 0, 'cuda')
    ts2 = TrainStep(model2, cfg)
    ts2.step(xs[0], gts[0][0], gts[0][1], 0.002)
    # (the discarded prefetch has advanced the backbone's running statistics once more, as a real extra forward would: losses of
    # the recomputed batch depend on batch statistics only, so they are those of a plain run on the modified tensor)
    l_ref = ts2.step(xs[1] * 0.5, gts[1][0], gts[1][1], 0.002).clone()
    torch.cuda.synchronize()
    assert torch.equal(l_mod.cpu(), l_ref.cpu())


def test_prefetched_backbone_is_ordered_behind_a_main_stream_backbone():
    """Round-5 advisor (medium): on step 0 (and after any discarded prefetch) the step runs its own backbone on the main stream; the
    next batch's prefetched backbone must wait for THAT one -- both read-modify-write the BatchNorm running statistics, and step 0
    also makes the frozen-weight caches.  The small case above is host-bound and cannot see an overlap; this one is GPU-bound
    (R50vd, 8 x 512 x 512: the backbone alone is several ms), so an unordered side stream would run beside step 0's backbone."""
    from ppyolo_hip.train import TrainStep
    cfg = PPYOLO_2x_Config()
    N, S = 8, 512
    xs = [synth.synth_images(N, S, seed=40 + i).cuda() for i in range(3)]
    gts = [synth_targets(cfg, N, S, 9 + i) for i in range(3)]
    gts = [(g.cuda(), [t.cuda() for t in tg]) for g, tg in gts]
    got = {}
    for mode in ('plain', 'pipe'):
        model, _ = build_model(cfg, 0, 'cuda')
        ts = TrainStep(model, cfg)
        losses = []
        for i in range(3):
            nxt = xs[i + 1] if (mode == 'pipe' and i + 1 < 3) else None
            losses.append(ts.step(xs[i], gts[i][0], gts[i][1], 0.002, next_x=nxt).clone())
        torch.cuda.synchronize()
        ts.sync_to_model()
        sd = model.state_dict()
        stats = torch.cat([sd[k].double().reshape(-1).cpu() for k in sorted(sd) if 'running_' in k or 'num_batches' in k])
        got[mode] = (torch.stack(losses).cpu(), ts.pflat.clone().cpu(), stats)
    for a, b in zip(got['plain'], got['pipe']):
        assert torch.equal(a, b), 'the pipelined loop differs from the plain one at a GPU-bound size'


def test_frozen_stage2_layers_without_raw_tensors_change_nothing(monkeypatch):
    """Round 4 (train.py conv_unit, PPYOLO_HIP_TRAIN_BN_EPILOGUE): the frozen 1x1 layers that run on the streaming kernel apply their
    BatchNorm from the convolution's own epilogue instead of storing the raw output -- losses, gradients, parameters and the
    BatchNorm running statistics after two steps are EQUAL to the two-tensor form's (R50vd: the stage-2 conv3 / shortcut layers)."""
    from ppyolo_hip.train import TrainStep
    cfg = PPYOLO_2x_Config()
    N, S = 2, 256
    x = synth.synth_images(N, S, seed=11).cuda()
    gt, targets = synth_targets(cfg, N, S, 5)
    gt, targets = gt.cuda(), [t.cuda() for t in targets]
    got, used = {}, {}
    for mode in ('0', '1'):
        monkeypatch.setenv('PPYOLO_HIP_TRAIN_BN_EPILOGUE', mode)
        model, _ = build_model(cfg, 0, 'cuda')
        ts = TrainStep(model, cfg)
        calls = []
        from ppyolo_hip import ops as K
        # (the tables know the bench geometry; at this size the conv3 / shortcut layers of stage 2 are put on the streaming kernel by hand)
        ts._tuned_f['conv:N%d:H%d:W%d:C64:K256:R1:s1:f' % (N, S // 4, S // 4)] = [K.stream_first_cfg(), 1, 0.0]
        real = K.conv1x1_bn_apply
        monkeypatch.setattr(K, 'conv1x1_bn_apply', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        losses = [ts.step(x, gt, targets, 0.002).clone() for _ in range(2)]
        torch.cuda.synchronize()
        monkeypatch.setattr(K, 'conv1x1_bn_apply', real)
        used[mode] = len(calls)
        sd = model.state_dict()
        got[mode] = [torch.stack(losses).cpu(), ts.gflat.clone().cpu(), ts.pflat.clone().cpu()] + \
                    [sd[k].clone().cpu() for k in sorted(sd) if 'running_' in k and 'stage2' in k]
    assert used['0'] == 0 and used['1'] >= 2 * 3, used
    for a, b in zip(got['0'], got['1']):
        assert torch.equal(a, b)
    # the default also moves the frozen C = 128 layers onto the streaming kernel (another tile than the table's: same step to fp32 noise)
    monkeypatch.setenv('PPYOLO_HIP_TRAIN_BN_EPILOGUE', '2')
    model, _ = build_model(cfg, 0, 'cuda')
    ts = TrainStep(model, cfg)
    l2 = torch.stack([ts.step(x, gt, targets, 0.002).clone() for _ in range(2)]).cpu()
    # (training-mode BatchNorm amplifies fp32 rounding ~1000x on the way to the head -- in the reference as much as here, see
    # test_train_step_three_way -- and the second step starts from parameters the first one's rounding has touched)
    l0 = got['0'][0].sum(1)
    assert ts.bn_epilogue_all and abs(float(l2[0].sum() - l0[0])) < 2e-3 * float(l0[0]) and abs(float(l2[1].sum() - l0[1])) < 2e-2 * float(l0[1])


def test_sgd_groups_and_ema_semantics():
    """optimizer.step() + ema.update() of reference train.py:442-444: weight decay on convolution weights only
    (custom_layers.py:167-215), momentum buffers start as the first gradient, EMA decay warms up as (1+t)/(10+t)."""
    from ppyolo_hip.train import TrainStep
    cfg = PPYOLO_r18vd_Config()
    N, S, lr = 2, 128, 0.01
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, S, seed=3).cuda()
    gt, targets = synth_targets(cfg, N, S, 9)
    gt, targets = gt.cuda(), [t.cuda() for t in targets]
    ts = TrainStep(model, cfg)
    wd, mu = cfg.optimizerBuilder['regularizer']['factor'], cfg.optimizerBuilder['optimizer']['momentum']
    keys = ['head.yolo_output_convs.1.conv.weight', 'head.yolo_output_convs.1.conv.bias', 'head.detection_blocks.0.layers.2.bn.weight',
            'head.detection_blocks.1.tip_layers.1.conv.weight']
    p = {k: sd[k].clone() for k in keys}
    v = {}
    shadow = {k: sd[k].clone().numpy() for k in keys}
    for step in range(3):
        ts.forward_backward(x, gt, targets)
        g = {k: t.cpu() for k, t in ts.grads().items() if k in keys}
        ts.sgd(lr)
        ts.sync_to_model()
        now = {k: model.state_dict()[k].cpu() for k in keys}
        for k in keys:
            d = g[k] + (wd if k.endswith('conv.weight') else 0.0) * p[k]
            v[k] = d if step == 0 else mu * v[k] + d
            p[k] = p[k] - lr * v[k]
            assert relmax(now[k], p[k]) <= 2e-6, (step, k, relmax(now[k], p[k]))
            shadow[k], decay = trn.ema_update(shadow[k], now[k].numpy(), step, cfg.ema_decay if hasattr(cfg, 'ema_decay') else 0.9998)
            p[k] = now[k]                      # follow the device values: the comparison is per step
        assert abs(decay - (1 + step) / (10 + step)) < 1e-12
    ts.sync_to_model(ema=True)
    for k in keys:
        assert relmax(model.state_dict()[k].cpu(), torch.from_numpy(shadow[k])) <= 2e-6, k
    ts.sync_to_model()
    assert relmax(model.state_dict()[keys[0]].cpu(), p[keys[0]]) <= 1e-7


def test_stage5_blocks_backward_strict():
    """The three DCNv2 bottlenecks of R50vd's stage 5 (strided deformable 3x3 + avg-pool projection shortcut, then two identity
    blocks) as they train with freeze_at = 4: forward and every gradient -- conv / DCN weights, conv_offset weight and bias,
    BatchNorm scales and offsets, and the data gradient back into stage 4's output -- against torch autograd through the
    oracle, on a well-conditioned random input."""
    from ppyolo_hip.train import TrainStep, Act
    cfg = PPYOLO_2x_Config()
    cfg.backbone['freeze_at'] = 4
    N, Hs = 4, 16
    model, sd = build_model(cfg, 0, 'cuda')
    g = torch.Generator().manual_seed(12)
    x = torch.relu(torch.randn(N, 1024, Hs, Hs, generator=g))
    state = {k: v.clone() for k, v in sd.items()}
    keys = [k for k in trn.trainable_keys(sd, 4) if k.startswith('backbone.')]
    assert keys and all(k.startswith('backbone.stage5_') for k in keys)
    for k in keys:
        state[k].requires_grad_(True)
    xo = x.clone().requires_grad_()
    orc.TRAIN_MODE[0] = True
    try:
        torch.set_num_threads(16)
        y = xo
        for b in range(3):
            y = orc._bottleneck(state, 'backbone.stage5_%d' % b, y, 2 if b == 0 else 1, b == 0, False)
    finally:
        orc.TRAIN_MODE[0] = False
    dy = torch.randn(y.shape, generator=g) / y.numel() ** 0.5
    y.backward(dy)
    ts = TrainStep(model, cfg)
    ts.tape, ts._nbt = [], []
    xin = Act(x.permute(0, 2, 3, 1).contiguous().cuda(), 0, 1024, True)
    with torch.no_grad():
        out = xin
        for b in range(3):
            out = ts._bottleneck('backbone.stage5_%d' % b, out, 2 if b == 0 else 1, b == 0, False)
        ts._alloc_flat()
        e = relmax(out.dense_nchw(), y.detach())
        print('stage 5 output: max error %.2e of the maximum' % e)
        assert e <= 2e-4
        out.g = Act(dy.permute(0, 2, 3, 1).contiguous().cuda(), 0, out.C)
        for fn in reversed(ts.tape):
            fn()
    torch.cuda.synchronize()
    grads = ts.grads()
    worst = {k: relmax(grads[k], state[k].grad) for k in keys}
    worst['d input'] = relmax(xin.g.dense_nchw(), xo.grad)
    print('stage 5 backward, %d tensors: max relative gradient error %.2e (%s), median %.2e'
          % (len(worst), max(worst.values()), max(worst, key=worst.get), float(np.median(list(worst.values())))))
    cos = {k: float(torch.nn.functional.cosine_similarity(grads[k].double().cpu().reshape(1, -1), state[k].grad.double().reshape(1, -1)))
           for k in keys}
    cos['d input'] = float(torch.nn.functional.cosine_similarity(xin.g.dense_nchw().double().cpu().reshape(1, -1), xo.grad.double().reshape(1, -1)))
    print('   min cosine similarity %.6f (%s)' % (min(cos.values()), min(cos, key=cos.get)))
    # With a random (zero-mean) upstream gradient every weight-gradient entry is a 256-term random walk, and ONE ReLU of the 524 288
    # per layer that flips -- the activations of the two implementations agree to 4e-5 of the maximum, so a few dozen elements
    # within that distance of zero do -- moves the entries of its row by 1/sqrt(256) = 6 % of their typical size.  Hence: every
    # tensor points the same way to 1e-3, and no entry is further off than a flip explains.
    assert min(cos.values()) >= 0.999, {k: v for k, v in cos.items() if v < 0.999}
    assert max(worst.values()) <= 0.12 and float(np.median(list(worst.values()))) <= 1.5e-2, {k: v for k, v in worst.items() if v > 1.5e-2}


def _assert_three_way(res, capsys=None):
    """The whole step against the float64 oracle, held to the fp32 oracle's own distance from it (tests/train_parity_util.py).
    Why not a plain tolerance: the training-mode forward normalises every layer with BATCH statistics, which makes the random-
    weight network mildly chaotic -- the reference's own fp32 forward drifts 1.2x per layer from a float64 run (3.5x per DCNv2
    block), to 2.4e-3 rms at the head outputs and 12 % at the gradients, at 608 px as much as at 192 (profiles/r03_train_parity.txt).
    Two correct fp32 implementations are therefore this far apart; what can be held is that the HIP path is no further from
    the exact answer than the reference is."""
    import train_parity_util as tp
    lines = tp.summarize(res)
    if capsys is not None:
        with capsys.disabled():
            print('\n' + '\n'.join(lines))
    for d in res['feats']:
        assert d['hip_rms'] <= 1.5 * d['ref_rms'] + 1e-7 * d['rms'], ('backbone feature map', d)
    for d in res['heads']:
        assert d['hip_rms'] <= 1.5 * d['ref_rms'], ('head outputs', d)
    # a loss term is a sum over P ~ 10^2..10^3 positive cells of quantities that carry the head outputs' error delta ~ 3e-3:
    # relative error ~ delta / sqrt(P) ~ 1e-3 for either implementation (the reference's own: 3e-7 .. 4e-4, by luck of the draw)
    for k, (h, a, b) in res['losses'].items():
        assert abs(h - b) <= 5e-3 * abs(b), (k, h, a, b)
    gh = np.array([v[0] for v in res['grads'].values()])
    gr = np.array([v[1] for v in res['grads'].values()])
    # no tensor further from float64 than 2.5x the reference's WORST tensor (measured 0.98 .. 1.9x); each within 2.5x its own reference error + half that
    # worst (the per-tensor errors are draws of the same noise process -- a few flipped LeakyReLU / |.| signs -- not a bias: the
    # HIP / reference ratio of a given tensor ranges 0.3 .. 5 in both directions, profiles/r03_train_parity.txt)
    assert gh.max() <= 2.5 * gr.max() + 1e-5 and bool((gh <= 2.5 * gr + 0.5 * gr.max() + 1e-6).all()), \
        (np.median(gh), np.median(gr), gh.max(), gr.max())
    bh = np.array([v[0] for v in res['bn'].values()])
    br = np.array([v[1] for v in res['bn'].values()])
    assert np.median(bh) <= 1e-5 and bh.max() <= 1.5 * br.max() + 1e-6, (np.median(bh), bh.max(), br.max())
    ho = res.get('head_only')
    if ho:
        # identical inputs to the part that trains: fp32-level tolerances
        for k, (h, a, b) in ho['losses'].items():
            assert abs(h - b) <= 1e-5 * abs(b), (k, h, a, b)                   # (asked for: 1e-4; measured <= 7.2e-7)
        assert all(v[0] <= 1e-5 for v in ho['outs']), ho['outs']               # measured <= 2.6e-6 of the maximum
        assert all(v[0] <= 2e-5 for v in ho['douts']), ho['douts']             # measured <= 3.4e-6
        hh = np.array([v[0] for v in ho['grads'].values()])
        hr = np.array([v[1] for v in ho['grads'].values()])
        # (the R50vd head's backward carries ~1e-3 of fp32 noise in BOTH implementations: reference fp32 vs float64 median
        # 1.5e-4 .. 6e-4, max 1.3e-3 .. 2e-3; HIP median 2.4e-4 .. 7e-4, max 6e-4 .. 3e-3; r18vd: 1e-6 for both.  It is LeakyReLU
        # slopes: the level-0 gradient sits on a few dozen positive cells, so ONE element of the 3 M of a layer whose pre-activation is
        # within 2e-6 of zero and lands on the other side (slope 1 vs 0.1) moves that channel's gradients by ~10 % = 2e-3 of the
        # tensor -- and everything upstream of it.  tools/probes/train_tip_probe2.py: on the HIP path's OWN saved tensors its
        # BatchNorm backward is 4e-8, its weight gradient 4e-7 and its batch statistics 3e-8 from float64 evaluations of the same
        # formulas; train_tip_probe.py: d loss / d output of the level-0 tip 3e-6, its weight gradient 2e-3.)
        assert hh.max() <= 3.0 * hr.max() + 1e-5 and hh.max() <= 1e-2 and np.median(hh) <= 2.5 * np.median(hr) + 1e-6, \
            (np.median(hh), np.median(hr), hh.max(), hr.max())
        # Round 5, PER TENSOR: against the float64 oracle locked to each implementation's own LeakyReLU slope pattern (the lottery of
        # the free-running comparison above is gone: what is left is rounding), every tensor of the HIP step is within 3x the
        # reference's own fp32 error or within 1e-5 -- measured at R50vd-608: worst ratio 2.6x at 6e-6 (profiles/r05_train_parity.txt)
        lk = ho.get('locked') or {}
        assert lk, 'the slope-locked comparison did not run'
        bad = {k: v for k, v in lk.items() if not (v[0] <= 3.0 * v[1] or v[0] <= 1e-5)}
        assert not bad, ('gradient tensors further from their slope-locked float64 twin than 3x the reference is from its own', bad)


def test_train_step_full_size(capsys):
    """BASELINE config 5 at ITS OWN workload: R50vd 608x608, 8 images, freeze_at = 5, DropBlock active (the oracle's masks),
    the f16x2 kernels -- three-way against the fp32 and the float64 training oracle (reference train.py:416-443,
    model/losses.py:121-241, model/iou_losses.py:39-246); numbers in profiles/r03_train_parity.txt."""
    import train_parity_util as tp
    res = tp.three_way(PPYOLO_2x_Config, 608, 8, 5, True)
    assert len(res['grads']) == 69 and len(res['losses']) == 6 and 'head_only' in res
    _assert_three_way(res, capsys)


@pytest.mark.parametrize('cfgc,S,N', [(PPYOLO_2x_Config, 320, 4), (PPYOLO_r18vd_Config, 416, 8)])
def test_train_step_three_way(cfgc, S, N, capsys):
    """The whole step at a size the CPU oracle runs in seconds (golden g16's 320 px for R50vd; config 2's workload for r18vd)."""
    import train_parity_util as tp
    _assert_three_way(tp.three_way(cfgc, S, N, 5, True), capsys)


@pytest.mark.parametrize('cfgc,S,fa', [(PPYOLO_r18vd_Config, 256, 2), (PPYOLO_r18vd_Config, 192, 0), (PPYOLO_2x_Config, 320, 3)])
def test_train_step_with_backbone_stages(cfgc, S, fa, capsys):
    """freeze_at < 5: the stages above it train with the head (reference model/resnet_vd.py:174-200) -- strided 3x3 data gradients,
    avg-pool shortcuts, DCNv2 backward in the loop; three-way against the oracle's autograd in fp32 and float64, DropBlock off."""
    import train_parity_util as tp
    res = tp.three_way(cfgc, S, 4, fa, False, want_model=True)
    ts, model = res.pop('_ts'), res.pop('_model')
    bb = [k for k in res['grads'] if k.startswith('backbone.')]
    assert any(k.startswith('backbone.stage%d' % (fa + 1)) for k in bb)
    _assert_three_way(res, capsys)
    # and one SGD step runs over the enlarged parameter set (conv_offset biases in the weight-decay group)
    before = model.state_dict()[bb[0]].clone()
    ts.sgd(1e-3)
    ts.sync_to_model()
    assert not torch.equal(model.state_dict()[bb[0]], before)


def test_multi_scale_steps():
    """The reference trains at a different input size every few iterations (config/ppyolo_2x.py: random shapes 320..608): the same
    TrainStep takes batches of different sizes and batch counts back to back; loss and every gradient stay finite, the parameters
    move, and a size seen before gives the same loss for the same parameters (no state leaks between shapes)."""
    from ppyolo_hip.train import TrainStep
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    cfg = PPYOLO_2x_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    cfg.head['drop_active'] = False
    ts = TrainStep(model, cfg)
    hc = cfg.head

    def batch(N, S, seed):
        bb, cc, ss = synth_ground_truth(N, seed)
        tg = [torch.from_numpy(t).cuda() for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
        return synth.synth_images(N, S, seed=seed).cuda(), torch.from_numpy(bb).cuda(), tg
    first = None
    for N, S, seed in ((2, 320, 1), (3, 416, 2), (1, 352, 3), (2, 320, 1)):
        x, gt, tg = batch(N, S, seed)
        loss6 = ts.forward_backward(x, gt, tg)
        torch.cuda.synchronize()
        assert torch.isfinite(loss6).all() and float(loss6.sum()) > 0
        assert torch.isfinite(ts.gflat).all() and float(ts.gflat.abs().max()) > 0
        if first is None:
            first = loss6.clone()
    # the last batch repeats the first with unchanged parameters (no sgd() in between): identical loss terms
    assert torch.equal(loss6, first)
    ts.sgd(1e-3)
    x, gt, tg = batch(2, 320, 1)
    assert not torch.equal(ts.forward_backward(x, gt, tg), first)
