"""The reference's training LOOP run unchanged on this package (train.py:264-286, :416-444): backbone.freeze(),
model.add_param_group -> torch.optim.SGD, ExponentialMovingAverage, then `losses = model(images, None, False, gt_bbox, gt_class,
gt_score, targets); sum(losses).backward(); optimizer.step(); ema.update()` three times -- against golden g14, the same loop
on the reference itself (tools/make_goldens.py g14)."""
import numpy as np
import pytest
import torch

from conftest import build_train_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from ppyolo_hip import synth

pytestmark = pytest.mark.gpu


def digest(t):
    d = t.detach().double().reshape(-1).cpu()
    step = max(1, d.numel() // 64)
    return np.array([d.sum().item(), d.abs().sum().item(), d.pow(2).sum().sqrt().item()]), d[::step][:64].float().numpy()


def test_reference_training_loop_runs_unchanged(golden):
    from model.EMA import ExponentialMovingAverage
    g = golden('g14_train_loop')
    S, N = int(g['meta'][0]), int(g['meta'][1])
    base_lr, wd, mom, decay = [float(v) for v in g['hyper']]
    cfg = PPYOLO_r18vd_Config()
    m = build_train_model(cfg, 0, 'cuda')
    m.head.set_dropblock(is_test=True)
    init = {k: v.detach().clone() for k, v in m.named_parameters()}
    groups = []
    m.add_param_group(groups, base_lr, wd)
    opt = torch.optim.SGD(groups, lr=base_lr, momentum=mom, weight_decay=wd)
    ema = ExponentialMovingAverage(m, decay)
    ema.register()
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    targets = [T(g['target%d' % i]) for i in range(2)]
    names = [str(v) for v in g['loss_names']]
    worst_loss = 0.0
    for it in range(3):
        x = synth.synth_images(N, S, seed=1234 + it).cuda()
        losses = m(x, None, False, T(g['gt_bbox']), T(g['gt_class']), T(g['gt_score']), targets)
        assert list(losses.keys()) == names
        all_loss = 0.0
        for k in losses:
            all_loss = all_loss + losses[k]
        lr = base_lr * (1.0 - 0.25 * it)
        for gr in opt.param_groups:
            gr['lr'] = lr * gr['base_lr'] / base_lr
        opt.zero_grad()
        all_loss.backward()
        opt.step()
        ema.update()
        got = np.array([float(losses[k].detach()) for k in names], np.float32)
        want = g['losses%d' % it]
        err = np.abs(got - want).max() / np.abs(want).max()
        worst_loss = max(worst_loss, float(err))
        # iteration 0 sees identical parameters; later ones the parameters the previous (fp32-noisy) updates produced
        assert err <= (1e-4 if it == 0 else 5e-4), (it, got, want)          # measured 6e-6
    # the parameters after three updates: error relative to how far the loop moved the tensor
    sd = dict(m.named_parameters())
    errs = {}
    for i, k in enumerate(str(v) for v in g['param_names']):
        _, smp = digest(sd[k])
        ref = g['param_samples'][i][:len(smp)]
        moved = float(g['update_l2'][i]) / max(1, sd[k].numel()) ** 0.5          # rms update of an element
        errs[k] = float(np.abs(smp - ref).max() / max(moved, 1e-12))
        _, ssm = digest(ema._shadow[k])
        sref = g['shadow_samples'][i][:len(ssm)]
        # shadow = init + (1 - decay_t)-weighted steps: same relative bar on its own (smaller) movement
        smoved = float(np.abs(sref - digest(init[k])[1]).max())
        assert np.abs(ssm - sref).max() <= 0.05 * smoved + 1e-7, (k, 'shadow')
    v = np.array(list(errs.values()))
    print('loss terms: max relative error over 3 iterations %.2e; parameters after 3 SGD steps: max error / rms update: median %.2e, '
          'max %.2e (%s)' % (worst_loss, np.median(v), v.max(), max(errs, key=errs.get)))
    assert np.median(v) <= 1e-3 and v.max() <= 2e-2                          # measured 4e-5 / 8e-5
    assert np.allclose(sd['head.yolo_output_convs.1.conv.bias'].detach().cpu().numpy(), g['after.head.yolo_output_convs.1.conv.bias'],
                       rtol=0, atol=2e-5)
    # apply / restore keep the parameters' storage (optimizers and the HIP step hold references to it)
    ptr = {k: q.data_ptr() for k, q in m.named_parameters()}
    ema.apply()
    assert torch.equal(sd['head.yolo_output_convs.0.conv.bias'], ema._shadow['head.yolo_output_convs.0.conv.bias'])
    ema.restore()
    assert all(q.data_ptr() == ptr[k] for k, q in m.named_parameters())


def test_r50_loss_terms_and_non_uniform_backward_is_refused():
    """All six terms on the R50vd head (IoU-aware branch); the bridge differentiates the SUM of the terms only."""
    from ppyolo_hip._lib import PPYoloHipError
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    cfg = PPYOLO_2x_Config()
    m = build_train_model(cfg, 0, 'cuda')
    S, N = 128, 2
    bb, cc, ss = synth_ground_truth(N, 3)
    hc = cfg.head
    targets = [torch.from_numpy(t).cuda() for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
    x = synth.synth_images(N, S, seed=5).cuda()
    losses = m(x, None, False, torch.from_numpy(bb).cuda(), torch.from_numpy(cc).cuda(), torch.from_numpy(ss).cuda(), targets)
    assert list(losses) == ['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware']
    assert all(torch.isfinite(v) and v.requires_grad for v in losses.values())
    with pytest.raises(PPYoloHipError):
        (losses['loss_xy'] * 2.0 + losses['loss_wh']).backward()
    losses = m(x, None, False, torch.from_numpy(bb).cuda(), None, None, targets)
    sum(losses.values()).backward()
    gw = m.head.yolo_output_convs[0].conv.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().max()) > 0
    assert m.backbone.stage1_conv1_1.conv.weight.grad is None
    # a model whose backbone still trains is refused (only the head's backward exists)
    m2 = build_train_model(cfg, 0, 'cuda')
    for q in m2.backbone.parameters():
        q.requires_grad = True
    with pytest.raises(PPYoloHipError):
        m2(x, None, False, torch.from_numpy(bb).cuda(), None, None, targets)


@pytest.mark.parametrize('tag,cfgc', [('r18vd_96_fa2', PPYOLO_r18vd_Config), ('r18vd_96_fa0', PPYOLO_r18vd_Config), ('r50vd_128_fa3', PPYOLO_2x_Config), ('r50vd_320_fa3', PPYOLO_2x_Config)])
def test_backbone_stages_train_through_the_reference_surface(golden, tag, cfgc):
    """freeze_at < 5 through the reference's calls: backbone.freeze() leaves the stages above freeze_at trainable, forward(eval=False)
    + backward() fills their .grad too (DCNv2, strided 3x3, avg-pool shortcuts) -- against golden g16, the reference's own
    loss terms and gradient norms for the same inputs."""
    g = golden('g16_train_' + tag)
    S, N, wseed, iseed, fa = [int(v) for v in g['meta']]
    cfg = cfgc()
    cfg.backbone['freeze_at'] = fa
    m = build_train_model(cfg, wseed, 'cuda')
    m.head.set_dropblock(is_test=True)
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    x = synth.synth_images(N, S, seed=iseed).cuda()
    targets = [T(g['target%d' % i]) for i in range(len(cfg.head['anchor_masks']))]
    losses = m(x, None, False, T(g['gt_bbox']), T(g['gt_class']), T(g['gt_score']), targets)
    names = [str(v) for v in g['loss_names']]
    got = np.array([float(losses[k].detach()) for k in names], np.float32)
    err = float(np.abs(got - g['loss_values']).max() / np.abs(g['loss_values']).max())
    sum(losses.values()).backward()
    sd = dict(m.named_parameters())
    gn = [str(v) for v in g['grad_names']]
    assert [k for k, q in sd.items() if q.grad is not None] == gn
    ratio = np.array([float(sd[k].grad.double().norm()) / max(float(g['grad_digest'][i][2]), 1e-30) for i, k in enumerate(gn)])
    print('%s: loss terms within %.1e; gradient norms / reference: median %.4f, range %.3f .. %.3f over %d tensors (%d backbone)'
          % (tag, err, np.median(ratio), ratio.min(), ratio.max(), len(gn), sum(k.startswith('backbone.') for k in gn)))
    assert err <= 2e-3
    if tag.startswith('r18'):
        assert np.abs(ratio - 1.0).max() <= 2e-3          # measured: 1.000 for all 64 tensors
    else:
        # R50vd at sizes the CPU reference can run: stage 5 is 4x4 / 10x10 -- BatchNorm over 32..200 samples per channel amplifies
        # fp32 rounding to 3e-3 of the head outputs (the reference's own fp32 gradient norms are 0.90..1.03 of a float64 run), and
        # at 320 px ONE level-0 cell then sits on a kink of the IoU loss: the predicted box edge crosses the ground-truth edge
        # between the two forwards (t_w = 0.128 vs 0.121), d loss / d t_w jumps from -1.49 to +0.03, and everything upstream of
        # level 0 carries 0.92..0.95 of the reference's norm (tools/probes/chain_probe.py; levels 1 and 2: 0.995..1.004).  The
        # loss kernel on the reference's own head outputs is exact (7e-7, tools/probes/loss_probe.py), r18vd matches tensor by
        # tensor, and the R50 blocks are held to fp32 tolerance on well-conditioned inputs in test_stage5_blocks_backward_strict.
        assert abs(np.median(ratio) - 1.0) <= 0.15 and ratio.min() >= 0.6 and ratio.max() <= 1.6


def test_ema_class_matches_reference(golden):
    """model/EMA.py on the device against golden g13 (the reference's own class, numpy float32 arithmetic)."""
    from model.EMA import ExponentialMovingAverage
    g = golden('g13_ema')
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.from_numpy(g['a0']).cuda())
    m.b = torch.nn.Parameter(torch.from_numpy(g['b0']).cuda())
    m.frozen = torch.nn.Parameter(torch.zeros(3).cuda(), requires_grad=False)
    ema = ExponentialMovingAverage(m, 0.9998)
    ema.register()
    for t in range(4):
        with torch.no_grad():
            m.a.copy_(torch.from_numpy(g['a_param%d' % t]))
            m.b.copy_(torch.from_numpy(g['b_param%d' % t]))
        d = ema.update()
        assert d == g['decays'][t] and 'frozen' not in ema._shadow
        assert np.array_equal(ema._shadow['a'].cpu().numpy(), g['a_shadow%d' % t])
        assert np.array_equal(ema._shadow['b'].cpu().numpy(), g['b_shadow%d' % t])


def _r18_loop_pieces(S=128, N=2, seed=7):
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    cfg = PPYOLO_r18vd_Config()
    m = build_train_model(cfg, 0, 'cuda')
    m.head.set_dropblock(is_test=True)          # (no random masks: the runs below are compared with each other)
    bb, cc, ss = synth_ground_truth(N, seed)
    hc = cfg.head
    targets = [torch.from_numpy(t).cuda() for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
    return cfg, m, torch.from_numpy(bb).cuda(), targets


def test_eval_between_training_iterations_sees_the_current_weights():
    """The reference's loop (train.py:481-499): every eval_iter iterations `ema.apply(); model.eval(); eval; model.train();
    ema.restore()`.  The inference plans hold COPIES of the folded weights; each evaluation must run the parameters of its own
    iteration -- checked against a freshly built model that loads the same state_dict."""
    from model.EMA import ExponentialMovingAverage
    from conftest import build_model
    S, N = 128, 2
    cfg, m, gt, targets = _r18_loop_pieces(S, N)
    groups = []
    m.add_param_group(groups, 0.002, 0.0005)
    opt = torch.optim.SGD(groups, lr=0.002, momentum=0.9, weight_decay=0.0005)
    ema = ExponentialMovingAverage(m, 0.9)
    ema.register()
    xe = synth.synth_images(N, S, seed=99).cuda()
    ims = synth.synth_im_size(N).cuda()

    def head_outputs(mod):
        ex = mod._plans.executor(xe)
        return [ex.view(a).dense().clone() for a in ex.plan.head_outs]

    def evaluate(with_ema):
        if with_ema:
            ema.apply()
        m.eval()
        preds = [p.clone() for p in m(xe, ims)] + head_outputs(m)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.train()
        if with_ema:
            ema.restore()
        fresh, _ = build_model(cfg, 0, 'cuda')
        fresh.load_state_dict(sd)
        want = [p for p in fresh(xe, ims)] + head_outputs(fresh)
        assert len(preds) == len(want)
        for a, b in zip(preds, want):
            assert torch.equal(a, b), 'the evaluation ran stale weights'
        return preds[N:]          # (the raw head outputs: detections may be empty on both sides)

    seen = []
    for it in range(4):
        x = synth.synth_images(N, S, seed=300 + it).cuda()
        losses = m(x, None, False, gt, None, None, targets)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        ema.update()
        if it in (1, 3):
            seen.append(evaluate(with_ema=True))
            seen.append(evaluate(with_ema=False))          # restore() changed the parameters again
    # the four evaluations did run four different sets of weights
    flat = [torch.cat([p.reshape(-1) for p in ps]) for ps in seen]
    for i in range(len(flat)):
        for j in range(i + 1, len(flat)):
            assert flat[i].shape != flat[j].shape or not torch.equal(flat[i], flat[j])
    # InFlight lanes notice as well
    pipe = m.eval().in_flight(2)
    t = pipe.submit(xe, ims)
    first = t.result()
    with torch.no_grad():
        m.head.yolo_output_convs[0].conv.bias.add_(0.5)
    def lane_outputs(k):
        ex = pipe.lanes(xe)[k][0]
        return [ex.view(a).dense().clone() for a in ex.plan.head_outs]
    first = first + lane_outputs(0)
    with torch.no_grad():
        m.head.yolo_output_convs[0].conv.bias.add_(0.5)
    second = pipe.submit(xe, ims).result()
    second = second + lane_outputs(1)           # (round robin: the second submit went to lane 1, rebuilt from the new weights)
    fresh, _ = build_model(cfg, 0, 'cuda')
    fresh.load_state_dict(m.state_dict())
    for a, b in zip(second, [p for p in fresh(xe, ims)] + head_outputs(fresh)):
        assert torch.equal(a, b)
    assert any(a.shape != b.shape or not torch.equal(a, b) for a, b in zip(first, second))
    # pinned: the check is off, the caller vouches for the weights
    m(xe, ims)                                  # (builds forward's own executor from the current weights)
    m.pin_weights()
    with torch.no_grad():
        m.head.yolo_output_convs[0].conv.bias.add_(0.5)
    m(xe, ims)
    for a, b in zip(head_outputs(m), second[N:]):
        assert torch.equal(a, b)


def test_two_training_forwards_before_one_backward_keep_their_own_gradients():
    """Gradient accumulation through the reference's surface: l1 = model(a); l2 = model(b); (l1 + l2).backward() must deliver
    grad(a) + grad(b), not twice the last forward's gradients (the step's flat gradient buffer is reused by every forward)."""
    S, N = 128, 2
    cfg, m, gt, targets = _r18_loop_pieces(S, N)
    xa, xb = synth.synth_images(N, S, seed=11).cuda(), synth.synth_images(N, S, seed=12).cuda()
    names = [k for k, q in m.named_parameters() if q.requires_grad]
    # BatchNorm running statistics do not enter a training forward's output, so the three passes see the same function
    single = []
    for x in (xa, xb):
        m.zero_grad()
        sum(m(x, None, False, gt, None, None, targets).values()).backward()
        single.append({k: q.grad.detach().clone() for k, q in m.named_parameters() if q.grad is not None})
    assert sorted(single[0]) == sorted(names)
    m.zero_grad()
    la = sum(m(xa, None, False, gt, None, None, targets).values())
    lb = sum(m(xb, None, False, gt, None, None, targets).values())
    (la + lb).backward()
    worst = 0.0
    for k, q in m.named_parameters():
        if q.grad is None:
            continue
        want = single[0][k] + single[1][k]
        assert torch.equal(q.grad, want), k
        worst = max(worst, float((single[0][k] - single[1][k]).abs().max()))
    assert worst > 0          # the two batches do have different gradients


def test_validation_forward_without_optimizer_step_invalidates_the_plans():
    """A training-mode forward that is NOT followed by optimizer.step() -- a validation-loss forward, the first micro-batches of a
    gradient accumulation -- still moves the BatchNorm running statistics, written by the HIP kernels through raw pointers (no
    autograd version bump), and may re-bind the num_batches_tracked buffers.  The inference plans folded from the old statistics
    must be dropped (PlanCache.mark_dirty, round 4): the next eval equals a freshly built model on the same state_dict."""
    from conftest import build_model
    S, N = 128, 2
    cfg, m, gt, targets = _r18_loop_pieces(S, N)
    xe = synth.synth_images(N, S, seed=99).cuda()
    ims = synth.synth_im_size(N).cuda()

    def head_outputs(mod):
        ex = mod._plans.executor(xe)
        return [ex.view(a).dense().clone() for a in ex.plan.head_outs]
    m.eval()
    before = [p.clone() for p in m(xe, ims)] + head_outputs(m)          # builds (and caches) the inference executor
    m.train()
    for it in range(2):                                                   # loss forwards only: no backward(), no step()
        m(synth.synth_images(N, S, seed=500 + it).cuda(), None, False, gt, None, None, targets)
    m.eval()
    after = [p.clone() for p in m(xe, ims)] + head_outputs(m)
    fresh, _ = build_model(cfg, 0, 'cuda')
    fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    want = [p for p in fresh(xe, ims)] + head_outputs(fresh)
    for a, b in zip(after, want):
        assert torch.equal(a, b), 'the evaluation ran plans folded from stale BatchNorm statistics'
    assert any(not torch.equal(a, b) for a, b in zip(before[N:], after[N:])), 'the statistics did move'
