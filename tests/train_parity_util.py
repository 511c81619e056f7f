"""Three-way comparison of ONE training step (BASELINE config 5; reference train.py:416-443, model/losses.py:121-241,
model/iou_losses.py:39-246): the HIP step (ppyolo_hip/train.py), the CPU training oracle in float32 (= the reference's own
arithmetic, bit-equal to it on the build box: tests/test_train_oracle.py) and the same oracle in float64 (= the exact answer),
on the same images, targets and DropBlock masks.  Test infrastructure (it imports oracle/): used by
tests/test_gpu_train_step.py and tools/train_fullsize_parity.py only."""
import time

import numpy as np
import torch

from conftest import build_model
from oracle import ppyolo_oracle as orc, train_oracle as trn
from ppyolo_hip import synth

LOSS_NAMES = ['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware']


def _capture_feats(fn_name, store):
    real = getattr(orc, fn_name)

    def wrapped(state, x, fmaps=None):
        feats = real(state, x, fmaps) if fmaps is not None else real(state, x)
        store.append([f.detach() for f in feats])
        return feats
    return real, wrapped


def _oracle_step(sd, cfg, x, gt, targets, masks_in=None, seed=99, drop=True, feats_out=None):
    """One oracle step.  masks_in None: DropBlock draws its masks (recorded exactly: the draw is repeated on a tensor of ones
    with the generator state restored); else the given masks are applied.  -> (result dict, masks)."""
    real = orc.drop_block_train
    masks = []
    pending = list(masks_in) if masks_in is not None else None

    def draw(t, block_size=3, keep_prob=0.9):
        if not drop:
            return t
        if pending is not None:
            m = pending.pop(0).to(t.dtype)
            masks.append(m)
            return t * m * float(t.numel()) / m.sum()
        state = torch.get_rng_state()
        m = (real(torch.ones_like(t), block_size, keep_prob) != 0).to(t.dtype)
        torch.set_rng_state(state)
        masks.append(m)
        return real(t, block_size, keep_prob)
    orc.drop_block_train = draw
    bname = 'resnet50vd' if cfg.backbone_type == 'Resnet50Vd' else 'resnet18vd'
    breal, bwrap = _capture_feats(bname, feats_out if feats_out is not None else [])
    setattr(orc, bname, bwrap)
    try:
        r = trn.train_step(sd, cfg, x, gt, targets, rng_seed=seed)
    finally:
        orc.drop_block_train = real
        setattr(orc, bname, breal)
    return r, masks


def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def three_way(cfgc, S, N, freeze_at=5, drop=True, threads=16, image_seed=11, gt_seed=50, targets_fn=None, want_model=False):
    """-> dict: per loss term (hip, ref32, ref64), per head level rms errors vs float64, per gradient tensor relative L2 errors vs
    float64 (hip / ref32), BatchNorm running statistics (max relative error of the HIP path vs the float32 oracle)."""
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    from ppyolo_hip.train import TrainStep
    cfg = cfgc()
    cfg.backbone['freeze_at'] = freeze_at
    model, sd = build_model(cfg, 0, 'cuda')
    if not drop:
        cfg.head['drop_active'] = False
    x = synth.synth_images(N, S, seed=image_seed)
    hc = cfg.head
    if targets_fn is None:
        bb, cc, ss = synth_ground_truth(N, gt_seed)
        gt = torch.from_numpy(bb)
        targets = [torch.from_numpy(t) for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
    else:
        gt, targets = targets_fn(cfg, N, S)
    torch.set_num_threads(threads)
    t0 = time.time()
    f32, f64 = [], []
    r32, masks = _oracle_step(sd, cfg, x, gt, targets, None, drop=drop, feats_out=f32)
    t32 = time.time() - t0
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    t0 = time.time()
    r64, _ = _oracle_step(sd64, cfg, x.double(), gt.double(), [t.double() for t in targets], [m.clone() for m in masks], drop=drop, feats_out=f64)
    t64 = time.time() - t0
    ts = TrainStep(model, cfg)
    hip_feats = []
    real_backbone = ts.backbone

    def backbone_spy(xn):
        fs = real_backbone(xn)
        hip_feats.extend(f.dense_nchw().double().cpu() for f in fs)
        return fs
    ts.backbone = backbone_spy
    loss6 = ts.forward_backward(x.cuda(), gt.cuda(), [t.cuda() for t in targets], dropblock_masks=[m.float() for m in masks] if drop else None)
    torch.cuda.synchronize()
    ts.backbone = real_backbone
    out = dict(S=S, N=N, freeze_at=freeze_at, oracle_seconds=(round(t32, 1), round(t64, 1)), losses={}, heads=[], grads={}, bn={}, feats=[])
    for i, h in enumerate(hip_feats):
        w64 = f64[0][i]
        out['feats'].append(dict(hip_rms=float((h - w64).pow(2).mean().sqrt()), ref_rms=float((f32[0][i].double() - w64).pow(2).mean().sqrt()),
                                 rms=float(w64.pow(2).mean().sqrt())))
    for j, nme in enumerate(LOSS_NAMES):
        if nme in r64['losses']:
            out['losses'][nme] = (float(loss6[j]), float(r32['losses'][nme]), float(r64['losses'][nme]))
    for i, o in enumerate(ts.outs):
        h = o.dense_nchw().double().cpu()
        w64 = r64['outs'][i].detach()
        out['heads'].append(dict(hip_rms=float((h - w64).pow(2).mean().sqrt()), ref_rms=float((r32['outs'][i].detach().double() - w64).pow(2).mean().sqrt()),
                                 hip_vs_ref32_max=float((h - r32['outs'][i].detach().double()).abs().max()), scale=float(w64.abs().max())))
    grads = ts.grads()
    assert set(grads) == set(r64['grads']), 'the sets of trained tensors differ'
    for k, g in grads.items():
        out['grads'][k] = (rel_l2(g, r64['grads'][k]), rel_l2(r32['grads'][k], r64['grads'][k]))
    msd = model.state_dict()
    for k in sd:
        if k.endswith(('running_mean', 'running_var')):
            a, b, c = msd[k].double().cpu(), r32['state'][k].double(), r64['state'][k]
            den = c.abs().max().clamp_min(1e-30)
            out['bn'][k] = (float((a - c).abs().max() / den), float((b - c).abs().max() / den))
    # ---- part B: the part of the network that TRAINS (freeze_at = 5: the head), on identical inputs -- the float64 oracle's
    # backbone features, rounded to float32, through the fp32 oracle's head (autograd) and through the HIP head + loss +
    # backward, same DropBlock masks: no 53-layer chaotic amplification in front, so fp32-level tolerances apply
    if freeze_at == 5:
        out['head_only'] = head_only(cfg, sd, model, [f.float() for f in f64[0]], gt, targets, masks if drop else None)
    if want_model:
        out['_ts'], out['_model'] = ts, model
    return out


def head_only(cfg, sd, model, feats, gt, targets, masks):
    from ppyolo_hip import ops as K
    from ppyolo_hip.train import TrainStep, Act
    real = orc.drop_block_train
    pending = [m.clone() for m in masks] if masks is not None else None

    def draw(t, block_size=3, keep_prob=0.9):
        if pending is None:
            return t
        m = pending.pop(0).to(t.dtype)
        return t * m * float(t.numel()) / m.sum()
    res = {}
    # LeakyReLU slope patterns (round 5).  A gradient is a LINEAR function of d loss / d output once every LeakyReLU's slope pattern is
    # fixed, and one element of a layer's 3 M whose pre-activation lies within the forward's 1e-6 of zero can land on either side
    # (slope 1 vs 0.1) in any fp32 evaluation -- which moves every gradient upstream of it by up to 1e-3 and makes the per-tensor
    # HIP / reference ratio a lottery (30x in either direction).  So besides the free-running oracles, the float64 oracle is run
    # LOCKED to the slope pattern of the implementation it judges: `lock64hip` with the HIP forward's own signs, `lock64ref` with the
    # fp32 oracle's.  Against its own locked float64 twin an implementation shows its rounding error only.
    real_cu = orc.conv_unit
    recorded = {}

    def make_cu(record, lock):
        def cu(sd_, prefix, x, stride=1, act=None):
            if act == 'leaky' and lock is not None and prefix in lock:
                y = real_cu(sd_, prefix, x, stride, None)
                return y * (0.1 + 0.9 * lock[prefix].to(y.dtype))
            y = real_cu(sd_, prefix, x, stride, act)
            if act == 'leaky' and record is not None:
                record[prefix] = (y.detach() > 0)
            return y
        return cu
    ts = TrainStep(model, cfg)
    hip_signs = {}
    real_hip_cu = ts.conv_unit

    def hip_cu(prefix, x, stride=1, act=None, **kw):
        y = real_hip_cu(prefix, x, stride, act, **kw)
        if act == 'leaky':
            hip_signs[prefix] = (y.dense_nchw() > 0).cpu()
        return y
    ts.conv_unit = hip_cu
    ts.tape, ts._nbt = [], []
    ts.masks = [m.float() for m in masks] if masks is not None else None
    if masks is None:
        cfg.head['drop_active'] = False
    fa = []
    for f in feats:
        t = f.permute(0, 2, 3, 1).contiguous().cuda()
        fa.append(Act(t, 0, t.shape[3], False, K.amax_slots(t) if ts.f16 else None))      # (tracked maxima: the f16x2 kernels, as in the step)
    loss6 = ts.head_loss_backward(fa, gt.cuda(), [t.cuda() for t in targets])
    torch.cuda.synchronize()
    ts.conv_unit = real_hip_cu
    for tag, dt in (('ref32', torch.float32), ('ref64', torch.float64), ('lock64hip', torch.float64), ('lock64ref', torch.float64)):
        state = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in trn.trainable_keys(sd):
            state[k].requires_grad_(True)
        pending = [m.clone() for m in masks] if masks is not None else None
        orc.drop_block_train = draw
        orc.TRAIN_MODE[0] = True
        recorded[tag] = {}
        orc.conv_unit = make_cu(recorded[tag], {'lock64hip': hip_signs, 'lock64ref': recorded.get('ref32')}.get(tag))
        try:
            outs = orc.head_outputs(state, [f.to(dt) for f in feats], cfg.head)
        finally:
            orc.TRAIN_MODE[0] = False
            orc.drop_block_train = real
            orc.conv_unit = real_cu
        for o in outs:
            o.retain_grad()
        losses = trn.yolov3_loss(outs, [t.to(dt) for t in targets], gt.to(dt), cfg)
        sum(losses.values()).backward()
        res[tag] = dict(losses={k: float(v.detach()) for k, v in losses.items()}, outs=[o.detach() for o in outs], douts=[o.grad for o in outs],
                        grads={k: state[k].grad for k in trn.trainable_keys(sd)})
    out = dict(losses={}, outs=[], douts=[], grads={}, locked={}, flips={})
    for j, nme in enumerate(LOSS_NAMES):
        if nme in res['ref64']['losses']:
            out['losses'][nme] = (float(loss6[j]), res['ref32']['losses'][nme], res['ref64']['losses'][nme])
    for i, o in enumerate(ts.outs):
        w = res['ref64']['outs'][i]
        out['outs'].append((float((o.dense_nchw().double().cpu() - w).abs().max() / w.abs().max()),
                            float((res['ref32']['outs'][i].double() - w).abs().max() / w.abs().max())))
        wd = res['ref64']['douts'][i]
        out['douts'].append((rel_l2(o.g.dense_nchw(), wd), rel_l2(res['ref32']['douts'][i], wd)))
    grads = ts.grads()
    for k, g in grads.items():
        out['grads'][k] = (rel_l2(g, res['ref64']['grads'][k]), rel_l2(res['ref32']['grads'][k], res['ref64']['grads'][k]))
        out['locked'][k] = (rel_l2(g, res['lock64hip']['grads'][k]), rel_l2(res['ref32']['grads'][k], res['lock64ref']['grads'][k]))
    # LeakyReLU elements on the other side of zero than in the free-running float64 oracle: (HIP, fp32 oracle, elements)
    for prefix, m64 in recorded['ref64'].items():
        nh = int((hip_signs[prefix] != m64).sum()) if prefix in hip_signs else -1
        nr = int((recorded['ref32'][prefix] != m64).sum())
        if nh or nr:
            out['flips'][prefix] = (nh, nr, m64.numel())
    return out


def summarize(res):
    """Human-readable lines for profiles/ and -s output."""
    L = ['%s S=%d N=%d freeze_at=%d (oracle fp32 %.0f s, float64 %.0f s on the host)' % ('train step', res['S'], res['N'], res['freeze_at'],
                                                                                       res['oracle_seconds'][0], res['oracle_seconds'][1])]
    for k, (h, a, b) in res['losses'].items():
        L.append('  %-15s hip %.6f  ref32 %.6f  float64 %.6f   |hip-f64|/f64 %.2e   |ref32-f64|/f64 %.2e' % (k, h, a, b, abs(h - b) / abs(b), abs(a - b) / abs(b)))
    for i, d in enumerate(res['heads']):
        L.append('  head level %d: rms error vs float64: HIP %.3e, reference fp32 %.3e (max|out| %.2f); max |hip-ref32| %.3e'
                 % (i, d['hip_rms'], d['ref_rms'], d['scale'], d['hip_vs_ref32_max']))
    gh = np.array([v[0] for v in res['grads'].values()])
    gr = np.array([v[1] for v in res['grads'].values()])
    worst = max(res['grads'], key=lambda k: res['grads'][k][0] / max(res['grads'][k][1], 1e-12))
    L.append('  gradients, %d tensors, relative L2 error vs float64: HIP median %.2e max %.2e | reference fp32 median %.2e max %.2e | '
             'worst HIP/ref ratio %.2f (%s: %.2e vs %.2e)' % (len(gh), np.median(gh), gh.max(), np.median(gr), gr.max(),
                                                             res['grads'][worst][0] / max(res['grads'][worst][1], 1e-12), worst,
                                                             res['grads'][worst][0], res['grads'][worst][1]))
    L += _worst_five(res['grads'], '  ')
    for i, d in enumerate(res['feats']):
        L.append('  backbone feature map %d (training-mode forward): rms error vs float64: HIP %.3e, reference fp32 %.3e (rms %.2f)'
                 % (i, d['hip_rms'], d['ref_rms'], d['rms']))
    bh = np.array([v[0] for v in res['bn'].values()])
    br = np.array([v[1] for v in res['bn'].values()])
    L.append('  BatchNorm running statistics, %d tensors, max error / max|.| vs float64: HIP median %.2e max %.2e | reference fp32 median %.2e max %.2e'
             % (len(bh), np.median(bh), bh.max(), np.median(br), br.max()))
    ho = res.get('head_only')
    if ho:
        L.append('  -- head + loss + backward on IDENTICAL backbone features (the float64 oracle\'s, rounded to fp32):')
        for k, (h, a, b) in ho['losses'].items():
            L.append('     %-15s hip %.6f  ref32 %.6f  float64 %.6f   |hip-f64|/f64 %.2e   |ref32-f64|/f64 %.2e' % (k, h, a, b, abs(h - b) / abs(b), abs(a - b) / abs(b)))
        L.append('     head outputs, max error / max|out| vs float64 per level: HIP %s | reference fp32 %s'
                 % (' '.join('%.2e' % v[0] for v in ho['outs']), ' '.join('%.2e' % v[1] for v in ho['outs'])))
        L.append('     d loss / d outputs, relative L2 error vs float64 per level: HIP %s | reference fp32 %s'
                 % (' '.join('%.2e' % v[0] for v in ho['douts']), ' '.join('%.2e' % v[1] for v in ho['douts'])))
        gh = np.array([v[0] for v in ho['grads'].values()])
        gr = np.array([v[1] for v in ho['grads'].values()])
        wk = max(ho['grads'], key=lambda k: ho['grads'][k][0])
        L.append('     gradients, %d tensors, relative L2 error vs float64: HIP median %.2e max %.2e (%s) | reference fp32 median %.2e max %.2e'
                 % (len(gh), np.median(gh), gh.max(), wk, np.median(gr), gr.max()))
        L += _worst_five(ho['grads'], '     ')
        if ho.get('locked'):
            lh = np.array([v[0] for v in ho['locked'].values()])
            lr = np.array([v[1] for v in ho['locked'].values()])
            L.append('     ... against the float64 oracle LOCKED to each implementation\'s own LeakyReLU slope pattern (rounding error only): HIP median %.2e max %.2e | '
                     'reference fp32 median %.2e max %.2e' % (np.median(lh), lh.max(), np.median(lr), lr.max()))
            L += _worst_five(ho['locked'], '     locked: ')
            fl = ho.get('flips', {})
            L.append('     LeakyReLU elements on the other side of zero than in the free float64 run: HIP %d, reference fp32 %d (in %d of the layers; e.g. %s)'
                     % (sum(max(v[0], 0) for v in fl.values()), sum(v[1] for v in fl.values()), len(fl),
                        ', '.join('%s %d/%d of %d' % (k, v[0], v[1], v[2]) for k, v in list(fl.items())[:4])))
    return L


def _worst_five(grads, ind):
    """The five tensors with the largest HIP / reference error ratio (each error = relative L2 distance from float64)."""
    rk = sorted(grads, key=lambda k: -grads[k][0] / max(grads[k][1], 1e-12))[:5]
    return ['%sworst HIP/ref ratios: %s' % (ind, '; '.join('%s %.1fx (%.2e vs %.2e)' % (k, grads[k][0] / max(grads[k][1], 1e-12), grads[k][0], grads[k][1])
                                                           for k in rk))]
