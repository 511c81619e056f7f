"""Training step (BASELINE config 5): the CPU oracle of one forward + backward
(oracle/train_oracle.py) against what the REFERENCE computed for the same inputs (tests/golden/g12_train_*.npz, made from
/root/reference by tools/make_goldens.py g12): loss terms, training-mode head outputs, d loss / d head outputs, digests of
every parameter gradient, updated BatchNorm statistics."""
import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from oracle import train_oracle as trn
from ppyolo_hip import synth


def T(a):
    return torch.from_numpy(np.asarray(a))


def _step(golden, tag, cfgc):
    g = golden('g12_train_' + tag)
    S, N, wseed, iseed, rseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    _, sd = build_model(cfg, wseed, 'cpu')
    x = synth.synth_images(N, S, seed=iseed)
    L = len(cfg.head['anchor_masks'])
    targets = [T(g['target%d' % i]) for i in range(L)]
    torch.set_num_threads(8)
    r = trn.train_step(sd, cfg, x, T(g['gt_bbox']), targets, rng_seed=rseed)
    return g, cfg, sd, r


@pytest.mark.parametrize('tag,cfgc', [('r18vd_96', PPYOLO_r18vd_Config), ('r50vd_96', PPYOLO_2x_Config)])
def test_train_step_matches_reference(golden, tag, cfgc):
    g, cfg, sd, r = _step(golden, tag, cfgc)
    names = [str(n) for n in g['loss_names']]
    assert names == list(r['losses'].keys())
    got = np.array([float(r['losses'][k]) for k in names], np.float32)
    assert np.array_equal(got, g['loss_values']), (got, g['loss_values'])
    assert float(r['all_loss']) == float(g['all_loss'])
    for i, (o, d) in enumerate(zip(r['outs'], r['douts'])):
        assert torch.equal(o.detach(), T(g['out%d' % i])), 'training-mode head output %d' % i
        assert torch.equal(d, T(g['dout%d' % i])), 'd loss / d head output %d' % i
    gnames = [str(n) for n in g['grad_names']]
    assert gnames == [k for k in r['grads']] and set(gnames) == set(trn.trainable_keys(sd))
    for k, dig, smp in zip(gnames, g['grad_digest'], g['grad_samples']):
        d = r['grads'][k].double().reshape(-1)
        mine = np.array([d.sum().item(), d.abs().sum().item(), d.pow(2).sum().sqrt().item()])
        assert np.allclose(mine, dig, rtol=1e-12, atol=0), k
        step = max(1, d.numel() // 64)
        assert np.array_equal(d[::step][:64].float().numpy(), smp[:len(d[::step][:64])]), k
    for i in range(len(r['outs'])):
        assert torch.equal(r['grads']['head.yolo_output_convs.%d.conv.weight' % i], T(g['gw_out%d' % i]))
        assert torch.equal(r['grads']['head.yolo_output_convs.%d.conv.bias' % i], T(g['gb_out%d' % i]))
    # BatchNorm buffers moved (momentum 0.1, batch statistics), frozen or not
    for k in ('backbone.stage1_conv1_1.bn.running_mean', 'backbone.stage1_conv1_1.bn.running_var'):
        assert torch.equal(r['state'][k], T(g['after.' + k])) and not torch.equal(r['state'][k], sd[k])
    assert torch.equal(r['state'][str(g['after_name'])], T(g['after_value']))
    assert all(int(v) == 1 for k, v in r['state'].items() if k.endswith('num_batches_tracked'))


@pytest.mark.parametrize('tag,cfgc', [('r18vd_96_fa2', PPYOLO_r18vd_Config), ('r18vd_96_fa0', PPYOLO_r18vd_Config), ('r50vd_128_fa3', PPYOLO_2x_Config)])
def test_backbone_stages_training_matches_reference(golden, tag, cfgc, monkeypatch):
    """freeze_at < 5 (golden g16: the reference with stages above freeze_at training -- DCNv2 bottlenecks, strided 3x3,
    avg-pool shortcuts): loss terms and the gradient of every trainable tensor, backbone included."""
    from oracle import ppyolo_oracle as orc
    g = golden('g16_train_' + tag)
    S, N, wseed, iseed, fa = [int(v) for v in g['meta']]
    cfg = cfgc()
    _, sd = build_model(cfg, wseed, 'cpu')
    cfg.backbone['freeze_at'] = fa
    monkeypatch.setattr(orc, 'drop_block_train', lambda x, *a, **k: x)        # the golden ran DropBlock in test mode
    x = synth.synth_images(N, S, seed=iseed)
    targets = [T(g['target%d' % i]) for i in range(len(cfg.head['anchor_masks']))]
    torch.set_num_threads(8)
    r = trn.train_step(sd, cfg, x, T(g['gt_bbox']), targets)
    names = [str(n) for n in g['loss_names']]
    got = np.array([float(r['losses'][k]) for k in names], np.float32)
    assert np.array_equal(got, g['loss_values']), (got, g['loss_values'])
    gnames = [str(n) for n in g['grad_names']]
    assert gnames == list(r['grads']) and any(k.startswith('backbone.stage%d' % (fa + 1)) for k in gnames)
    assert fa == 0 or not any(k.startswith('backbone.stage%d' % fa) for k in gnames)
    for k, dig, smp in zip(gnames, g['grad_digest'], g['grad_samples']):
        d = r['grads'][k].double().reshape(-1)
        mine = np.array([d.sum().item(), d.abs().sum().item(), d.pow(2).sum().sqrt().item()])
        # (head: bit-equal as in g12; backbone tensors: autograd accumulates the two consumers of a block input and DCNv2's
        # corner scatter in another order than the reference's module graph -- 1e-7 apart)
        assert np.allclose(mine[2], dig[2], rtol=1e-5, atol=0), k
        step = max(1, d.numel() // 64)
        mine_s = d[::step][:64].float().numpy()
        assert np.abs(mine_s - smp[:len(mine_s)]).max() <= 1e-5 * float(d.abs().max()), k


def test_loss_pieces_have_the_documented_shape_quirks():
    """The IoU-aware term is summed over grid x and broadcast back (reference model/iou_losses.py:241-242); the ignore mask
    uses boxes in (anchor, h, w) order with IoU computed without eps (model/losses.py:56-60, model/matrix_nms.py:31-47)."""
    g = torch.Generator().manual_seed(3)
    N, A, S = 2, 3, 4
    x, y, w, h = [torch.randn(N, A, S, S, generator=g) * 0.3 for _ in range(4)]
    tx, ty = torch.rand(N, A, S, S, generator=g), torch.rand(N, A, S, S, generator=g)
    tw, th = torch.randn(N, A, S, S, generator=g) * 0.2, torch.randn(N, A, S, S, generator=g) * 0.2
    ioup = torch.rand(N, A, S, S, generator=g)
    anchors = [10, 13, 16, 30, 33, 23]
    la = trn.iou_aware_loss(ioup, x, y, w, h, tx, ty, tw, th, anchors, 8, 1.05)
    assert tuple(la.shape) == (N, A, S, 1)
    k = trn.iou_pairs(trn.bbox_transform(x, y, w, h, anchors, 8, False, 1.05), trn.bbox_transform(tx, ty, tw, th, anchors, 8, True, 1.05))
    assert torch.allclose(la[..., 0], (k * -torch.log(ioup + 1e-9)).sum(-1))
    out = torch.randn(N, A * 85, S, S, generator=g)
    b = trn.train_boxes(out, np.array(anchors, np.float32).reshape(-1, 2), 8, 80, 1.05)
    assert tuple(b.shape) == (N, A * S * S, 4)
    o = out.reshape(N, A, 85, S, S)
    a, hh, ww = 2, 1, 3                                        # box index = (a*S + h)*S + w
    cx = (1.05 * torch.sigmoid(o[0, a, 0, hh, ww]) + ww - 0.025) * 8
    bw = torch.exp(o[0, a, 2, hh, ww]) * anchors[2 * a]
    assert torch.allclose(b[0, (a * S + hh) * S + ww, 0], (cx - bw / 2) / S / 8)


def test_ema_matches_reference(golden):
    """ExponentialMovingAverage (reference model/EMA.py) through four updates: golden g13 from the reference's own class."""
    g = golden('g13_ema')
    sa, sb = g['a0'], g['b0']
    for t in range(4):
        sa, d = trn.ema_update(sa, g['a_param%d' % t], t)
        sb, _ = trn.ema_update(sb, g['b_param%d' % t], t)
        assert d == g['decays'][t]
        assert sa.dtype == np.float32 and np.array_equal(sa, g['a_shadow%d' % t]) and np.array_equal(sb, g['b_shadow%d' % t])
