"""Weight formats: the reference's checkpoint is a `torch.save`d state_dict with ITS key
names (1_ppyolo_2x_2pytorch.py:321 -> demo.py:91); this build loads such a file unchanged, and can cache everything it
derives from it (folded BN, KRSC weights, pre-split fp16 / bf16 planes) in a native blob (ppyolo_hip/blob.py)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from ppyolo_hip import blob, synth
from ppyolo_hip._lib import PPYoloHipError


def _fake_executor():
    g = torch.Generator().manual_seed(1)
    ops = [dict(op='stem', w=torch.randn(8, 3, 3, 3, generator=g), scale=torch.rand(8, generator=g), shift=torch.randn(8, generator=g)),
           dict(op='maxpool'),
           dict(op='conv', w=torch.randn(16, 3, 3, 32, generator=g), scale=torch.rand(16, generator=g), shift=torch.randn(16, generator=g),
                w3=torch.randint(-30000, 30000, (3, 16, 3, 3, 32), generator=g).to(torch.int16),
                wf16=(torch.randint(-30000, 30000, (2, 16, 3, 3, 32), generator=g).to(torch.int16), torch.rand(16, generator=g)))]
    setup = [dict(op='conv', w=torch.randn(4, 1, 1, 32, generator=g), scale=torch.ones(4), shift=torch.zeros(4))]
    return types.SimpleNamespace(math='f16x2', plan=types.SimpleNamespace(ops=ops, setup_ops=setup))


def test_blob_file_round_trip_host_side(tmp_path):
    ex = _fake_executor()
    path = str(tmp_path / 'w.ppyblob')
    size = blob.save(ex, path, 'abc123')
    assert size == os.path.getsize(path) and open(path, 'rb').read(8) == b'PPYBLOB1'
    hdr, data = blob.read(path, 'abc123')
    assert hdr['math'] == 'f16x2' and hdr['lists']['ops'][1] is None and data.numel() == hdr['data_bytes']
    own = blob.views(hdr, data, 'cpu')
    for name in ('ops', 'setup_ops'):
        for mine, src in zip(getattr(own.plan, name), getattr(ex.plan, name)):
            for k in ('w', 'scale', 'shift', 'w3'):
                assert (k in mine) == (src.get(k) is not None)
                if k in mine:
                    assert mine[k].dtype == src[k].dtype and torch.equal(mine[k], src[k])
                    assert mine[k].data_ptr() % 256 == data.data_ptr() % 256
            if src.get('wf16') is not None:
                assert torch.equal(mine['wf16'][0], src['wf16'][0]) and torch.equal(mine['wf16'][1], src['wf16'][1])
    with pytest.raises(PPYoloHipError, match='fingerprint'):
        blob.read(path, 'other')
    with open(path, 'r+b') as fh:
        fh.truncate(size - 100)
    with pytest.raises(PPYoloHipError, match='truncated'):
        blob.read(path)
    with open(path, 'wb') as fh:
        fh.write(b'not a blob at all')
    with pytest.raises(PPYoloHipError, match='PPYBLOB1'):
        blob.read(path)
    with pytest.raises(PPYoloHipError, match='ROCm device'):
        blob.load(path, 'cpu')


def test_fingerprint_follows_the_parameters(model_shapes):
    shapes = model_shapes(PPYOLO_r18vd_Config())
    sd = synth.synth_state_dict(shapes, seed=0)
    a = blob.fingerprint(sd)
    assert a == blob.fingerprint({k: v.clone() for k, v in sd.items()})
    k = 'backbone.stage3_0.conv1.conv.weight' if 'backbone.stage3_0.conv1.conv.weight' in sd else sorted(sd)[5]
    sd2 = dict(sd)
    sd2[k] = sd[k].clone()
    sd2[k].view(-1)[7] += 1e-3
    assert blob.fingerprint(sd2) != a
    sd3 = dict(sd)
    t = sd[k].clone().view(-1)
    t[[3, 4]] = t[[4, 3]]                      # same multiset of values, other positions
    sd3[k] = t.view(sd[k].shape)
    assert blob.fingerprint(sd3) != a


def test_pt_checkpoint_keys_are_the_references(golden, model_shapes, tmp_path):
    """A `torch.save`d state_dict in the reference's layout loads strictly: key names, shapes and dtypes are those of the
    reference's own modules (tests/golden/g11_state_dict_layout.npz, made from /root/reference)."""
    g = golden('g11_state_dict_layout')
    for tag, cfgc in (('r18vd', PPYOLO_r18vd_Config), ('r50vd', PPYOLO_2x_Config)):
        keys = [bytes(k).decode() for k in np.split(g[tag + '_keys'], np.cumsum(g[tag + '_keylens'])[:-1])]
        model, sd = build_model(cfgc(), 0, 'cpu')
        mine = model.state_dict()
        assert list(mine.keys()) == keys                    # same names in the same order
        for k, shp, dt in zip(keys, np.split(g[tag + '_shapes'], np.cumsum(g[tag + '_ranks'])[:-1]), g[tag + '_dtypes']):
            assert tuple(mine[k].shape) == tuple(int(v) for v in shp) and str(mine[k].dtype) == bytes(dt).decode().strip(), k
        path = str(tmp_path / (tag + '.pt'))
        torch.save({k: sd[k] for k in keys}, path)          # what 1_ppyolo_2x_2pytorch.py:321 writes
        fresh, _ = build_model(cfgc(), 1, 'cpu')
        res = fresh.load_state_dict(torch.load(path))       # demo.py:91
        assert not res.missing_keys and not res.unexpected_keys
        assert all(torch.equal(fresh.state_dict()[k], sd[k]) for k in keys)


@pytest.mark.gpu
@pytest.mark.parametrize('cfgc,S', [(PPYOLO_r18vd_Config, 320), (PPYOLO_2x_Config, 160)])
def test_pt_round_trip_and_blob_give_the_same_forward(cfgc, S, tmp_path):
    cfg = cfgc()
    model, sd = build_model(cfg, 0, 'cuda')
    x, ims = synth.synth_images(2, S).cuda(), synth.synth_im_size(2).cuda()
    want = [p.clone() for p in model(x, ims)]
    assert any(p.shape[0] > 1 for p in want)
    # (1) checkpoint written from the device-resident model, read back the way demo.py does
    pt = str(tmp_path / 'ckpt.pt')
    torch.save(model.state_dict(), pt)
    m2, _ = build_model(cfg, 1, 'cuda')
    m2.load_state_dict(torch.load(pt))
    got = m2(x, ims)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # (2) native blob: written once, then a fresh model takes every weight tensor from it
    bp = str(tmp_path / 'w.ppyblob')
    size = model.save_native_blob(bp)
    assert size == os.path.getsize(bp) > 4 * sum(v.numel() for k, v in sd.items() if k.endswith('conv.weight'))
    m3, _ = build_model(cfg, 1, 'cuda')
    with pytest.raises(PPYoloHipError, match='fingerprint'):
        m3.load_native_blob(bp)                              # seed-1 parameters: not this blob's
    m3.load_state_dict(torch.load(pt))
    nbytes = m3.load_native_blob(bp)
    got = m3(x, ims)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    ex = m3._plans.executor(x)
    base = m3._plans.blob.storage.data_ptr()
    for op in ex.plan.ops + ex.plan.setup_ops:
        for k in ('w', 'scale', 'shift', 'w3'):
            if op.get(k) is not None:
                assert base <= op[k].data_ptr() < base + nbytes, 'executor re-derived %s instead of taking it from the blob' % k
        if op.get('wf16') is not None:
            assert base <= op['wf16'][0].data_ptr() < base + nbytes
    # (3) another input shape shares the same weight tensors (no second copy in HBM)
    x2 = synth.synth_images(1, S + 32).cuda()
    ex2 = m3._plans.executor(x2)
    assert all(a.get('w') is None or a['w'].data_ptr() == b['w'].data_ptr() for a, b in zip(ex.plan.ops, ex2.plan.ops))
    ref2 = model(x2, synth.synth_im_size(1).cuda())
    got2 = m3(x2, synth.synth_im_size(1).cuda())
    assert all(torch.equal(a, b) for a, b in zip(got2, ref2))
    # (3b) inference-only start without the checkpoint: modules on the meta device, every weight from the blob
    from config import select_backbone, select_head
    from model.ppyolo import PPYOLO
    with torch.device('meta'):
        m4 = PPYOLO(select_backbone(cfg.backbone_type)(**cfg.backbone),
                    select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head))
    m4.eval()
    m4.head.set_dropblock(is_test=True)
    m4.attach_native_blob(bp)
    got = m4(x, ims)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # (4) new parameters drop the blob
    m3.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed=2))
    assert m3._plans.blob is None
