"""End-to-end GPU parity of the drop-in `model.ppyolo.PPYOLO` against the reference-generated
goldens and the CPU oracle (north-star bar: boxes <= 1e-3 px, scores <= 1e-4, Matrix-NMS keep
indices identical), plus size-independent properties at BASELINE.json's full sizes."""
import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from oracle import ppyolo_oracle as orc
from ppyolo_hip import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_close(a, b, rel, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= rel * scale, '%s: max abs err %.3e at scale %.3e' % (what, err, scale)


@pytest.mark.parametrize('tag,cfgc', [('r18vd_64', PPYOLO_r18vd_Config), ('r50vd_96', PPYOLO_2x_Config)])
def test_backbone_and_head_outputs_golden(golden, tag, cfgc):
    from ppyolo_hip.runtime import build_plan
    from ppyolo_hip.engine import HipExecutor
    g = golden('g6_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    model, _ = build_model(cfgc(), seed, 'cuda')
    plan = build_plan(model, N, S, S, torch.device('cuda'))
    ex = HipExecutor(plan, 'cuda', use_graph=False)
    ex.set_inputs(synth.synth_images(N, S, seed=iseed).cuda(), synth.synth_im_size(N).cuda())
    ex.run()
    torch.cuda.synchronize()
    for i, a in enumerate(plan.feats):
        rel_close(ex.view(a).dense().permute(0, 3, 1, 2), T(g['feat%d' % i]), 1e-4, '%s feat%d' % (tag, i))
    for i, a in enumerate(plan.head_outs):
        rel_close(ex.view(a).dense().permute(0, 3, 1, 2), T(g['out%d' % i]), 1e-4, '%s out%d' % (tag, i))


def _check_preds(preds, refs, keep=None, ref_keep=None, box_tol=1e-3, near_tie=2e-6):
    """Position-wise comparison; rows may only swap places when their scores are within
    `near_tie` of each other (fp32 noise can reorder near-ties end to end; on the committed
    goldens no such pair exists and the comparison is strictly positional)."""
    for i, (p, r) in enumerate(zip(preds, refs)):
        p = p.cpu()
        assert p.shape == r.shape, (i, p.shape, r.shape)
        subst = []
        if keep is not None and r[0, 0] >= 0:
            mine = keep[i][:r.shape[0]].cpu().numpy().astype(np.int64)
            if not np.array_equal(mine, ref_keep[i]):
                pos = {int(k): j for j, k in enumerate(mine)}
                cut = float(r[-1, 1])
                perm = []
                for j, k in enumerate(ref_keep[i]):
                    if int(k) in pos:
                        perm.append(pos[int(k)])
                        assert abs(float(r[j, 1]) - float(r[pos[int(k)], 1])) <= near_tie or pos[int(k)] == j, \
                            'image %d: rows %d/%d reordered but not a near-tie' % (i, j, pos[int(k)])
                    else:       # only admissible at the keep_top_k cut: a different detection with a tied score
                        assert abs(float(r[j, 1]) - cut) <= near_tie, 'image %d: kept set differs' % i
                        subst.append(j)
                        perm.append(-1)
                free = [j for j in range(len(mine)) if j not in set(perm)]      # my rows the reference cut off
                assert len(free) == len(subst)
                for j in subst:
                    perm[j] = free.pop()
                    assert abs(float(p[perm[j], 1]) - cut) <= near_tie, 'image %d: kept set differs' % i
                p = p[torch.tensor(perm)]
        same = torch.ones(r.shape[0], dtype=torch.bool)
        same[subst] = False
        assert torch.equal(p[same, 0], r[same, 0]), 'image %d: labels / order differ' % i
        assert (p[:, 1] - r[:, 1]).abs().max() <= 1e-4, 'image %d scores' % i
        err = (p[same, 2:] - r[same, 2:]).abs().max()
        assert err <= box_tol, 'image %d boxes: %.3e > %.3e' % (i, err, box_tol)


@pytest.mark.parametrize('tag,cfgc', [('r18vd_320', PPYOLO_r18vd_Config), ('r50vd_160', PPYOLO_2x_Config)])
@pytest.mark.parametrize('graph,math', [('0', 'bf16x3'), ('1', 'bf16x3'), ('1', 'fp32'), ('1', 'f16x2'), ('0', 'f16x2')])
def test_end_to_end_golden(golden, tag, cfgc, graph, math, monkeypatch):
    monkeypatch.setenv('PPYOLO_HIP_GRAPH', graph)
    monkeypatch.setenv('PPYOLO_HIP_MATH', math)
    g = golden('g7_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    model, sd = build_model(cfg, seed, 'cuda')
    x, ims = synth.synth_images(N, S, seed=iseed), T(g['im_size'])
    preds = model(x.cuda(), ims.cuda())
    refs = [T(g['pred%d' % i]) for i in range(N)]
    dets, cnt, keep = model.forward_padded(x.cuda(), ims.cuda())
    oracle = orc.ppyolo_forward(sd, cfg, x, ims, return_index=True)
    _check_preds(preds, refs, keep, [o[1] for o in oracle])
    # second call (graph replay) is identical
    again = model(x.cuda(), ims.cuda())
    for a, b in zip(preds, again):
        assert torch.equal(a, b)


def test_demo_call_surface():
    """The reference's harness contract (model/decode_np.py:142-150, :41-57): numpy in,
    list of numpy [K,6] out, three arrays from detect_image."""
    from model.decode_np import Decode
    cfg = PPYOLO_r18vd_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    dec = Decode(model, ['c%d' % i for i in range(80)], True, cfg, for_test=True)
    x = synth.synth_images(1, 320).numpy()
    ims = np.array([[480, 640]], dtype=np.int32)
    pred = dec.predict(x, ims)
    assert isinstance(pred, list) and pred[0].dtype == np.float32 and pred[0].shape[1] == 6
    image, boxes, scores, classes = dec.detect_image(None, x, ims, draw_image=False)
    assert boxes.shape[1] == 4 and scores.ndim == 1 and classes.dtype == np.int32
    ref = orc.ppyolo_forward(sd, cfg, torch.from_numpy(x), torch.tensor([[480., 640.]]))[0]
    assert np.array_equal(classes, ref[:, 0].numpy().astype(np.int32))
    assert np.abs(scores - ref[:, 1].numpy()).max() <= 1e-4


def test_decode_harness_golden():
    """The reference's own Decode.detect_image / detect_batch on numpy inputs (tests/golden/g9_decode_harness.npz,
    made from /root/reference by tools/make_goldens.py g9) against this repo's Decode on the same inputs."""
    import os
    from model.decode_np import Decode
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_decode_harness.npz'))
    S, N, wseed, iseed = [int(v) for v in g['meta']]
    cfg = PPYOLO_r18vd_Config()
    model, _ = build_model(cfg, wseed, 'cuda')
    dec = Decode(model, ['c%d' % i for i in range(80)], True, cfg, for_test=True)
    x = synth.synth_images(N, S, seed=iseed).numpy()
    ims = g['im_size']
    image, boxes, scores, classes = dec.detect_image(None, x[:1], ims[:1], draw_image=False)
    assert image is None and boxes.dtype == np.float32 and scores.dtype == np.float32 and classes.dtype == np.int32
    assert np.array_equal(classes, g['classes'])
    assert np.abs(scores - g['scores']).max() <= 1e-4 and np.abs(boxes - g['boxes']).max() <= 1e-3
    imgs, bb, ss, cc = dec.detect_batch([None] * N, x, ims, draw_image=False)
    for i in range(N):
        assert np.array_equal(cc[i], g['b_classes%d' % i])
        assert np.abs(ss[i] - g['b_scores%d' % i]).max() <= 1e-4 and np.abs(bb[i] - g['b_boxes%d' % i]).max() <= 1e-3


def test_empty_result_sentinel():
    """Nothing above the threshold -> the reference's [[-1]*6] row (matrix_nms.py:113)."""
    cfg = PPYOLO_r18vd_Config()
    cfg.nms_cfg['score_threshold'] = 0.999
    model, _ = build_model(cfg, 0, 'cuda')
    out = model(synth.synth_images(2, 64).cuda(), synth.synth_im_size(2).cuda())
    for o in out:
        assert tuple(o.shape) == (1, 6) and torch.all(o == -1)


@pytest.mark.parametrize('cfgc,S', [(PPYOLO_r18vd_Config, 416), (PPYOLO_2x_Config, 608)])
def test_full_size_vs_oracle_and_batch_properties(cfgc, S):
    """BASELINE.json configs[1] / configs[2] at batch 8.  Images 0-1 are checked against the CPU
    oracle.  Box tolerance at full size: two correct fp32 implementations (MKLDNN vs the MFMA
    fma chain) differ by summation-order noise that the reference itself has against exact
    arithmetic (see test_fp64_three_way): a 1000-px-tall box whose log-size logit moves by 5e-6
    (40 fp32 ulps after ~70 layers) moves by 2.5e-3 px.  Measured max 1.0e-3 .. 2.5e-3 px on a
    480x640 image depending on the tile configuration, so the bar here is 5e-3 px x (image extent
    / 640); the committed reference goldens are held to the north-star 1e-3 px.  Size-independent properties cover the other
    images: permuting the batch permutes the results (bit-exactly without DCN; images are independent), a
    batch-of-1 run agrees to fp32 noise (different tile configs => different summation
    order), and the run is repeatable bit for bit."""
    cfg = cfgc()
    model, sd = build_model(cfg, 0, 'cuda')
    N = 8
    x = synth.synth_images(N, S)
    ims = torch.tensor([[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]] * 2)
    preds = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    dets, cnt, keep = model.forward_padded(x.cuda(), ims.cuda())
    keep = keep.clone()
    ref = orc.ppyolo_forward(sd, cfg, x[:2], ims[:2], return_index=True)
    _check_preds(preds[:2], [r[0] for r in ref], keep, [r[1] for r in ref], box_tol=5e-3)
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4])
    pp = [p.cpu() for p in model(x[perm].cuda(), ims[perm].cuda())]
    has_dcn = bool(cfg.backbone.get('dcn_v2_stages'))
    for j, i in enumerate(perm.tolist()):
        if has_dcn:
            # the reference's DCNv2 folds the batch index into the fp32 row coordinate before floor()
            # (custom_layers.py:626-633), so ITS results depend on the batch position at the 1e-5
            # level; this build reproduces that arithmetic, hence a tolerance instead of equality
            _check_preds([pp[j]], [preds[i]], box_tol=5e-3 * max(1.0, float(ims[i].max()) / 640.0))
        else:
            assert torch.equal(pp[j], preds[i]), 'image %d depends on its position in the batch' % i
    for i in (3, 7):
        solo = model(x[i:i + 1].cuda(), ims[i:i + 1].cuda())[0].cpu()
        scale = float(ims[i].max()) / 640.0
        _check_preds([solo], [preds[i]], box_tol=5e-3 * max(1.0, scale))
    again = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    for a, b in zip(preds, again):
        assert torch.equal(a, b)
    for p in preds:
        assert 1 <= p.shape[0] <= 100 and torch.all(p[:-1, 1] >= p[1:, 1]), 'scores not sorted descending'


@pytest.mark.parametrize('math', ['f16x2', 'bf16x3', 'fp32'])
def test_fp64_three_way(math, monkeypatch):
    """What "parity" means for a 70-layer fp32 network: run the oracle in float64 as the exact
    answer; the HIP path -- in both math modes: bf16x3 split products on the bf16 MFMA (default) and the
    exact-fp32 MFMA -- must be as close to it as the reference's own fp32 forward is (measured on MI355X:
    head outputs rms error 1.3-2.1e-6 HIP vs 1.7-3.0e-6 reference)."""
    monkeypatch.setenv('PPYOLO_HIP_MATH', math)
    cfg = PPYOLO_2x_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    S, N = 320, 2
    x = synth.synth_images(N, S)
    ex = model._plans.executor(x.cuda())
    ex.set_inputs(x.cuda(), synth.synth_im_size(N).cuda())
    ex.run()
    torch.cuda.synchronize()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, o32 = orc.backbone_and_head(sd, cfg, x)
    _, o64 = orc.backbone_and_head(sd64, cfg, x.double())
    for i, a in enumerate(ex.plan.head_outs):
        h = ex.view(a).dense().permute(0, 3, 1, 2).cpu().double()
        e_hip = (h - o64[i]).pow(2).mean().sqrt().item()
        e_ref = (o32[i].double() - o64[i]).pow(2).mean().sqrt().item()
        print('math %s level %d: rms error vs fp64: HIP %.3e, reference fp32 %.3e' % (math, i, e_hip, e_ref))
        assert e_hip <= 1.5 * e_ref + 1e-7, 'level %d: HIP rms error %.3e vs reference fp32 %.3e' % (i, e_hip, e_ref)
        assert (h - o64[i]).abs().max() <= 1e-4
    assert ex.math == math and (math == 'fp32') == all(op.get('w3') is None for op in ex.plan.ops)
    assert (math == 'f16x2') == any(op.get('wf16') is not None for op in ex.plan.ops)


def test_autotuned_plan_same_answer():
    cfg = PPYOLO_r18vd_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    x, ims = synth.synth_images(2, 224).cuda(), synth.synth_im_size(2).cuda()
    base = [p.clone() for p in model(x, ims)]
    ex = model._plans.executor(x)
    rep = ex.autotune(iters=1)
    assert len(rep) > 10
    tuned = model(x, ims)
    _check_preds([t.cpu() for t in tuned], [b.cpu() for b in base])


@pytest.mark.parametrize('cfgc,S,N', [(PPYOLO_r18vd_Config, 352, 3), (PPYOLO_2x_Config, 224, 1), (PPYOLO_2x_Config, 288, 5)])
def test_other_input_sizes_use_heuristic_configs(cfgc, S, N):
    """Shapes that are NOT in the measured tile table (tuned_gfx950.json) fall back to the cost
    model inside the library; results must still match the oracle."""
    cfg = cfgc()
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, S, seed=77)
    ims = torch.tensor([[480., 640.], [375., 500.], [608., 608.], [333., 500.], [500., 375.]])[:N]
    preds = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    dets, cnt, keep = model.forward_padded(x.cuda(), ims.cuda())
    ref = orc.ppyolo_forward(sd, cfg, x, ims, return_index=True)
    _check_preds(preds, [r[0] for r in ref], keep.clone(), [r[1] for r in ref], box_tol=3e-3)


@pytest.mark.parametrize('cfgc,S,N,depth', [(PPYOLO_r18vd_Config, 320, 4, 2), (PPYOLO_2x_Config, 224, 2, 3)])
def test_in_flight_matches_forward(cfgc, S, N, depth):
    """runtime.InFlight (several batches on the device at once, one lane each) returns for every batch exactly what
    model.forward returns for it, in any collect order, and refuses to overwrite an uncollected lane."""
    from ppyolo_hip._lib import PPYoloHipError
    cfg = cfgc()
    model, sd = build_model(cfg, 0, 'cuda')
    batches = [(synth.synth_images(N, S, seed=300 + i).cuda(), synth.synth_im_size(N).cuda()) for i in range(7)]
    want = [[p.clone() for p in model(x, ims)] for x, ims in batches]
    assert any(p.shape[0] > 1 for w in want for p in w)
    pipe = model.in_flight(depth)
    tickets, got = [], {}
    for i, (x, ims) in enumerate(batches):
        if len(tickets) == depth:
            with pytest.raises(PPYoloHipError):
                pipe.submit(x, ims)
            j, t = tickets.pop(0)
            got[j] = t.result()
        tickets.append((i, pipe.submit(x, ims)))
    for j, t in reversed(tickets):                      # newest first
        dets, cnt, keep = t.padded()
        assert int(cnt[0]) == (0 if want[j][0][0, 0] == -1 else want[j][0].shape[0])
        got[j] = t.result()
        with pytest.raises(PPYoloHipError):
            t.result()
    for j in range(len(batches)):
        for a, b in zip(got[j], want[j]):
            assert torch.equal(a, b), 'batch %d differs between InFlight and forward' % j
    # ... and against the oracle for one batch
    ref = orc.ppyolo_forward(sd, cfg, batches[3][0].cpu(), batches[3][1].cpu())
    _check_preds([g.cpu() for g in got[3]], ref, box_tol=3e-3)


@pytest.mark.parametrize('math', ['f16x2', 'bf16x3'])
def test_two_lanes_are_bit_stable(math, monkeypatch):
    """Two lanes computing different batches at the same time give, in every buffer, what one executor gives alone.
    (With packed-fp32 VALU ops in the kernels the decode of one lane was corrupted by the other lane's convolutions
    in 4-40 % of the rounds -- ppyolo_hip/build.py, tools/pk_hazard_probe.py.)"""
    monkeypatch.setenv('PPYOLO_HIP_MATH', math)
    cfg = PPYOLO_r18vd_Config()
    model, _ = build_model(cfg, 0, 'cuda')
    N, S, NB = 4, 320, 4
    batches = [(synth.synth_images(N, S, seed=300 + i).cuda(), synth.synth_im_size(N).cuda()) for i in range(NB)]
    ex0 = model._plans.executor(batches[0][0])

    def snap(e):
        d = {'head%d' % i: e.view(a).dense().clone() for i, a in enumerate(e.plan.head_outs)}
        d.update(boxes=e.boxes.clone(), cand_count=e.cand_count.clone(), out_dets=e.out_dets.clone(),
                 out_count=e.out_count.clone(), out_keep=e.out_keep.clone())
        return d
    want = []
    for x, ims in batches:
        model(x, ims)
        want.append(snap(ex0))
    lanes = model.in_flight(2).lanes(batches[0][0])
    for r in range(60):
        pair = [(2 * r) % NB, (2 * r + 1 + r // NB) % NB]
        for k, (e, st) in enumerate(lanes):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                e.set_inputs(*batches[pair[k]])
                e.run()
        torch.cuda.synchronize()
        for k, (e, _) in enumerate(lanes):
            for nm, g in snap(e).items():
                assert torch.equal(g, want[pair[k]][nm]), 'round %d lane %d: %s differs from the solo run' % (r, k, nm)


def test_in_flight_follows_new_weights():
    """load_state_dict drops the plans; lanes built from them must not keep computing with the old weights."""
    cfg = PPYOLO_r18vd_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    x, ims = synth.synth_images(2, 224, seed=9).cuda(), synth.synth_im_size(2).cuda()
    pipe = model.in_flight(2)
    first = pipe.submit(x, ims).result()
    sd2 = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1)
    model.load_state_dict(sd2)
    want = [p.clone() for p in model(x, ims)]
    got = pipe.submit(x, ims).result()
    assert any(a.shape != b.shape or not torch.equal(a, b) for a, b in zip(first, want))      # the weights do matter
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_two_branch_plan_is_opt_in_and_same_answer(monkeypatch):
    """PPYOLO_HIP_STREAMS=2 (independent branches on a second stream; eager here) gives the one-branch results."""
    cfg = PPYOLO_r18vd_Config()
    x, ims = synth.synth_images(4, 256, seed=5).cuda(), synth.synth_im_size(4).cuda()
    model, _ = build_model(cfg, 0, 'cuda')
    base = [p.clone() for p in model(x, ims)]
    assert not model._plans.executor(x).multi_stream
    monkeypatch.setenv('PPYOLO_HIP_STREAMS', '2')
    monkeypatch.setenv('PPYOLO_HIP_GRAPH', '0')
    model2, _ = build_model(cfg, 0, 'cuda')
    model2._plans.use_graph = False
    got = model2(x, ims)
    assert model2._plans.executor(x).multi_stream
    for a, b in zip(got, base):
        assert torch.equal(a, b)


@pytest.mark.parametrize('S', [320, 352, 384, 416, 448, 480, 512, 544, 576, 608])
def test_every_reference_input_size(S):
    """The reference's multi-scale set (config randomShape sizes, also its eval sizes): r18vd at batch 2 against the
    oracle at every one of them -- most of these shapes are not in the measured tile tables."""
    cfg = PPYOLO_r18vd_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(2, S, seed=S)
    ims = torch.tensor([[480., 640.], [333., 500.]])
    preds = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    dets, cnt, keep = model.forward_padded(x.cuda(), ims.cuda())
    ref = orc.ppyolo_forward(sd, cfg, x, ims, return_index=True)
    _check_preds(preds, [r[0] for r in ref], keep.clone(), [r[1] for r in ref], box_tol=3e-3)


@pytest.mark.parametrize('N', [1, 3, 7, 16])
def test_batch_sizes(N):
    """Odd and large batches (row tiles that straddle several images, per-image scales): r18vd-256 vs the oracle on the
    first and last image, and against the same image run alone (another plan: other tiles / split-K, hence another
    fp32 summation order -- equal within rounding, not bit for bit; bit-exactness holds within one batch size)."""
    cfg = PPYOLO_r18vd_Config()
    model, sd = build_model(cfg, 0, 'cuda')
    x = synth.synth_images(N, 256, seed=40 + N)
    ims = synth.synth_im_size(N)
    preds = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    for i in sorted({0, N - 1}):
        ref = orc.ppyolo_forward(sd, cfg, x[i:i + 1], ims[i:i + 1])
        _check_preds([preds[i]], ref, box_tol=3e-3)
        alone = model(x[i:i + 1].cuda(), ims[i:i + 1].cuda())[0].cpu()
        _check_preds([preds[i]], [alone], box_tol=1e-3)


def test_graft_entry_smoke():
    """The driver's round-end smoke check, run as a test so that a regression shows up here first (the library
    once reported PyTorch's own stale hipGetLastError as a launch failure on exactly this path)."""
    import __graft_entry__ as ge
    ge.smoke()


def test_fresh_process_build_then_smoke():
    """build() loads the library before anything touched the GPU; smoke() in the same fresh process must still run
    (the library used to bind to the system HIP runtime instead of torch's when it was loaded first: hipErrorNoDevice
    at the first launch)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke()'], cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0 and 'smoke ok' in r.stdout, r.stdout[-2000:]
