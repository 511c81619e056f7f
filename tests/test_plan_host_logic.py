"""Host-side logic of the product, checked on CPU: the plan the HIP executor replays
(BN fold, KRSC weights, concat slices, CoordConv bias maps, fused residual / upsample, DCN
column order, decode parameters) is interpreted with reference ops (tests/plan_interp.py) and
must reproduce the oracle and the reference-generated goldens."""
import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from oracle import ppyolo_oracle as orc
from plan_interp import CpuPlanRunner
from ppyolo_hip import synth
from ppyolo_hip.runtime import build_plan

TOL = 2e-4   # BN folded into scale/shift + coord terms summed separately -> fp32 roundoff only


@pytest.mark.parametrize('tag,cfgc', [('r18vd_64', PPYOLO_r18vd_Config), ('r50vd_96', PPYOLO_2x_Config)])
def test_plan_matches_golden_features(golden, tag, cfgc):
    g = golden('g6_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    model, sd = build_model(cfg, seed)
    plan = build_plan(model, N, S, S, 'cpu')
    feats, outs, _ = CpuPlanRunner(plan).run(synth.synth_images(N, S, seed=iseed))
    for i, f in enumerate(feats):
        ref = torch.from_numpy(g['feat%d' % i])
        assert f.shape == ref.shape
        assert (f - ref).abs().max() <= TOL * max(1.0, ref.abs().max().item()), (tag, 'feat', i)
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g['out%d' % i])
        assert (o - ref).abs().max() <= TOL * max(1.0, ref.abs().max().item()), (tag, 'out', i)


@pytest.mark.parametrize('tag,cfgc', [('r18vd_320', PPYOLO_r18vd_Config), ('r50vd_160', PPYOLO_2x_Config)])
def test_plan_end_to_end_matches_golden(golden, tag, cfgc):
    g = golden('g7_' + tag)
    S, N, seed, iseed = [int(v) for v in g['meta']]
    cfg = cfgc()
    model, sd = build_model(cfg, seed)
    plan = build_plan(model, N, S, S, 'cpu')
    im_size = torch.from_numpy(g['im_size'])
    _, outs, preds = CpuPlanRunner(plan).run(synth.synth_images(N, S, seed=iseed), im_size)
    for i, p in enumerate(preds):
        ref = torch.from_numpy(g['pred%d' % i])
        assert p.shape == ref.shape
        assert torch.equal(p[:, 0], ref[:, 0])                         # labels, in order
        assert (p[:, 1] - ref[:, 1]).abs().max() <= 1e-4               # scores
        # boxes: 1e-3 px, or 2e-5 of the box side for large boxes -- the plan folds BatchNorm scales into the weights of the four
        # projection-shortcut layers (model/resnet_vd.py), i.e. it rounds in another order than the reference's eager ops, and a
        # box edge moves by (logit noise) x (box side) / 2: what the reference does against itself (tests/golden/g18_*: 1e-5)
        side = torch.maximum(ref[:, 4] - ref[:, 2], ref[:, 5] - ref[:, 3]).clamp_min(1.0)
        assert bool(((p[:, 2:] - ref[:, 2:]).abs().max(dim=1).values <= torch.clamp(2e-5 * side, min=1e-3)).all())


def test_plan_structure_r50():
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg)
    plan = build_plan(model, 2, 160, 160, 'cpu')
    kinds = [o['op'] for o in plan.ops]
    # (74 = 78 - the four projection-shortcut launches, folded into their block's conv3: ConvBlock._emit_folded)
    assert kinds.count('conv') == 74 and kinds.count('dcn') == 3 and kinds.count('spp') == 1
    assert kinds.count('stem') == 1 and kinds.count('maxpool') == 1 and kinds.count('avgpool') == 3
    assert len(plan.setup_ops) == 12                # one CoordConv bias map per coord conv
    # every conv reads a 32-channel-aligned, 16-byte aligned slice
    for o in plan.ops:
        if o['op'] == 'conv':
            assert o['x'].C % 32 == 0 and o['x'].coff % 4 == 0
    d = plan.decode
    assert d['M_total'] == 3 * (5 * 5 + 10 * 10 + 20 * 20)
    assert [len(l['anchors']) for l in d['levels']] == [3, 3, 3]
    assert d['levels'][0]['anchors'][0] == [116.0, 90.0] and d['levels'][2]['downsample'] == 8


def test_cpu_input_raises():
    from ppyolo_hip._lib import PPYoloHipError
    model, _ = build_model(PPYOLO_r18vd_Config())
    with pytest.raises(PPYoloHipError):
        model(torch.zeros(1, 3, 64, 64), torch.tensor([[64., 64.]]))


def test_pool_links_of_the_r50_plan():
    """engine.link_pools (device-free part of HipExecutor._link_pools): the average pools in front of the stage-3 and stage-4
    projection shortcuts go to the 1x1 convolutions that produce their inputs (C64 -> K256 and C128 -> K512, the latter into a
    slice of the stage-3 / route buffer that has a second writer), the one in front of stage 5 (input from a C = 256 layer) stays
    a launch of its own; nothing is linked without f16x2 operands; the linked plan still interprets to the same features."""
    from ppyolo_hip.engine import HipExecutor, link_pools
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg)
    plan = build_plan(model, 2, 160, 160, 'cpu')
    x = synth.synth_images(2, 160, seed=3)
    base_feats, base_outs, _ = CpuPlanRunner(plan).run(x)
    assert link_pools(plan.ops, HipExecutor._op_io, lambda c: False) == 0
    assert link_pools(plan.ops, HipExecutor._op_io, lambda c: True) == 2
    pools = [o for o in plan.ops if o['op'] == 'avgpool']
    assert [o.get('owner') is not None for o in pools] == [True, True, False]
    for o, (C, Kout) in zip(pools[:2], ((64, 256), (128, 512))):
        c = o['owner']
        assert tuple(c['w'].shape) == (Kout, 1, 1, C) and c['pool'] is o['y']
        assert (c['y'].buf, c['y'].coff, c['y'].C) == (o['x'].buf, o['x'].coff, o['x'].C) and c['res'] is not None
    assert HipExecutor._op_io(pools[0]) == ([], []) and pools[0]['y'].buf in HipExecutor._op_io(pools[0]['owner'])[1]
    feats, outs, _ = CpuPlanRunner(plan).run(x)             # (the interpreter pools where the plan says; same tensors)
    for a, b in zip(feats + outs, base_feats + base_outs):
        assert torch.equal(a, b)


def test_maxpool_link_of_the_stem():
    """engine.link_maxpools (device-free part of HipExecutor._link_maxpools): the stem's MaxPool2d(3, 2, 1) goes to the convolution in
    front of it (conv1_3: 3x3, C32 -> K64), whose launch then writes only the pooled tensor -- in the R50vd plan into the channel slice
    of the first block's wide buffer; nothing is linked without f16x2 operands, and nothing when somebody else reads the tensor."""
    from ppyolo_hip.engine import HipExecutor, link_maxpools
    for cfgc in (PPYOLO_2x_Config, PPYOLO_r18vd_Config):
        model, _ = build_model(cfgc())
        plan = build_plan(model, 2, 160, 160, 'cpu')
        assert link_maxpools(plan.ops, HipExecutor._op_io, set(), lambda c: False) == 0
        mp = [o for o in plan.ops if o['op'] == 'maxpool']
        assert len(mp) == 1
        assert link_maxpools(plan.ops, HipExecutor._op_io, {mp[0]['x'].buf}, lambda c: True) == 0          # (a pinned tensor must exist)
        assert link_maxpools(plan.ops, HipExecutor._op_io, set(), lambda c: True) == 1
        c = mp[0]['owner']
        assert tuple(c['w'].shape) == (64, 3, 3, 32) and c['mpool'] is mp[0]['y'] and c['act'] == 'relu'
        assert HipExecutor._op_io(mp[0]) == ([], [])
        assert HipExecutor._op_io(c) == ([c['x'].buf], [mp[0]['y'].buf])          # the full-resolution buffer is not written any more
        del c['mpool'], mp[0]['owner']


def test_split_pairs_of_the_r50_plan():
    """engine.split_pairs (device-free part of HipExecutor._link_splits): which tensors of the R50vd plan may travel pre-split.
    Every bottleneck's conv1 -> conv2 except where the DCNv2 reads conv1's output too (stage 5) -- 13 pairs; conv2 -> conv3
    (the 3x3's output has one reader) -- 13 more; the stem's conv2 -> conv3; the head's chains, incl. the two routes with two
    readers and the tips feeding the output convolutions.  Never a feature map, a head output, a shortcut operand, a pooled or
    upsampled tensor; nothing without f16x2 operands."""
    from ppyolo_hip.engine import HipExecutor, split_pairs
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg)
    plan = build_plan(model, 2, 160, 160, 'cpu')
    pinned = {a.buf for a in list(plan.head_outs) + list(plan.feats)}
    assert split_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: False) == []
    pairs = split_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: True)
    only3 = split_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: True, True)
    shape = lambda o: tuple(o['w'].shape)
    for pr, cons in pairs:
        b = pr['y'].buf
        assert pr['op'] == 'conv' and pr['res'] is None and not pr['ups'] and b not in pinned and plan.buffers[b][3] % 32 == 0
        for c in cons:
            assert c['op'] == 'conv' and c['x'].buf == b and c['x'].C == plan.buffers[b][3] and (c['res'] is None or c['res'].buf != b)
        # nobody else touches the buffer
        for o in plan.ops:
            ins, outs = HipExecutor._op_io(o)
            assert (b not in ins or any(o is c for c in cons)) and (b not in outs or o is pr)
    cons3 = [c for _, cs in pairs for c in cs if shape(c)[1] == 3]
    n3 = sum(1 for o in plan.ops if o['op'] == 'conv' and shape(o)[1] == 3)
    # all 3x3 launches but the three conv_offset (their input is shared with the DCNv2 launch) and the stem's second layer (its
    # input comes from the stem kernel, which is no 'conv' op)
    # (42 = 45 - the conv2 -> conv3 links of the first blocks of stages 2-4: their buffer now has a second writer, the shortcut operand)
    assert n3 == 27 and len(cons3) == 23 and len(pairs) == 42
    assert all(shape(c)[1] == 3 for _, cs in only3 for c in cs) and sum(len(cs) for _, cs in only3) <= len(cons3)
    two = [(pr, cs) for pr, cs in pairs if len(cs) == 2]
    assert len(two) == 2 and all(sorted(shape(c)[1] for c in cs) == [1, 3] for _, cs in two)      # the routes of levels 0 and 1
    # chains: an op can be consumer and producer
    prods = {id(pr) for pr, _ in pairs}
    assert any(id(c) in prods for _, cs in pairs for c in cs)


def test_b2b_pairs_of_the_r50_plan():
    """engine.b2b_pairs (device-free part of HipExecutor._mark_b2b, round 5): conv2 -> conv3 of the two IDENTITY bottlenecks of stage 2
    (64 -> 64 3x3, 64 -> 256 1x1 with the shortcut) run as one launch; not the stage's first block (its conv3 is the folded
    128-channel one), not the deeper stages, nothing in r18vd.  With a pair marked, its first launch reads the shortcut and writes
    the block's output (and the pooled twin), the second does nothing, and conv1 -> conv2 stays a pre-split pair."""
    from ppyolo_hip.engine import HipExecutor, b2b_pairs, split_pairs
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg)
    plan = build_plan(model, 2, 160, 160, 'cpu')
    pinned = {a.buf for a in list(plan.head_outs) + list(plan.feats)}
    assert b2b_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: False) == []
    pairs = b2b_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: True)
    assert len(pairs) == 2
    for a, b in pairs:
        assert tuple(a['w'].shape) == (64, 3, 3, 64) and tuple(b['w'].shape) == (256, 1, 1, 64) and b['res'] is not None
        assert b['x'].buf == a['y'].buf and a['y'].buf not in pinned
    n_before = len(split_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: True))
    for a, b in pairs:
        a['b2b'], b['b2b_of'] = b, a
        assert HipExecutor._op_io(b) == ([], [])
        ins, outs = HipExecutor._op_io(a)
        assert ins == [a['x'].buf, b['res'].buf] and outs[0] == b['y'].buf
    after = split_pairs(plan.ops, HipExecutor._op_io, plan.buffers, pinned, lambda c: True)
    # the two conv2 -> conv3 tensors are gone as links (never written), the conv1 -> conv2 links of those blocks stay
    assert len(after) == n_before - 2
    assert all(any(any(c is a for c in cs) for _, cs in after) for a, _ in pairs)
    for a, b in pairs:
        del a['b2b'], b['b2b_of']
    cfg18 = PPYOLO_r18vd_Config()
    m18, _ = build_model(cfg18)
    p18 = build_plan(m18, 2, 160, 160, 'cpu')
    assert b2b_pairs(p18.ops, HipExecutor._op_io, p18.buffers, set(), lambda c: True) == []


def test_tracked_maximum_blocks_of_the_r50_plan():
    """engine.assign_amax (device-free part of HipExecutor._assign_amax; round 6, the round-5 advisor's aliasing): the wide buffer
    [conv2 output | pooled block input] of a stage's first block (the folded projection shortcut, model/resnet_vd.py:27-33) has a
    block of its OWN for what conv2 / the DCNv2 launch writes and keeps the block of the tensor it was pooled from as a second,
    read-only one: the fused 1x1 reads both, and NO other reader of the block input -- the head's C3 / C4 convolutions, conv1 of
    the block itself -- sees a block that conv2 writes into."""
    from ppyolo_hip.engine import HipExecutor, assign_amax
    cfg = PPYOLO_2x_Config()
    model, _ = build_model(cfg)
    plan = build_plan(model, 2, 160, 160, 'cpu')
    n = assign_amax(plan.ops)
    convs = [o for o in plan.ops if o['op'] in ('conv', 'dcn')]
    assert n > 0 and all(o['amax_out_id'] is not None and 0 <= o['amax_out_id'] < n for o in convs)
    two = [o for o in convs if o.get('amax_in2_id') is not None]
    # the four fused conv3 launches (C = 128, 384, 768, 1536), and stage 2's conv1, which reads the pooled half of the same buffer
    fused = [o for o in two if o['x'].C == plan.buffers[o['x'].buf][3]]
    assert sorted(o['x'].C for o in fused) == [128, 384, 768, 1536] and all(tuple(o['w'].shape)[1:3] == (1, 1) for o in fused)
    writers = {}
    for o in convs:
        writers.setdefault(o['amax_out_id'], []).append(o)
    for o in fused:
        assert o['amax_in_id'] != o['amax_in2_id']
        # the own block is written by conv2 (a 3x3 or the DCNv2 launch) only; the second block by the producers of the block INPUT only
        assert all(w['y'].buf == o['x'].buf for w in writers[o['amax_in_id']])
        assert all(w['y'].buf != o['x'].buf for w in writers.get(o['amax_in2_id'], []))
    # the head's readers of C3 / C4 (feature maps; concat buffers with the upsampled routes) scale by their own producers' maxima
    feat_bufs = {a.buf for a in plan.feats}
    for o in convs:
        if o['x'].buf in feat_bufs and o.get('amax_in2_id') is None:
            assert all(w['y'].buf == o['x'].buf for w in writers[o['amax_in_id']]), 'a foreign writer merges into a feature map\'s block'
    # pooled tensors still inherit: every reader of a buffer that holds ONLY a pooled tensor has its source's block
    for o in plan.ops:
        if o['op'] in ('avgpool', 'maxpool'):
            for c in convs:
                if c['x'].buf == o['y'].buf:
                    assert c['amax_in_id'] is not None


def test_dcn_configuration_ids_and_weight_prep_descriptor_layout():
    """Host-side bookkeeping of round 3 that needs no device: (1) the fused-DCNv2 ids -- [0, 18) = scheme * 6 + four-wave tile, from
    18 the eight-wave f16x2 tiles -- as ops.dcnv2_scheme / dcnv2_configs hand them to the plan and the tuner; every committed
    'dcnf' table entry names an id of its own table's scheme.  (2) ops._WeightPrep mirrors PpyWeightPrep of include/ppyolo_hip.h
    field for field (six pointers, six ints: 72 bytes) -- the table is copied to the device as raw bytes."""
    import ctypes
    import json
    import os
    import re
    from ppyolo_hip import ops
    n = ops.dcnv2_num_configs()
    assert n > 3 * ops.DCN_TILES_4W
    assert [ops.dcnv2_scheme(c) for c in (0, 5, 6, 11, 12, 17, 18, n - 1)] == ['fp32', 'fp32', 'bf16x3', 'bf16x3', 'f16x2', 'f16x2', 'f16x2', 'f16x2']
    assert ops.dcnv2_configs('fp32') == list(range(6)) and ops.dcnv2_configs('bf16x3') == list(range(12))
    assert ops.dcnv2_configs('f16x2') == list(range(18)) + list(range(18, n))
    here = os.path.join(os.path.dirname(ops.__file__))
    for mode, fname in (('fp32', 'tuned_gfx950.json'), ('bf16x3', 'tuned_gfx950_bf16x3.json'), ('f16x2', 'tuned_gfx950_f16x2.json')):
        with open(os.path.join(here, fname)) as fh:
            table = json.load(fh)
        ids = [v[0] for k, v in table.items() if k.startswith('dcnf')]
        assert ids and all(c in ops.dcnv2_configs(mode) for c in ids), (mode, ids)
    # the descriptor: same fields, same order, same size as the C struct
    fields = [f for f, _ in ops._WeightPrep._fields_]
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(here)), 'include', 'ppyolo_hip.h')).read()
    body = hdr[hdr.index('typedef struct PpyWeightPrep {'):hdr.index('} PpyWeightPrep;')]
    names = re.findall(r'[\s\*](\w+)(?=[,;])', body.split('{', 1)[1])
    assert names == fields, (names, fields)
    assert ctypes.sizeof(ops._WeightPrep) == 6 * 8 + 6 * 4
    assert ops.PREP_SPLIT == int(re.search(r'#define PPY_PREP_SPLIT (\d+)', hdr).group(1))
