"""Plan builder + HIP executor for the PP-YOLO inference path.

The reference executes `PPYOLO.forward` as ~1000 eager ATen calls (reference
model/ppyolo.py:19-22 down to model/custom_layers.py:243-253).  Here the module tree is
walked ONCE per input shape into a flat plan: a list of kernel launches over pre-allocated
NHWC buffers, with BatchNorm folded to a per-channel affine, weights re-laid to KRSC, concats
turned into channel slices of one wide buffer (producers write their slice in place),
CoordConv turned into a precomputed per-position bias, nearest-x2 upsample fused into the
producing conv's store, and the residual add + ReLU fused into the conv epilogue.  The plan
is replayed with direct C-ABI calls on the current HIP stream, or as one captured hipGraph.

A plan is pure data (`Plan.ops` dicts over buffer ids), so the host logic can be checked on
a CPU-only box by interpreting the same plan with reference ops (tests/plan_interp.py);
the product executor below has no CPU path.
"""
import collections
import json
import os

import torch

from . import ops as K
from ._lib import PPYoloHipError

A = collections.namedtuple('A', 'buf coff C N H W')   # activation: channel slice of a buffer


def fold_bn(bn, bias, K_out, device):
    """Eval-mode BatchNorm2d as y = x*scale + shift (torch computes alpha = weight*invstd,
    beta = bias - mean*alpha on CPU); a conv bias without BN is scale=1, shift=bias."""
    if bn is not None:
        invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        scale = bn.weight.detach().float() * invstd
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        if bias is not None:
            shift = shift + bias.detach().float() * scale
    else:
        scale = torch.ones(K_out, dtype=torch.float32, device=device)
        shift = bias.detach().float().clone() if bias is not None else torch.zeros(K_out, dtype=torch.float32,
                                                                                   device=device)
    return scale.contiguous(), shift.contiguous()


_HERE = os.path.dirname(os.path.abspath(__file__))
_TUNED_PATHS = {'fp32': os.path.join(_HERE, 'tuned_gfx950.json'), 'bf16x3': os.path.join(_HERE, 'tuned_gfx950_bf16x3.json'),
                'f16x2': os.path.join(_HERE, 'tuned_gfx950_f16x2.json')}
_tuned = {}
NUM_FP32_CFGS = 31       # tile configuration ids below this are the exact-fp32 MFMA kernels (conv_igemm.hip)
NUM_X3_CFGS = 9          # then the bf16x3 kernels [31, 40), then the f16x2 kernels [40, 67): 9 tiles x {2, 3, 4} LDS stages


def math_mode():
    """How the convolution inner products are evaluated (PPYOLO_HIP_MATH):
      'f16x2' (default):  2-term fp16 split of both operands after power-of-two scaling into the fp16 range (weights per output
                channel at plan time, activations by the maximum their producers track), 3 partial products on the fp16
                MFMA, fp32 accumulate -- fp32-grade results at half the MFMA work of bf16x3; layers whose input
                maximum is not tracked (stem side) or that carry a CoordConv bias map use the modes below;
      'bf16x3': exact 3-term bf16 split of both fp32 operands, 6 partial products on the bf16 MFMA,
                fp32 accumulate -- fp32-grade results (csrc/conv_x3.hip) at 6/16 of the fp32 MFMA cost; the
                measured table may still pick an exact-fp32 kernel for a layer where that is faster;
      'fp32':   v_mfma_f32_32x32x2_f32 only (a k-ordered fp32 fma chain)."""
    m = os.environ.get('PPYOLO_HIP_MATH', 'f16x2')
    if m not in _TUNED_PATHS:
        raise PPYoloHipError('PPYOLO_HIP_MATH must be one of %s' % sorted(_TUNED_PATHS))
    return m


def tune_key(op, with_g=True):
    """Shape key of a conv / DCN launch in the measured (tile config, split-K) table.  Launches that can use the
    f16x2 kernels (split fp16 weights at hand, tracked input maximum) carry ':f' -- the same shape without them
    (CoordConv layers, stem side) needs its own entry."""
    x = op['x']
    Kout, R, S, C = op['w'].shape
    f16 = op.get('wf16') is not None and op.get('amax_in_id') is not None
    # ('dcnf': ids of the fused DCNv2 kernel, ops.dcnv2_num_configs -- not the convolution's numbering)
    # ':p': the layer also owns the 2x2 average of its output (HipExecutor._link_pools)
    # ':g': the layer's input can arrive pre-split from its one producer (HipExecutor._mark_split_candidates): its main loop has
    # no split work, another tile may win -- such entries are measured in that form; without one the plain entry is used
    return '%s:N%d:H%d:W%d:C%d:K%d:R%d:s%d%s%s%s' % ('dcnf' if op['op'] == 'dcn' else op['op'], x.N, x.H, x.W, C, Kout, R,
                                                     op['stride'], ':f' if f16 else '', ':p' if op.get('pool') is not None else '',
                                                     ':g' if (op.get('gp_in') and with_g) else '')


def tuned_table(mode=None):
    """Measured per-shape choices (written by HipExecutor.autotune on an MI355X and committed
    as tuned_gfx950.json / tuned_gfx950_bf16x3.json, one table per math mode; PPYOLO_HIP_TUNE_CACHE
    names an extra file).  Unknown shapes fall back to the cost model inside libppyolo_hip.so."""
    mode = mode or math_mode()
    if mode not in _tuned:
        _tuned[mode] = {}
        for path in (_TUNED_PATHS[mode], os.environ.get('PPYOLO_HIP_TUNE_CACHE')):
            if path and os.path.exists(path):
                with open(path) as fh:
                    _tuned[mode].update(json.load(fh))
    return _tuned[mode]


class Plan(object):
    def __init__(self, N, H, W):
        self.N, self.H, self.W = N, H, W
        self.buffers = []        # (N, H, W, ld)
        self.setup_ops = []      # run once after buffers are bound (CoordConv bias maps)
        self.ops = []            # run every forward
        self.consts = {}         # buffer id -> tensor to preload (coord grids)
        self.feats = []          # backbone feature maps (A), for per-stage checks
        self.head_outs = []      # raw head outputs (A)
        self.decode = None       # decode / NMS parameters


def _meta(*shape):
    return torch.empty(shape, dtype=torch.float32, device='meta')


class Builder(object):
    def __init__(self, N, H, W, device, skeleton=False):
        """`skeleton`: record the plan WITHOUT touching parameter data -- weight-carrying ops get shape-only placeholders
        (meta tensors) that the executor replaces by the tensors of a weight owner: another executor of the same model
        (any input shape; folded / split weights do not depend on it) or a native blob (ppyolo_hip/blob.py)."""
        self.plan = Plan(N, H, W)
        self.device = device
        self.skeleton = skeleton
        self._coord_cache = {}
        self._stream = 0
        self.wide_of_block = {}      # ConvBlock (by id) -> wide buffer its input was placed into (model/resnet_vd.py, shortcut fold)

    # ---- buffers -------------------------------------------------------------------------
    def new_buf(self, N, H, W, ld):
        self.plan.buffers.append((N, H, W, ld))
        return len(self.plan.buffers) - 1

    def new_act(self, N, H, W, C):
        return A(self.new_buf(N, H, W, C), 0, C, N, H, W)

    def slice(self, a, coff, C):
        return A(a.buf, a.coff + coff, C, a.N, a.H, a.W)

    def _emit(self, op, setup=False):
        if not setup:
            op['stream'] = self._stream
        (self.plan.setup_ops if setup else self.plan.ops).append(op)

    def side(self):
        """Context manager: ops emitted inside form an independent branch (projection shortcut,
        head tip/output convs) that the executor may run on a second HIP stream, concurrently with
        the main chain -- per-layer launches at batch 8 leave CUs idle (tile quantisation, short
        tails), a second kernel's workgroups fill them."""
        b = self

        class _Side(object):
            def __enter__(self_inner):
                self_inner.prev = b._stream
                b._stream = 1

            def __exit__(self_inner, *a):
                b._stream = self_inner.prev
        return _Side()

    # ---- ops -----------------------------------------------------------------------------
    def stem(self, weight, scale, shift, act='relu'):
        p = self.plan
        Ho, Wo = K.conv_out_hw(p.H, p.W, 3, 3, 2, 1)
        y = self.new_act(p.N, Ho, Wo, weight.shape[0])
        w = _meta(*weight.shape) if self.skeleton else weight.detach().float().contiguous()
        self._emit(dict(op='stem', y=y, w=w, scale=scale, shift=shift, act=act))
        return y

    def conv(self, x, weight, scale, shift, stride=1, act=None, res=None, out=None, ups=False, coord=False,
             setup=False):
        """weight: [K, C(+2 if coord), R, S] in the reference's KCRS layout."""
        Kout, Cin, R, S = weight.shape
        pad = (R - 1) // 2
        w = weight if self.skeleton else weight.detach().float()
        posb = None
        if coord:
            assert Cin == x.C + 2
            posb = self._coord_bias(w[:, x.C:], x.H, x.W, stride, pad)
            w = w[:, :x.C]
        else:
            assert Cin == x.C, (Cin, x.C)
        w_krsc = _meta(Kout, R, S, x.C) if self.skeleton else w.permute(0, 2, 3, 1).contiguous()
        Ho, Wo = K.conv_out_hw(x.H, x.W, R, S, stride, pad)
        y = None
        if out is None and Kout % 4 and Kout >= 64 and res is None and posb is None and not ups \
                and os.environ.get('PPYOLO_HIP_PAD_K', '1') == '1':
            # K that is no multiple of 4 (the 258-channel head outputs) would run the 4-byte scalar epilogue: the launch computes
            # K rounded up to 4 with all-zero extra filters (scale 1, shift 0) into a buffer whose pixel stride has the room;
            # consumers see the first K channels (the decode kernel reads such rows with 16-byte loads).  Not for the narrow
            # 27-channel offset / mask convolutions of DCNv2: measured neutral.
            Kp = (Kout + 3) // 4 * 4
            buf = self.new_buf(x.N, Ho, Wo, Kp)
            out, y = A(buf, 0, Kout, x.N, Ho, Wo), A(buf, 0, Kp, x.N, Ho, Wo)
            if self.skeleton:
                w_krsc = _meta(Kp, R, S, x.C)
            else:
                w_krsc = torch.cat([w_krsc, torch.zeros((Kp - Kout, R, S, x.C), dtype=w_krsc.dtype, device=w_krsc.device)])
                scale = torch.cat([scale, torch.ones(Kp - Kout, dtype=scale.dtype, device=scale.device)])
                shift = torch.cat([shift, torch.zeros(Kp - Kout, dtype=shift.dtype, device=shift.device)])
        if out is None:
            out = self.new_act(x.N, Ho * (2 if ups else 1), Wo * (2 if ups else 1), Kout)
        assert out.C == Kout and out.H == Ho * (2 if ups else 1) and out.W == Wo * (2 if ups else 1)
        if res is not None:
            assert (res.C, res.H, res.W) == (Kout, Ho, Wo)
        self._emit(dict(op='conv', x=x, y=out if y is None else y, w=w_krsc, scale=scale, shift=shift, stride=stride, pad=pad, act=act,
                        res=res, posb=posb, ups=ups, cfg=-1, splitk=0), setup)
        return out

    def _coord_bias(self, w_coord, H, W, stride, pad):
        """Contribution of CoordConv's two appended channels (reference
        model/custom_layers.py:261-272: x_range then y_range, each in [-1,1]) to the following
        conv.  It does not depend on the image, so it is computed ONCE per plan by running the
        conv kernel itself over the coordinate grid (channels padded 2 -> 32) and then added as
        a per-position bias [Ho*Wo][K] in the epilogue of the real conv."""
        key = (H, W)
        if key not in self._coord_cache:
            g = self.new_act(1, H, W, 32)
            grid = torch.zeros((1, H, W, 32), dtype=torch.float32, device=self.device)
            xr = torch.arange(0, W, dtype=torch.float32, device=self.device) / (W - 1) * 2.0 - 1
            yr = torch.arange(0, H, dtype=torch.float32, device=self.device) / (H - 1) * 2.0 - 1
            grid[0, :, :, 0] = xr.view(1, W)
            grid[0, :, :, 1] = yr.view(H, 1)
            self.plan.consts[g.buf] = grid
            self._coord_cache[key] = g
        g = self._coord_cache[key]
        Kout, two, R, S = w_coord.shape
        if self.skeleton:
            w32, one, zero = _meta(Kout, 32, R, S), None, None
        else:
            w32 = torch.zeros((Kout, 32, R, S), dtype=torch.float32, device=self.device)
            w32[:, :2] = w_coord
            one = torch.ones(Kout, dtype=torch.float32, device=self.device)
            zero = torch.zeros(Kout, dtype=torch.float32, device=self.device)
        out = self.conv(g, w32, one, zero, stride=stride, act=None, setup=True)
        return out          # A of shape [1,Ho,Wo,K]; bound to its tensor by the executor

    def maxpool(self, x, out=None):
        Ho, Wo = K.conv_out_hw(x.H, x.W, 3, 3, 2, 1)
        y = self.new_act(x.N, Ho, Wo, x.C) if out is None else out
        assert (y.N, y.H, y.W, y.C) == (x.N, Ho, Wo, x.C)
        self._emit(dict(op='maxpool', x=x, y=y))
        return y

    def avgpool(self, x, out=None):
        y = self.new_act(x.N, x.H // 2, x.W // 2, x.C) if out is None else out
        assert (y.N, y.H, y.W, y.C) == (x.N, x.H // 2, x.W // 2, x.C)
        self._emit(dict(op='avgpool', x=x, y=y))
        return y

    def spp(self, x_slot0):
        """x_slot0: slice [0,C) of a [N,H,W,4C] buffer; fills slices 1..3 with pool 5/9/13."""
        C = x_slot0.C
        ys = [self.slice(x_slot0, C * i, C) for i in (1, 2, 3)]
        self._emit(dict(op='spp', x=x_slot0, y5=ys[0], y9=ys[1], y13=ys[2]))
        return A(x_slot0.buf, x_slot0.coff, 4 * C, x_slot0.N, x_slot0.H, x_slot0.W)

    def dcn(self, x, om, weight, scale, shift, stride, act, out=None):
        Kout = weight.shape[0]
        w_krsc = (_meta(Kout, weight.shape[2], weight.shape[3], weight.shape[1]) if self.skeleton
                  else weight.detach().float().permute(0, 2, 3, 1).contiguous())
        Ho, Wo = K.dcn_out_hw(x.H, x.W, stride, 1)
        assert (om.H, om.W, om.C) == (Ho, Wo, 27)
        y = self.new_act(x.N, Ho, Wo, Kout) if out is None else out
        assert (y.N, y.H, y.W, y.C) == (x.N, Ho, Wo, Kout)
        self._emit(dict(op='dcn', x=x, om=om, y=y, w=w_krsc, scale=scale, shift=shift, stride=stride, pad=1, act=act,
                        cfg=-1, splitk=0))
        return y


def assign_amax(ops):
    """Blocks of tracked per-image maxima (HipExecutor._assign_amax): sets op['amax_out_id'] / ['amax_in_id'] / ['amax_in2_id'],
    returns the number of blocks.  A conv / DCN launch merges max|y| into the block of its output BUFFER (the writers of a concat
    buffer share one); a pooled tensor inherits the block of its input.  Round 6: when a convolution writes into a buffer that so
    far only holds a pooled tensor -- the folded projection shortcut's wide buffer [conv2 output | pooled block input] -- the
    buffer gets a block of its OWN for what convolutions write and keeps the inherited one as a second, read-only block
    ('amax_in2_id' of its readers): the block input's other readers (the head's C3 / C4 convolutions) no longer see conv2's
    maximum (round-5 advisor)."""
    amax_of, aux_of, inherited, nblocks = {}, {}, set(), 0
    for op in ops:
        t = op['op']
        if t in ('conv', 'dcn'):
            b = op['y'].buf
            if b in inherited:
                inherited.discard(b)
                aux_of[b] = amax_of.pop(b)
            if b not in amax_of:
                amax_of[b] = nblocks
                nblocks += 1
            op['amax_out_id'] = amax_of[b]
            op['amax_in_id'] = amax_of.get(op['x'].buf)
            op['amax_in2_id'] = aux_of.get(op['x'].buf)
        elif t in ('maxpool', 'avgpool'):
            src = amax_of.get(op['x'].buf)
            if src is not None:
                yb = op['y'].buf
                if yb in amax_of and yb not in inherited:      # a convolution wrote into this buffer first: second block
                    aux_of[yb] = src
                else:
                    amax_of[yb] = src
                    inherited.add(yb)
        elif t == 'stem':
            amax_of[op['y'].buf] = nblocks
            op['amax_out_id'] = nblocks
            nblocks += 1
    return nblocks


def link_pools(ops, op_io, has_f16, two_streams=False):
    """Host logic of HipExecutor._link_pools, device-free (tests/test_plan_host_logic.py): gives every 'avgpool' op whose input
    slice is written by exactly one 1x1 / stride-1 convolution that ppy_conv1x1_expand_f32 accepts (C = 64 with K % 64 == 0, or
    C = 128 with K % 128 == 0; K / 64 resp. K / 128 a power of two <= 16; no upsampling, no position bias; has_f16(op): f16x2
    operands at hand) to that convolution: conv['pool'] = the pooled slice, avgpool['owner'] = the convolution.  op_io(op) ->
    (input buffer ids, output buffer ids).  Returns the number of links."""
    n = 0
    for i, op in enumerate(ops):
        if op['op'] != 'avgpool':
            continue
        x = op['x']
        # (a route buffer has several writers, each of its own channel slice: the producer is the one that writes x's)
        prods = [o for o in ops[:i] if x.buf in op_io(o)[1] and (o['op'] != 'conv' or (o['y'].coff < x.coff + x.C
                                                                                      and x.coff < o['y'].coff + o['y'].C))]
        if len(prods) != 1 or prods[0]['op'] != 'conv':
            continue
        c = prods[0]
        Kout, R, S, C = c['w'].shape
        y = c['y']
        groups = Kout // 64 if C == 64 else Kout // 128           # (what ppy_conv1x1_expand_f32 accepts)
        if (R, S, c['stride']) != (1, 1, 1) or C not in (64, 128) or Kout % (64 if C == 64 else 128) or groups & (groups - 1) \
                or groups > 16 or c['ups'] or c['posb'] is not None or not has_f16(c) \
                or (two_streams and c.get('stream', 0) != op.get('stream', 0)) \
                or (y.buf, y.coff, y.C) != (x.buf, x.coff, x.C) or x.H % 2 or x.W % 2 or x.H * x.W < 32:
            continue
        c['pool'] = op['y']
        op['owner'] = c
        n += 1
    return n


def link_maxpools(ops, op_io, pinned, has_f16):
    """Host logic of HipExecutor._link_maxpools, device-free (tests/test_plan_host_logic.py): a 'maxpool' op (MaxPool2d(3, 2, 1), the
    stem's) whose input is the WHOLE buffer written by one convolution that ppy_conv3x3_maxpool_f32 accepts (3x3 / stride 1 / pad 1,
    C = 32 -> K = 64, no shortcut / upsampling / position bias; has_f16(op): f16x2 operands at hand) and read by nothing else:
    conv['mpool'] = the pooled slice, maxpool['owner'] = the convolution, whose launch then writes ONLY the pooled tensor.
    Returns the number of links."""
    n = 0
    for i, op in enumerate(ops):
        if op['op'] != 'maxpool':
            continue
        x = op['x']
        prods = [o for o in ops if x.buf in op_io(o)[1]]
        readers = [o for o in ops if o is not op and x.buf in op_io(o)[0]]
        if len(prods) != 1 or prods[0]['op'] != 'conv' or readers or x.buf in pinned:
            continue
        c = prods[0]
        Kout, R, S, C = c['w'].shape
        y = c['y']
        if (R, S, c['stride'], c['pad'], C, Kout) != (3, 3, 1, 1, 32, 64) or c['ups'] or c['posb'] is not None or c['res'] is not None \
                or c.get('pool') is not None or not has_f16(c) or c.get('stream', 0) != op.get('stream', 0) \
                or (y.buf, y.coff, y.C) != (x.buf, x.coff, x.C) or x.coff != 0:
            continue
        c['mpool'] = op['y']
        op['owner'] = c
        n += 1
    return n


def split_pairs(ops, op_io, buffers, pinned, has_f16, only_3x3=False):
    """Host logic of HipExecutor._link_splits, device-free (tests/test_plan_host_logic.py): [(producer, [consumers])] between which
    a tensor may travel PRE-SPLIT, whatever tiles they run on -- a buffer written by ONE convolution (the whole
    buffer, no shortcut term, no upsampled store, no pooled twin, a multiple of 32 channels) and read ONLY by convolutions, as
    their input: a bottleneck's conv1 -> conv2, the head's 1x1 -> 3x3 -> 1x1 chains, and a route with its two readers (the tip
    3x3 and the 1x1 in front of the upsampling).  pinned: buffers that something outside the convolution chain reads (feature
    maps, head outputs).  has_f16(op): f16x2 operands at hand.  only_3x3: A/B switch -- 1x1 consumers gain less (they split every
    activation once per wave column, a 3x3 nine times) but they gain: R50vd-608 bs 8 +0.9 % on top of the 3x3 links."""
    readers, writers = {}, {}
    for op in ops:
        ins, outs = op_io(op)
        for b in ins:
            readers.setdefault(b, []).append(op)
        for b in outs:
            writers.setdefault(b, []).append(op)
    pairs = []
    for b, ws_ in writers.items():
        if len(ws_) != 1 or b in pinned:
            continue
        pr = ws_[0]
        ld = buffers[b][3]
        y = pr.get('y')
        if pr['op'] != 'conv' or not has_f16(pr) or pr['ups'] or pr['res'] is not None or pr.get('pool') is not None \
                or y.buf != b or y.coff != 0 or y.C != ld or ld % 32:
            continue
        cons = readers.get(b, [])
        ok = bool(cons)
        for c in cons:
            x = c.get('x')
            if c['op'] != 'conv' or not has_f16(c) or x is None or x.buf != b or x.coff != 0 or x.C != ld \
                    or (c['res'] is not None and c['res'].buf == b) or (only_3x3 and c['w'].shape[1] != 3):
                ok = False
        if ok and len(set(id(c) for c in cons)) == len(cons):
            pairs.append((pr, cons))
    return pairs


def b2b_pairs(ops, op_io, buffers, pinned, has_f16):
    """Host logic of HipExecutor._mark_b2b, device-free: [(conv A, conv B)] that ppy_conv3x3_conv1x1_f32 can run as ONE launch --
    conv2 -> conv3 of an identity bottleneck (reference model/resnet_vd.py:81-87): A = 3x3 / stride 1 / pad 1, 64 -> 64, ReLU, no
    shortcut / position bias / upsampling, writing a whole buffer that ONLY B reads; B = 1x1 / stride 1, 64 -> 256, ReLU, with a
    shortcut, no position bias / upsampling."""
    readers, writers = {}, {}
    for op in ops:
        ins, outs = op_io(op)
        for b in ins:
            readers.setdefault(b, []).append(op)
        for b in outs:
            writers.setdefault(b, []).append(op)
    out = []
    for a in ops:
        if a['op'] != 'conv' or not has_f16(a):
            continue
        Ka, R, S, Ca = a['w'].shape
        y = a['y']
        if (R, S, a['stride'], a['pad'], Ca, Ka, a['act']) != (3, 3, 1, 1, 64, 64, 'relu') or a['res'] is not None or a['posb'] is not None \
                or a['ups'] or a.get('pool') is not None or a.get('mpool') is not None or y.buf in pinned \
                or y.coff != 0 or y.C != buffers[y.buf][3] or len(writers.get(y.buf, [])) != 1:
            continue
        rd = readers.get(y.buf, [])
        if len(rd) != 1 or rd[0]['op'] != 'conv':
            continue
        b = rd[0]
        Kb, Rb, Sb, Cb = b['w'].shape
        x = b['x']
        if (Rb, Sb, b['stride'], b['pad'], Cb, Kb, b['act']) != (1, 1, 1, 0, 64, 256, 'relu') or b['res'] is None or b['posb'] is not None \
                or b['ups'] or not has_f16(b) or (x.buf, x.coff, x.C) != (y.buf, 0, 64) or b['res'].buf == y.buf \
                or b.get('stream', 0) != a.get('stream', 0):
            continue
        out.append((a, b))
    return out


# =========================================================================================
class HipExecutor(object):
    """Binds a Plan to device buffers and replays it through libppyolo_hip.so."""

    def __init__(self, plan, device, use_graph=True, multi_stream=None, share=None):
        if torch.device(device).type != 'cuda':
            raise PPYoloHipError('the HIP executor needs a ROCm device (got %s); there is no CPU path' % device)
        self.plan = plan
        self.device = torch.device(device)
        self.use_graph = use_graph
        self.graph = None
        self._graph_stream = None
        p = plan
        self.bufs = None          # allocated after the pooling links are known (_alloc_buffers)
        self.x_in = torch.zeros((p.N, 3, p.H, p.W), dtype=torch.float32, device=self.device)
        self.im_size = torch.zeros((p.N, 2), dtype=torch.float32, device=self.device)
        d = p.decode
        if d is not None:
            M, C = d['M_total'], d['num_classes']
            self.boxes = torch.zeros((p.N, M, 4), dtype=torch.float32, device=self.device)
            cap = M * C
            self.cand_key = torch.zeros((p.N, cap), dtype=torch.int32, device=self.device)
            self.cand_idx = torch.zeros((p.N, cap), dtype=torch.int32, device=self.device)
            self.cand_count = torch.zeros((p.N,), dtype=torch.int32, device=self.device)
            kk = d['nms']['keep_top_k']
            self.out_dets = torch.zeros((p.N, kk, 6), dtype=torch.float32, device=self.device)
            self.out_count = torch.zeros((p.N,), dtype=torch.int32, device=self.device)
            self.out_keep = torch.zeros((p.N, kk), dtype=torch.int32, device=self.device)
            self.nms_ws = K.matrix_nms_workspace(p.N, self.device)
        self.math = math_mode()
        placeholders = any(op.get('w') is not None and op['w'].is_meta for op in p.setup_ops + p.ops)
        if share is not None and share.math == self.math and len(share.plan.ops) == len(p.ops) \
                and len(share.plan.setup_ops) == len(p.setup_ops):
            # weights are read-only and do not depend on the input shape: one copy in HBM serves every executor of the
            # model -- further lanes (runtime.InFlight), other input shapes, or the tensors of a native blob
            for mine, theirs in ((p.setup_ops, share.plan.setup_ops), (p.ops, share.plan.ops)):
                for op, src in zip(mine, theirs):
                    assert (op.get('w') is None) == (src.get('w') is None) and (op.get('w') is None or
                                                                              tuple(op['w'].shape) == tuple(src['w'].shape))
                    for k in ('w', 'scale', 'shift', 'w3', 'wf16'):
                        if src.get(k) is not None:
                            op[k] = src[k]
        else:
            if placeholders:
                raise PPYoloHipError('skeleton plan without a matching weight owner (math mode %s; this plan has %d + %d ops, the owner %s) -- '
                                     'a native blob written under another PPYOLO_HIP_FOLD_SHORTCUT / PPYOLO_HIP_MATH must be rebuilt'
                                     % (self.math, len(p.setup_ops), len(p.ops),
                                        'none' if share is None else '%d + %d, math %s' % (len(share.plan.setup_ops), len(share.plan.ops), share.math)))
            self._to_device(p.setup_ops)
            self._to_device(p.ops)
            if self.math in ('bf16x3', 'f16x2'):
                with torch.cuda.device(self.device):
                    for op in p.ops:        # (setup ops -- the CoordConv bias maps -- stay on the exact-fp32 kernel)
                        if op['op'] in ('conv', 'dcn'):
                            op['w3'] = K.split_weights_bf16x3(op['w'])
                            if self.math == 'f16x2':
                                op['wf16'] = K.split_weights_f16x2(op['w'], op['scale'])
        self._assign_amax()
        self._want_streams = os.environ.get('PPYOLO_HIP_STREAMS', '1') == '2' if multi_stream is None else bool(multi_stream)
        self._link_pools()
        self._link_maxpools()
        self._alloc_buffers()
        tab = tuned_table(self.math)
        tab_x3 = tuned_table('bf16x3') if self.math == 'f16x2' else {}
        self._mark_split_candidates()
        for op in p.ops:
            if op['op'] in ('conv', 'dcn') and op['cfg'] < 0:
                ent = tab.get(tune_key(op)) or tab.get(tune_key(op, False)) or tab_x3.get(tune_key(op, False))      # (a layer without ':f' behaves as in bf16x3 mode)
                if not ent and op.get('pool') is not None:                   # no entry for the pooled form: the plain shape's
                    k0 = tune_key(dict(op, pool=None), False)
                    ent = tab.get(k0) or tab_x3.get(k0)
                if ent:
                    op['cfg'], op['splitk'] = ent[:2]
        if os.environ.get('PPYOLO_HIP_HEAD_TAIL_FP32', '0') == '1':
            # experiment (round-5 review, item 7): the last two convolutions of every head level -- the tip 3x3 and the output 1x1 -- on
            # the exact-fp32 MFMA kernels, everything in front of them as usual.  Measured: does not move the count of boxes beyond
            # 1e-3 px at R50vd-608; off by default.
            outs = {a.buf for a in p.head_outs}
            tail = [op for op in p.ops if op['op'] == 'conv' and op['y'].buf in outs]
            tips = [op for op in p.ops if op['op'] == 'conv' and any(op['y'].buf == t['x'].buf for t in tail)]
            for op in tail + tips:
                x = op['x']
                Kout, R, S, C = op['w'].shape
                op['cfg'], op['splitk'] = K.conv2d_pick(x.N, x.H, x.W, C, Kout, R, S, op['stride'], op['pad'])
        self.ws = None
        self.ws_side = None
        self._size_workspace()
        # Independent branches (projection shortcut, head tip / output convs) on a second stream: OPT-IN
        # (PPYOLO_HIP_STREAMS=2).  A forked graph gains 3 % for one batch at a time (R50-608 bs8 1481 -> 1526 img/s; small
        # batches lose on the fork/join), but keeping two batches in flight on two single-branch graphs (runtime.InFlight)
        # gains 24 % where two forked ones gain 3 %, and a forked hipGraph replayed under another stream than its first
        # has crashed the ROCm 7.2 runtime -- so the default is one branch, and a forked graph refuses a stream change.
        want = self._want_streams
        self.multi_stream = want and any(op.get('stream', 0) for op in p.ops)
        self.side_stream = torch.cuda.Stream(device=self.device) if self.multi_stream else None
        self._build_sync_plan()
        with torch.cuda.device(self.device):
            for op in p.setup_ops:
                self._run_op(op)
            torch.cuda.synchronize()
            for op in p.ops:      # CoordConv bias maps in the scaled-weight domain of the f16x2 kernels (exact: powers of two)
                if op.get('wf16') is not None and op.get('posb') is not None:
                    s_w = torch.where(op['scale'] != 0, op['scale'] / op['wf16'][1], torch.ones_like(op['scale']))
                    op['posb_f16'] = (self.bufs[op['posb'].buf] * s_w).contiguous()
            self._link_splits()

    def _alloc_buffers(self):
        """One tensor per plan buffer that some op still reads or writes once the pooling links are made: the full-resolution stem
        tensor between conv1_3 and the max pool written from its epilogue (189 MB at 608 x 608, batch 8, per executor and lane) is
        referenced by nothing and stays unallocated (None)."""
        p = self.plan
        used = set(p.consts)
        for op in p.setup_ops + p.ops:
            ins, outs = self._op_io(op)
            used.update(ins)
            used.update(outs)
            if op.get('posb') is not None:
                used.add(op['posb'].buf)
        used.update(a.buf for a in list(p.feats) + list(p.head_outs))
        # (also behind a fused pair: the plan may fall back to two launches -- but NOT the full-resolution tensor in front of a
        # linked max pool: its launch writes the pooled tensor only, and 189 MB per executor and lane stay unallocated)
        used.update(op['y'].buf for op in p.ops if op['op'] == 'conv' and op.get('mpool') is None)
        self.bufs = []
        for i, (N, H, W, ld) in enumerate(p.buffers):
            if i in p.consts:
                self.bufs.append(p.consts[i].to(self.device).contiguous())
            elif i in used:
                self.bufs.append(torch.empty((N, H, W, ld), dtype=torch.float32, device=self.device))
            else:
                self.bufs.append(None)

    def _assign_amax(self):
        """Tracked per-image tensor maxima for the f16x2 kernels: every conv / DCN launch merges max|y| into the slots of its
        output buffer; pooled tensors inherit the slots of their input (max- and average-pooling never exceed it; SPP
        writes into its own input buffer; the DCN columns are bounded by the DCN input); the stem kernel tracks its
        output as well."""
        nblocks = assign_amax(self.plan.ops)
        self._amax_block = self.plan.N * K.AMAX_FLOATS_PER_IMAGE
        self.amax = torch.zeros(max(1, nblocks) * self._amax_block, dtype=torch.float32, device=self.device)

    def _link_pools(self):
        """The vd shortcut's AvgPool2d(2, 2) (reference model/resnet_vd.py:29-33) belongs to the launch that produces its
        input where that is a 1x1 convolution the streaming kernel (csrc/conv_stream.hip) can run: the producer then
        writes the 2x2 average from its own epilogue (cfg = a streaming id) or, on any other tile, the pooling launch
        follows it immediately; the 'avgpool' op of the plan is skipped either way."""
        if self.math != 'f16x2' or os.environ.get('PPYOLO_HIP_POOL_FOLD', '1') != '1':
            return
        link_pools(self.plan.ops, self._op_io, lambda c: c.get('wf16') is not None and c.get('amax_in_id') is not None,
                   self._want_streams)

    def _link_maxpools(self):
        """The stem's MaxPool2d(3, 2, 1) (reference model/resnet_vd.py:103, 136) belongs to the launch of the convolution in front of
        it (csrc/conv_patch.hip, MPOOL): the 304 x 304 x 64 tensor between them is neither written nor read.
        PPYOLO_HIP_MAXPOOL_FOLD=0: two launches."""
        if self.math != 'f16x2' or os.environ.get('PPYOLO_HIP_MAXPOOL_FOLD', '1') != '1':
            return
        pinned = {a.buf for a in list(self.plan.head_outs) + list(self.plan.feats)}
        link_maxpools(self.plan.ops, self._op_io, pinned, lambda c: c.get('wf16') is not None and c.get('amax_in_id') is not None)

    def _split_capable(self, cfg, consumer):
        """Tile configurations that read (consumer) / write pre-split tensors: the f16x2 tiles of csrc/conv_x3.hip without slab
        reuse and the specialised-wave tiles of csrc/conv_ws.hip (as consumers: those whose producer waves do not split)."""
        f0 = NUM_FP32_CFGS + NUM_X3_CFGS
        if f0 <= cfg < f0 + 27 or f0 + 45 <= cfg < f0 + 54:          # 9 tiles x {2, 3, 4} stages; the 96 / 192-row tiles
            return True
        if cfg >= K.small_first_cfg():          # the wave-private tiles for small outputs (csrc/conv_small.hip, round 6): both sides
            return True
        w0 = K.ws_first_cfg()
        return cfg - w0 in ((0, 1, 2, 3, 7, 8, 9, 10, 11, 12, 13, 14, 15) if consumer else tuple(range(16)))      # (9-15: the k-parity tiles, round 6)

    @staticmethod
    def _split_leaves_launch(op):
        """Does this op's split-K go through partial sums in memory (a second launch combines them)?  Such a launch neither reads nor
        writes pre-split tensors.  The small-output tiles (csrc/conv_small.hip) add their k-parts inside the workgroup."""
        return op.get('splitk', 0) > 1 and not (op['op'] == 'conv' and op.get('cfg', -1) >= K.small_first_cfg())

    def _split_pairs(self):
        """[(producer, [consumers])] that qualify STRUCTURALLY for a pre-split tensor between them (split_pairs below)."""
        if self.math != 'f16x2' or os.environ.get('PPYOLO_HIP_PRESPLIT', '1') != '1':
            return []
        pinned = {a.buf for a in list(self.plan.head_outs) + list(self.plan.feats)}      # read from outside the conv chain
        return split_pairs(self.plan.ops, self._op_io, self.plan.buffers, pinned,
                           lambda c: c.get('wf16') is not None and c.get('amax_in_id') is not None,
                           os.environ.get('PPYOLO_HIP_PRESPLIT_3X3_ONLY', '0') == '1')

    def _mark_split_candidates(self):
        for op in self.plan.ops:
            op.pop('gp_in', None)
        for _, cons in self._split_pairs():
            for c in cons:
                c['gp_in'] = True

    def _unlink_splits(self):
        for op in self.plan.ops:
            op.pop('x_split', None)
            op.pop('y_split', None)

    def _link_splits(self, _retry=False):
        """"Global pre-split" (DESIGN.md 4.1g): where a convolution's output buffer is read by exactly ONE op, a convolution on
        an f16x2 tile kernel, the producer stores it as that consumer's finished MFMA operands (two fp16 terms of y * s_image,
        same bytes per pixel) and the consumer's main loop carries no scale / split work.  s_image comes from a static bound
        of |y| -- per output channel |scale| * sum|w| times the input's tracked maximum, plus |shift| and the CoordConv bias --
        so the producer needs no second pass.  Bottleneck conv1 -> conv2 (3x3) and the head's 1x1 -> 3x3 pairs qualify."""
        if not _retry:
            self._b2b_rejected = set()      # (op indices of fused pairs that fell back to two launches in this derivation)
        self._unlink_splits()
        self._mark_b2b()
        n = 0
        for pr, cons in self._split_pairs():
            if pr.get('b2b') is not None or pr.get('b2b_of') is not None:
                continue          # (the fused pair's output is plain fp32: it is a shortcut as well)
            if self._split_leaves_launch(pr) or not self._split_capable(pr['cfg'], False) \
                    or any(c.get('b2b') is None and (self._split_leaves_launch(c) or not self._split_capable(c['cfg'], True)) for c in cons):
                continue          # (every reader must take the tensor in that form, or none does)
            w, sc, sh = pr['w'], pr['scale'], pr['shift']
            l1 = w.abs().double().sum(dim=(1, 2, 3))
            mul = float((sc.abs().double() * l1).max())
            add = sh.abs().double()
            if pr['posb'] is not None:
                pb = self.bufs[pr['posb'].buf].abs().double().reshape(-1, w.shape[0]).amax(dim=0)
                add = add + pb * sc.abs().double()
            add = float(add.max())
            ps = torch.ones(self.plan.N, dtype=torch.float32, device=self.device)
            pr['y_split'] = (ps, mul * (1.0 + 2.0 ** -8), add * (1.0 + 2.0 ** -8) + 1e-30)
            for c in cons:
                c['x_split'] = ps
                n += 1
        # a fused pair needs its input pre-split (csrc/conv_b2b.hip reads finished operands): without the link it is two launches again
        undone = False
        for i, op in enumerate(self.plan.ops):
            if op.get('b2b') is not None and op.get('x_split') is None:
                self._b2b_rejected.add(i)
                undone = True
        if undone:      # (the links were derived with that pair fused: its stand-alone form may allow others; the other pairs stay fused)
            return self._link_splits(_retry=True)
        return n

    def _unlink_b2b(self):
        for op in self.plan.ops:
            if op.get('b2b') is not None:
                op['b2b'].pop('b2b_of', None)
                op.pop('b2b', None)

    def _mark_b2b(self):
        """conv2 -> conv3 of an identity bottleneck as ONE launch (round 5, csrc/conv_b2b.hip; PPYOLO_HIP_B2B=0: two launches): the
        64-channel tensor between them is neither written nor read.  The static bound of the intermediate (per-image operand scale
        of the second contraction) is derived as for a pre-split link."""
        self._unlink_b2b()
        if self.math != 'f16x2' or os.environ.get('PPYOLO_HIP_B2B', '1') != '1':
            return 0
        pinned = {a.buf for a in list(self.plan.head_outs) + list(self.plan.feats)}
        pairs = b2b_pairs(self.plan.ops, self._op_io, self.plan.buffers, pinned,
                          lambda c: c.get('wf16') is not None and c.get('amax_in_id') is not None)
        # pairs that did not get their pre-split input in an earlier pass of _link_splits stay two launches (round-5 advisor: only
        # those, not every pair of the plan); the set is cleared whenever the links are rebuilt from scratch (a new tile table)
        rejected = {id(self.plan.ops[i]) for i in getattr(self, '_b2b_rejected', ())}
        pairs = [(a, b) for a, b in pairs if id(a) not in rejected]
        for a, b in pairs:
            if a.get('t_bound') is None:
                w, sc, sh = a['w'], a['scale'], a['shift']
                l1 = w.abs().double().sum(dim=(1, 2, 3))
                a['t_bound'] = (float((sc.abs().double() * l1).max()) * (1.0 + 2.0 ** -8), float(sh.abs().double().max()) * (1.0 + 2.0 ** -8) + 1e-30)
            a['b2b'] = b
            b['b2b_of'] = a
        return len(pairs)

    def presplit_headroom(self):
        """Diagnostic (host sync; after a run): for every pre-split link and image, (key of the producer, log2 of the SCALED
        maximum s_image * max|y|).  The scale comes from a static bound of |y| (above), so the scaled maximum sits below 2^14 by
        however pessimistic that bound is for the data at hand; both fp16 terms stay normal numbers for values down to
        2^-(24 - (14 - log2)) of the maximum.  tests/test_gpu_model.py holds the R50vd plan to >= 2^4 on its synthetic inputs."""
        out = []
        for op in self.plan.ops:
            ys = op.get('y_split')
            if ys is None or op.get('amax_out_id') is None:
                continue
            mx = self._amax(op['amax_out_id']).view(self.plan.N, -1).amax(dim=1)
            scaled = (mx * ys[0]).cpu()
            out.append((tune_key(op), [float(torch.log2(v)) if v > 0 else float('-inf') for v in scaled]))
        return out

    def _amax(self, idx):
        return None if idx is None else self.amax[idx * self._amax_block:(idx + 1) * self._amax_block]

    # ---- helpers ---------------------------------------------------------------------------
    def _to_device(self, oplist):
        for op in oplist:
            for k in ('w', 'scale', 'shift'):
                if k in op and op[k] is not None:
                    op[k] = op[k].to(self.device).contiguous()

    def view(self, a):
        return K.View(self.bufs[a.buf], a.coff, a.C)

    def _ws_need(self, op):
        if op['op'] == 'conv':
            x = op['x']
            Kout, R, S, C = op['w'].shape
            return K.conv2d_workspace_bytes(x.N, x.H, x.W, C, Kout, R, S, op['stride'], op['pad'], op['cfg'],
                                            op['splitk'])
        if op['op'] == 'dcn':
            x = op['x']
            return K.dcnv2_workspace_bytes(x.N, x.H, x.W, x.C, op['w'].shape[0], op['stride'], op['pad'], op['cfg'],
                                           op['splitk'])
        return 0

    def _size_workspace(self):
        need, need_side = 16, 16
        for op in self.plan.setup_ops + self.plan.ops:
            if op.get('stream', 0):
                need_side = max(need_side, self._ws_need(op))
            else:
                need = max(need, self._ws_need(op))
        if self.ws is None or self.ws.numel() * 4 < need:
            self.ws = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=self.device)
        if self.ws_side is None or self.ws_side.numel() * 4 < need_side:      # concurrent branch: own scratch
            self.ws_side = torch.empty(((need_side + 3) // 4,), dtype=torch.float32, device=self.device)

    @staticmethod
    def _op_io(op):
        """(input buffer ids, output buffer ids) of a plan op."""
        t = op['op']
        if t == 'conv' and op.get('b2b_of') is not None:         # computed inside the launch of the convolution in front of it (_mark_b2b)
            return [], []
        if t == 'conv' and op.get('b2b') is not None:
            b = op['b2b']
            return [op['x'].buf, b['res'].buf], [b['y'].buf] + ([b['pool'].buf] if b.get('pool') is not None else [])
        if t == 'conv':
            ins = [op['x'].buf] + ([op['res'].buf] if op['res'] is not None else [])
            if op.get('mpool') is not None:          # only the pooled tensor is written (_link_maxpools)
                return ins, [op['mpool'].buf]
            return ins, [op['y'].buf] + ([op['pool'].buf] if op.get('pool') is not None else [])
        if t == 'stem':
            return [], [op['y'].buf]
        if t in ('avgpool', 'maxpool') and op.get('owner') is not None:      # written by its producer's launch (_link_pools / _link_maxpools)
            return [], []
        if t in ('maxpool', 'avgpool'):
            return [op['x'].buf], [op['y'].buf]
        if t == 'spp':
            return [op['x'].buf], [op['y5'].buf]
        if t == 'dcn':
            return [op['x'].buf, op['om'].buf], [op['y'].buf]
        raise PPYoloHipError('unknown plan op %r' % t)

    def _build_sync_plan(self):
        """Cross-stream dependencies, derived once: an op waits for every earlier writer of a buffer
        it reads that ran on the other stream (buffers are never reused, concat buffers have
        several writers of disjoint slices -> wait for all of them)."""
        ops = self.plan.ops
        writers = {}
        self._waits = [[] for _ in ops]          # op index -> producer op indices on the other stream
        self._needs_event = set()
        for i, op in enumerate(ops):
            ins, outs = self._op_io(op)
            s = op.get('stream', 0) if self.multi_stream else 0
            for bid in ins:
                for j in writers.get(bid, []):
                    sj = ops[j].get('stream', 0) if self.multi_stream else 0
                    if sj != s and j not in self._waits[i]:
                        self._waits[i].append(j)
                        self._needs_event.add(j)
            for bid in outs:
                writers.setdefault(bid, []).append(i)
        # tail of the side stream must be joined before decode / the end of the step
        self._side_tail = max([i for i, op in enumerate(ops) if op.get('stream', 0)], default=None) \
            if self.multi_stream else None

    @staticmethod
    def _stream_first():
        """First tile-configuration id of the streaming 1x1 kernel."""
        return K.stream_first_cfg()

    def _run_op(self, op, ws=None):
        t = op['op']
        ws = self.ws if ws is None else ws
        if t == 'conv' and op.get('b2b_of') is not None:
            return                  # (computed by the launch of the convolution in front of it)
        if op.get('amax_in2_id') is not None and (t != 'conv' or op.get('b2b') is not None or op.get('mpool') is not None or (
                op.get('pool') is not None and self._stream_first() <= op['cfg'] < self._stream_first() + 2)):
            raise PPYoloHipError('plan op %s reads a buffer with two tracked-maximum blocks through an entry point that takes one' % tune_key(op))
        if t == 'conv' and op.get('b2b') is not None:
            b = op['b2b']
            K.conv3x3_conv1x1(self.view(op['x']), op['x_split'], self._amax(op['amax_in_id']), op['wf16'], op['shift'], b['wf16'], b['shift'],
                              self.view(b['res']), self.view(b['y']), op['t_bound'][0], op['t_bound'][1], self._amax(b.get('amax_out_id')),
                              None if b.get('pool') is None else self.view(b['pool']))
            return
        if t == 'conv' and op.get('pool') is not None and self._stream_first() <= op['cfg'] < self._stream_first() + 2:
            K.conv1x1_expand(self.view(op['x']), op['wf16'], op['shift'], self.view(op['y']), op['act'],
                             None if op['res'] is None else self.view(op['res']), self.view(op['pool']),
                             op['cfg'] - self._stream_first(), self._amax(op.get('amax_in_id')), self._amax(op.get('amax_out_id')))
        elif t == 'conv' and op.get('mpool') is not None:
            K.conv3x3_maxpool(self.view(op['x']), op['wf16'], op['shift'], self.view(op['mpool']), op['act'],
                              self._amax(op.get('amax_in_id')), self._amax(op.get('amax_out_id')))
        elif t == 'conv':
            posb = op['posb']
            K.conv2d_bn_act(self.view(op['x']), op['w'], op['scale'], op['shift'], self.view(op['y']), op['stride'],
                            op['pad'], op['act'], None if op['res'] is None else self.view(op['res']),
                            None if posb is None else self.bufs[posb.buf], op['ups'], op['cfg'], op['splitk'],
                            ws, op.get('w3'), op.get('wf16'), self._amax(op.get('amax_in_id')),
                            self._amax(op.get('amax_out_id')), op.get('posb_f16'), op.get('x_split'), op.get('y_split'),
                            self._amax(op.get('amax_in2_id')))
            if op.get('pool') is not None:
                K.avgpool2x2(self.view(op['y']), self.view(op['pool']))
        elif t == 'stem':
            # (16-bit modes: the stem on the bf16 MFMA too; PPYOLO_HIP_STEM_MFMA=0 or the exact-fp32 mode: the fp32 fma chain)
            K.stem_conv(self.x_in, op['w'], op['scale'], op['shift'], self.view(op['y']), op['act'],
                        self._amax(op.get('amax_out_id')),
                        mfma=self.math != 'fp32' and op['w'].shape[0] == 32 and os.environ.get('PPYOLO_HIP_STEM_MFMA', '1') == '1')
        elif t == 'maxpool':
            if op.get('owner') is None:          # (else: written by the producer's launch, _link_maxpools)
                K.maxpool3x3s2(self.view(op['x']), self.view(op['y']))
        elif t == 'avgpool':
            if op.get('owner') is None:          # (else: written by the producer's launch, _link_pools)
                K.avgpool2x2(self.view(op['x']), self.view(op['y']))
        elif t == 'spp':
            K.spp(self.view(op['x']), self.view(op['y5']), self.view(op['y9']), self.view(op['y13']))
        elif t == 'dcn':
            K.dcnv2(self.view(op['x']), op['w'], op['scale'], op['shift'], self.view(op['om']), self.view(op['y']),
                    op['stride'], op['pad'], op['act'], ws, op['cfg'], op['splitk'], op.get('w3'), op.get('wf16'),
                    self._amax(op.get('amax_in_id')), self._amax(op.get('amax_out_id')))
        else:
            raise PPYoloHipError('unknown plan op %r' % t)

    def _run_decode(self):
        d = self.plan.decode
        self.cand_count.zero_()
        K.yolo_decode_levels([self.view(a) for a in self.plan.head_outs], [lvl['anchors'] for lvl in d['levels']],
                             [lvl['downsample'] for lvl in d['levels']], d['num_classes'], d['scale_x_y'], d['iou_aware'],
                             d['iou_aware_factor'], d['clip_bbox'], self.im_size, self.boxes,
                             d['nms']['score_threshold'], self.cand_key, self.cand_idx, self.cand_count)
        n = d['nms']
        K.matrix_nms(self.boxes, d['num_classes'], self.cand_key, self.cand_idx, self.cand_count,
                     n['post_threshold'], n['nms_top_k'], n['keep_top_k'], n['use_gaussian'], n['gaussian_sigma'],
                     self.out_dets, self.out_count, self.out_keep, self.nms_ws)

    def _launch_all(self):
        self.amax.zero_()        # tracked tensor maxima restart with every step (one memset node in the graph)
        if not self.multi_stream:
            for op in self.plan.ops:
                self._run_op(op)
        else:
            main = torch.cuda.current_stream()
            side = self.side_stream
            events = {}
            for i, op in enumerate(self.plan.ops):
                on_side = bool(op.get('stream', 0))
                st = side if on_side else main
                for j in self._waits[i]:
                    st.wait_event(events[j])
                if on_side:
                    with torch.cuda.stream(side):
                        self._run_op(op, self.ws_side)
                else:
                    self._run_op(op)
                if i in self._needs_event or i == self._side_tail:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    events[i] = ev
            if self._side_tail is not None:
                main.wait_event(events[self._side_tail])      # join
        if self.plan.decode is not None:
            self._run_decode()

    # ---- public ----------------------------------------------------------------------------
    def set_inputs(self, x, im_size=None):
        if tuple(x.shape) != tuple(self.x_in.shape):
            raise PPYoloHipError('plan was built for input %s, got %s' % (tuple(self.x_in.shape), tuple(x.shape)))
        self.x_in.copy_(x)
        if im_size is not None:
            self.im_size.copy_(im_size.to(torch.float32))

    def run(self):
        """Enqueue one forward on the current stream (device-resident in -> device-resident out)."""
        with torch.cuda.device(self.device):
            if not self.use_graph:
                self._launch_all()
                return
            cur = torch.cuda.current_stream()
            if self.graph is not None and self.multi_stream and cur != self._graph_stream:
                raise PPYoloHipError('PPYOLO_HIP_STREAMS=2: this forked hipGraph was first replayed on another stream; '
                                     'replaying it here has crashed the ROCm runtime -- stay on one stream, or drop the option')
            if self.graph is None:
                self._launch_all()                      # warm-up: module load, func attributes
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch_all()
                self.graph = g
                self._graph_stream = cur
            self.graph.replay()

    def invalidate_graph(self):
        self.graph = None
        self._graph_stream = None

    # ---- autotune --------------------------------------------------------------------------
    def autotune(self, iters=3, verbose=False, kinds=('conv', 'dcn'), match=None, only_cfgs=None):
        """Per-layer (tile config, split-K) search measured on the device: 'measure, don't
        guess'.  Results are written into the plan ops; the graph is re-captured lazily.  only_cfgs (convolutions): measure just these
        tile ids (new candidate kernels) against the layer's current entry instead of the whole table of configurations."""
        from ._lib import lib
        ncfg_conv = {'fp32': NUM_FP32_CFGS, 'bf16x3': NUM_FP32_CFGS + NUM_X3_CFGS}.get(self.math, lib().ppy_conv2d_num_configs())
        cfgs_dcn = K.dcnv2_configs(self.math)      # schemes up to this mode's (+ the eight-wave f16x2 tiles)
        splits = (1, 2, 3, 4, 6, 8, 9, 12, 16)
        report = []
        self._unlink_splits()          # (layers are measured on plain fp32 tensors; the links are re-derived from the new choices)
        self._unlink_b2b()
        with torch.cuda.device(self.device):
            big = 0
            for op in self.plan.ops:
                if op['op'] in ('conv', 'dcn'):
                    for c in (cfgs_dcn if op['op'] == 'dcn' else range(ncfg_conv)):
                        for s in splits:
                            o = dict(op, cfg=c, splitk=s)
                            big = max(big, self._ws_need(o))
            if self.ws.numel() * 4 < big:
                self.ws = torch.empty(((big + 3) // 4,), dtype=torch.float32, device=self.device)
            for op in self.plan.ops:
                if op['op'] not in kinds or (match and not all(m in tune_key(op) for m in match)) or op.get('mpool') is not None:
                    continue          # (a convolution that owns the stem's max pool has one kernel: nothing to choose)
                Kout = op['w'].shape[0]
                Kred = op['w'].shape[1] * op['w'].shape[2] * op['w'].shape[3]
                chunks = Kred // 32
                base_cfg, base_split = op['cfg'], op['splitk']

                gp_scales = torch.ones(self.plan.N, dtype=torch.float32, device=self.device) if op.get('gp_in') else None

                def measure(c, s, n):
                    op['cfg'], op['splitk'] = c, s
                    # a layer whose input will arrive pre-split is measured in that form on the tiles that can read it (the
                    # bytes it reads are whatever the buffer holds: the timing does not depend on the values)
                    op.pop('x_split', None)
                    if gp_scales is not None and (s <= 1 or c >= K.small_first_cfg()) and self._split_capable(c, True):
                        op['x_split'] = gp_scales
                    try:
                        self._run_op(op)
                    except PPYoloHipError:
                        return None
                    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    st.record()
                    for _ in range(n):
                        self._run_op(op)
                    en.record()
                    en.synchronize()
                    ms = st.elapsed_time(en) / n
                    if ms < 0.04:
                        # a launch this short is hidden behind the ~10-20 us the host needs per ctypes launch: time it
                        # the way it will run, as nodes of a captured graph
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            for _ in range(16):
                                self._run_op(op)
                        g.replay()
                        st.record()
                        for _ in range(max(1, n // 4)):
                            g.replay()
                        en.record()
                        en.synchronize()
                        ms = st.elapsed_time(en) / (16 * max(1, n // 4))
                        g.reset()
                        del g
                    return ms

                cands = []
                pool = cfgs_dcn if op['op'] == 'dcn' else (range(ncfg_conv) if only_cfgs is None else sorted(set(only_cfgs)))
                if only_cfgs is not None and op['op'] == 'conv' and base_cfg >= 0:
                    ms = measure(base_cfg, base_split, iters)             # the current entry defends its place
                    if ms is not None:
                        cands.append((ms, base_cfg, base_split))
                for c in pool:
                    for s in splits:
                        if s > 1 and chunks // s < 4:
                            continue
                        if only_cfgs is not None and op['op'] == 'conv' and (c, s) == (base_cfg, base_split):
                            continue
                        ms = measure(c, s, iters)
                        if ms is not None:
                            cands.append((ms, c, s))
                # second look at the front-runners: short kernels are noisy at `iters` repetitions
                best = None
                front = []
                for ms, c, s in sorted(cands)[:6]:
                    again = min(measure(c, s, 4 * iters), measure(c, s, 4 * iters))
                    front.append((again, c, s))
                    if best is None or again < best[0]:
                        best = (again, c, s)
                op['_front'] = sorted(front)           # kept for co_tune()
                op['_cands'] = sorted(cands)[:24]      # (single look; insitu_tune(topk > 6) re-ranks more of them)
                op.pop('x_split', None)
                op.pop('_incumbent', None)
                if only_cfgs is not None and op['op'] == 'conv' and base_cfg >= 0:
                    # new candidate kernels against a measured table: the incumbent keeps its place unless it loses by >= 3 % (launches of
                    # 10-20 us repeat to +-3 %; without the margin every re-measurement reshuffles the small layers)
                    op['_incumbent'] = (base_cfg, base_split)
                    inc = [t for t in front if (t[1], t[2]) == (base_cfg, base_split)]
                    if not inc:
                        ms_inc = measure(base_cfg, base_split, 4 * iters)
                        inc = [(ms_inc, base_cfg, base_split)] if ms_inc is not None else []
                        op['_front'] = sorted(front + inc)
                    if inc and best is not None and (best[1], best[2]) != (base_cfg, base_split) and best[0] > 0.97 * inc[0][0]:
                        best = inc[0]
                    op.pop('x_split', None)
                if best is None:
                    op['cfg'], op['splitk'] = base_cfg, base_split
                    continue
                op['cfg'], op['splitk'] = best[1], best[2]
                tuned_table(self.math)[tune_key(op)] = [best[1], best[2], round(best[0], 4)]
                report.append((tune_key(op), best))
                if verbose:
                    print('autotune %s w=%s H=%d -> cfg %d split %d  %.3f ms' % (op['op'], tuple(op['w'].shape),
                                                                               op['x'].H, best[1], best[2], best[0]))
        self._size_workspace()
        self._link_splits()
        self.graph = None
        return report

    def co_tune(self, other, topk=4, reps=3, verbose=False):
        """Second tuning stage for lanes that run beside each other (runtime.InFlight): among the `topk` front-runners
        autotune() measured for a layer, take the one with the shortest MAKESPAN of {this layer repeated for as long as
        one whole step takes} on one stream and {one whole step of `other`, an executor of the same plan} on another.
        The fastest kernel alone is not always the best neighbour: a tile shape that leaves CUs, LDS or power to the
        other lane can finish the pair sooner.  Measured, R50-608 bs8, two lanes: +0.6 %."""
        import time
        self._unlink_splits()
        self._unlink_b2b()
        with torch.cuda.device(self.device):
            sa, sb = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(sb):
                other.run()
                other.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(sb):
                for _ in range(5):
                    other.run()
            torch.cuda.synchronize()
            t_step = (time.perf_counter() - t0) / 5
            changed = 0
            for op in self.plan.ops:
                front = op.get('_front')
                if op['op'] not in ('conv', 'dcn') or not front or len(front) < 2:
                    continue
                n = int(max(4, min(1500, round(t_step * 1e3 / max(front[0][0], 1e-3)))))
                scored = []
                for ms, c, s in front[:topk]:
                    op['cfg'], op['splitk'] = c, s
                    self._run_op(op)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(n):
                            self._run_op(op)
                    best = None
                    for _ in range(reps):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        with torch.cuda.stream(sb):
                            other.run()
                        with torch.cuda.stream(sa):
                            g.replay()
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    g.reset()
                    del g
                    scored.append((best, c, s, ms))
                scored.sort()
                win = scored[0]
                if (win[1], win[2]) != (front[0][1], front[0][2]):
                    changed += 1
                op['cfg'], op['splitk'] = win[1], win[2]
                tuned_table(self.math)[tune_key(op)] = [win[1], win[2], round(win[3], 4)]
                if verbose:
                    print('co_tune %s: %s -> cfg %d split %d' % (tune_key(op), ['%d/%d %.3f' % (c, s, 1e3 * b)
                                                                              for b, c, s, _ in scored], win[1], win[2]))
        self._size_workspace()
        self._link_splits()
        self.graph = None
        return changed

    def insitu_tune(self, topk=6, reps=5, verbose=False):
        """Third tuning stage (round 5): re-rank the `topk` front-runners autotune() found for a layer by their duration INSIDE a
        whole eager pass of the plan (HIP events around that one launch, best of `reps` passes) -- where the layer's weights and
        activations come from wherever the rest of the step left them, not from the caches its own previous repetition warmed
        (the stage-4 1x1 layers take 29 us back to back and 39-42 us in the step)."""
        self._unlink_splits()
        self._unlink_b2b()
        changed = 0
        with torch.cuda.device(self.device):
            self._size_workspace()
            ops = self.plan.ops
            for i, op in enumerate(ops):
                front = op.get('_front')
                if op['op'] not in ('conv', 'dcn') or not front or len(front) < 2:
                    continue
                pool = front if topk <= len(front) else front + [t for t in op.get('_cands', []) if (t[1], t[2]) not in {(c, s) for _, c, s in front}]
                scored = []
                for ms, c, s in pool[:topk]:
                    op['cfg'], op['splitk'] = c, s
                    best = None
                    for _ in range(reps):
                        self.amax.zero_()
                        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        for j, o in enumerate(ops):
                            if j == i:
                                st.record()
                                self._run_op(o)
                                en.record()
                            else:
                                self._run_op(o)
                        en.synchronize()
                        t = st.elapsed_time(en)
                        best = t if best is None else min(best, t)
                    scored.append((best, c, s, ms))
                scored.sort()
                win = scored[0]
                inc = [t for t in scored if (t[1], t[2]) == op.get('_incumbent')]
                if inc and (win[1], win[2]) != (inc[0][1], inc[0][2]) and win[0] > 0.97 * inc[0][0]:
                    win = inc[0]          # (autotune(only_cfgs=...): the table's entry stays unless a challenger is >= 3 % faster inside the step too)
                if (win[1], win[2]) != (front[0][1], front[0][2]):
                    changed += 1
                op['cfg'], op['splitk'] = win[1], win[2]
                tuned_table(self.math)[tune_key(op)] = [win[1], win[2], round(win[3], 4)]
                if verbose:
                    print('insitu_tune %s: %s -> cfg %d split %d' % (tune_key(op), ['%d/%d %.1f' % (c, s, 1e3 * b) for b, c, s, _ in scored], win[1], win[2]))
        self._size_workspace()
        self._link_splits()
        self.graph = None
        return changed

    def save_tuning(self, path):
        """Write the entries THIS executor measured (not the whole loaded table: merging files of several workloads
        would otherwise let one file's stale copies override another's fresh entries)."""
        tab = tuned_table(self.math)
        keys = {tune_key(op) for op in self.plan.ops if op['op'] in ('conv', 'dcn')}
        with open(path, 'w') as fh:
            json.dump({k: tab[k] for k in sorted(keys) if k in tab}, fh, indent=0, sort_keys=True)


def run_single(unit, x_nchw):
    """Run ONE Conv2dUnit through the HIP kernels on an NCHW tensor (layout conversion by torch
    is test / API glue, not part of the timed path)."""
    if not x_nchw.is_cuda:
        raise PPYoloHipError('HIP path needs a ROCm device tensor; there is no CPU fallback')
    N, C, H, W = x_nchw.shape
    b = Builder(N, H, W, x_nchw.device)
    xin = None
    if C == 3 and unit.stride == 2:
        y = unit.emit(b, None)
    else:
        xin = b.new_act(N, H, W, C)
        y = unit.emit(b, xin)
    ex = HipExecutor(b.plan, x_nchw.device, use_graph=False)
    if xin is None:
        ex.set_inputs(x_nchw.float())
    else:
        ex.bufs[xin.buf].copy_(x_nchw.float().permute(0, 2, 3, 1))
    ex.run()
    return ex.view(y).dense().permute(0, 3, 1, 2).contiguous()
