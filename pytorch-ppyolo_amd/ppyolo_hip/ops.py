"""Tensor-level wrappers over the C ABI (include/ppyolo_hip.h).

PyTorch-ROCm tensors are backing storage only: every function takes fp32 NHWC tensors
that live on a ROCm device, passes `data_ptr()`s and the current HIP stream to
libppyolo_hip.so and returns.  No function here computes anything with torch ops and
none has a CPU path -- a CPU tensor raises.

A "view" is (tensor, channel_offset, channels): channel slice [off, off+C) of an NHWC
buffer whose last dimension is the pixel stride `ld`.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT, check, lib


class View(object):
    """Channel slice [coff, coff+C) of an NHWC fp32 buffer [N,H,W,ld]."""
    __slots__ = ('t', 'coff', 'C')

    def __init__(self, t, coff=0, C=None):
        assert t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous()
        self.t = t
        self.coff = coff
        self.C = t.shape[3] - coff if C is None else C
        assert 0 <= coff and coff + self.C <= t.shape[3]

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def H(self):
        return self.t.shape[1]

    @property
    def W(self):
        return self.t.shape[2]

    @property
    def ld(self):
        return self.t.shape[3]

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * self.coff

    def dense(self):
        """Contiguous [N,H,W,C] copy of the slice (tests / API glue)."""
        return self.t[..., self.coff:self.coff + self.C].contiguous()


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PPYoloHipError('HIP path needs ROCm device tensors; got a %s tensor (no CPU fallback)'
                                      % t.device)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv2d_workspace_bytes(N, H, W, C, K, R, S, stride, pad, cfg=-1, splitk=0):
    return int(lib().ppy_conv2d_workspace_bytes(N, H, W, C, K, R, S, stride, pad, cfg, splitk))


def conv2d_pick(N, H, W, C, K, R, S, stride, pad):
    c, s = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().ppy_conv2d_pick(N, H, W, C, K, R, S, stride, pad, ctypes.byref(c), ctypes.byref(s)), 'conv2d_pick')
    return c.value, s.value


def split_weights_bf16x3(w_krsc):
    """fp32 weights -> the three bf16 planes [3, *w.shape] (int16 storage) the "bf16x3" conv kernels read."""
    _dev(w_krsc)
    assert w_krsc.is_contiguous() and w_krsc.dtype == torch.float32
    out = torch.empty((3,) + tuple(w_krsc.shape), dtype=torch.int16, device=w_krsc.device)
    check(lib().ppy_conv2d_split_weights_bf16x3(w_krsc.data_ptr(), w_krsc.numel(), out.data_ptr(), _stream()),
          'ppy_conv2d_split_weights_bf16x3')
    return out


AMAX_FLOATS_PER_IMAGE = 128      # PPY_AMAX_FLOATS_PER_IMAGE: 8 slots x 16 floats of a tracked per-image maximum


def split_weights_f16x2(w_krsc, scale):
    """-> (planes [2, *w.shape] int16 storage, scale_f16x2 [K]): the operands of the "f16x2" conv kernels."""
    _dev(w_krsc, scale)
    assert w_krsc.is_contiguous() and w_krsc.dtype == torch.float32 and scale.is_contiguous()
    K = w_krsc.shape[0]
    planes = torch.empty((2,) + tuple(w_krsc.shape), dtype=torch.int16, device=w_krsc.device)
    sc = torch.empty_like(scale)
    check(lib().ppy_conv2d_split_weights_f16x2(w_krsc.data_ptr(), K, w_krsc.numel() // K, scale.data_ptr(),
                                               planes.data_ptr(), sc.data_ptr(), _stream()),
          'ppy_conv2d_split_weights_f16x2')
    return planes, sc


def amax_slots(t=None, device=None, N=None):
    """A zeroed block of tracked per-image maximum slots for N images; with an NHWC tensor, pre-filled with each image's
    max|.| (tests / stand-alone calls)."""
    N = t.shape[0] if t is not None else N
    a = torch.zeros(N * AMAX_FLOATS_PER_IMAGE, dtype=torch.float32, device=device if t is None else t.device)
    if t is not None:
        a.view(N, AMAX_FLOATS_PER_IMAGE)[:, 0] = t.reshape(N, -1).abs().amax(dim=1)
    return a


def conv2d_bn_act(x, w_krsc, scale, shift, y, stride=1, pad=0, act=None, residual=None, posbias=None,
                  upsample2x=False, cfg=-1, splitk=0, ws=None, w_x3=None, w_f16=None, amax_in=None, amax_out=None,
                  posbias_f16=None, x_split=None, y_split=None, amax_in2=None):
    """x, y, residual: View.  w_krsc: [K,R,S,C].  w_x3: split_weights_bf16x3(w_krsc) or None.
    w_f16: split_weights_f16x2(w_krsc, scale) or None; amax_in / amax_out: amax_slots blocks or None.
    x_split: [N] per-image scales of a pre-split input; y_split: ([N] scale tensor to fill, bound_mul, bound_add): write y
    pre-split for its one consumer.  amax_in2: a second amax_slots block covering part of x's channels (the launch scales by the
    larger maximum).  See ppy_conv2d_bn_act_f32 / ppy_conv2d_bn_act_split_f32."""
    _dev(x.t, w_krsc, scale, shift, y.t)
    K, R, S, C = w_krsc.shape
    assert C == x.C and K == y.C and w_krsc.is_contiguous()
    if x_split is not None or y_split is not None or amax_in2 is not None:
        ys, ym, ya = y_split if y_split is not None else (None, 0.0, 0.0)
        rc = lib().ppy_conv2d_bn_act_split_f32(
            x.ptr, x.ld, w_krsc.data_ptr(), _p(w_x3), None if w_f16 is None else w_f16[0].data_ptr(), scale.data_ptr(),
            None if w_f16 is None else w_f16[1].data_ptr(), shift.data_ptr(),
            None if residual is None else residual.ptr, 0 if residual is None else residual.ld,
            _p(posbias), _p(posbias_f16), y.ptr, y.ld, x.N, x.H, x.W, C, K, R, S, stride, pad, ACT[act], int(bool(upsample2x)),
            cfg, splitk, _p(amax_in), _p(amax_out), _p(ws), 0 if ws is None else ws.numel() * ws.element_size(), _stream(),
            _p(x_split), _p(ys), float(ym), float(ya), _p(amax_in2))
        check(rc, 'ppy_conv2d_bn_act_split_f32')
        return
    rc = lib().ppy_conv2d_bn_act_f32(
        x.ptr, x.ld, w_krsc.data_ptr(), _p(w_x3), None if w_f16 is None else w_f16[0].data_ptr(), scale.data_ptr(),
        None if w_f16 is None else w_f16[1].data_ptr(), shift.data_ptr(),
        None if residual is None else residual.ptr, 0 if residual is None else residual.ld,
        _p(posbias), _p(posbias_f16), y.ptr, y.ld, x.N, x.H, x.W, C, K, R, S, stride, pad, ACT[act], int(bool(upsample2x)),
        cfg, splitk, _p(amax_in), _p(amax_out), _p(ws), 0 if ws is None else ws.numel() * ws.element_size(), _stream())
    check(rc, 'ppy_conv2d_bn_act_f32')


def conv3x3_conv1x1(x, x_split, amax_in, wA_f16, shiftA, wB_f16, shiftB, residual, y, t_mul, t_add, amax_out=None, pool=None):
    """conv2 -> conv3 of an identity bottleneck in one launch (ppy_conv3x3_conv1x1_f32, csrc/conv_b2b.hip).  x: View of the
    PRE-SPLIT input, x_split its [N] scales; wA_f16 / wB_f16: split_weights_f16x2 results; residual, y: Views; pool: View that
    receives AvgPool2d(2, 2) of y, or None."""
    _dev(x.t, x_split, amax_in, wA_f16[0], wA_f16[1], shiftA, wB_f16[0], wB_f16[1], shiftB, residual.t, y.t)
    KA, KB = shiftA.numel(), shiftB.numel()
    check(lib().ppy_conv3x3_conv1x1_f32(x.ptr, x.ld, x_split.data_ptr(), amax_in.data_ptr(), wA_f16[0].data_ptr(), wA_f16[1].data_ptr(),
                                        shiftA.data_ptr(), wB_f16[0].data_ptr(), wB_f16[1].data_ptr(), shiftB.data_ptr(), residual.ptr,
                                        residual.ld, y.ptr, y.ld, None if pool is None else pool.ptr, 0 if pool is None else pool.ld, x.N, x.H, x.W, x.C, KA, KB,
                                        float(t_mul), float(t_add), _p(amax_out),
                                        _stream()), 'ppy_conv3x3_conv1x1_f32')


def stream_first_cfg():
    """First conv cfg id of the streaming 1x1 kernel (csrc/conv_stream.hip; + variant 0 / 1)."""
    return lib().ppy_conv2d_stream_first_config()


def ws_first_cfg():
    """First conv cfg id of the f16x2 tiles with specialised waves (csrc/conv_ws.hip)."""
    return lib().ppy_conv2d_ws_first_config()


def ws_num_cfgs():
    return lib().ppy_conv2d_small_first_config() - lib().ppy_conv2d_ws_first_config()


def small_first_cfg():
    """First conv cfg id of the wave-private tiles for small outputs (csrc/conv_small.hip, round 6; four ids).  For these ids `splitk`
    counts k-parts inside the workgroup: no workspace, no combine launch, pre-split tensors allowed with splitk > 1."""
    return lib().ppy_conv2d_small_first_config()


def small_num_cfgs():
    return lib().ppy_conv2d_num_configs() - lib().ppy_conv2d_small_first_config()


def patch_first_cfg():
    """Conv cfg id of the patch kernel for the 3x3 stem layers (csrc/conv_patch.hip)."""
    return lib().ppy_conv2d_patch_first_config()


def conv1x1_expand(x, w_f16, shift, y, act=None, residual=None, pooled=None, variant=0, amax_in=None, amax_out=None):
    """The streaming 1x1 kernel (C == 64, f16x2 operands): y as conv2d_bn_act gives it and, with `pooled` (a View of
    [N, H/2, W/2, K]), the 2x2 average of y from the same epilogue.  w_f16: split_weights_f16x2(w_krsc, scale).
    See ppy_conv1x1_expand_f32."""
    _dev(x.t, w_f16[0], w_f16[1], shift, y.t)
    rc = lib().ppy_conv1x1_expand_f32(
        x.ptr, x.ld, w_f16[0].data_ptr(), w_f16[1].data_ptr(), shift.data_ptr(),
        None if residual is None else residual.ptr, 0 if residual is None else residual.ld, y.ptr, y.ld,
        None if pooled is None else pooled.ptr, 0 if pooled is None else pooled.ld, x.N, x.H, x.W, x.C, y.C, ACT[act], variant,
        _p(amax_in), _p(amax_out), _stream())
    check(rc, 'ppy_conv1x1_expand_f32')


def conv3x3_maxpool(x, w_f16, shift, pooled, act=None, amax_in=None, amax_out=None):
    """The last stem convolution (3x3 / stride 1 / pad 1, C = 32 -> K = 64, f16x2 operands) and the MaxPool2d(3, 2, 1) behind it in
    one launch: `pooled` (a View of [N, (H-1)//2+1, (W-1)//2+1, 64]) is all that is written.  See ppy_conv3x3_maxpool_f32."""
    _dev(x.t, w_f16[0], w_f16[1], shift, pooled.t)
    assert (pooled.N, pooled.H, pooled.W) == (x.N, (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1)
    rc = lib().ppy_conv3x3_maxpool_f32(x.ptr, x.ld, w_f16[0].data_ptr(), w_f16[1].data_ptr(), shift.data_ptr(), pooled.ptr, pooled.ld,
                                       x.N, x.H, x.W, x.C, pooled.C, ACT[act], _p(amax_in), _p(amax_out), _stream())
    check(rc, 'ppy_conv3x3_maxpool_f32')


def conv2d_train_fwd(x, w_krsc, w_f16, bias, y, stride, pad, cfg, amax_in, partials):
    """Training-mode convolution + the first pass of its BatchNorm from the epilogue: y = conv(x, w) + bias on the f16x2 tile `cfg`
    (conv_x3.hip / conv_ws.hip ids), (n, mean, M2) triples into `partials` (a float32 tensor of at least
    conv2d_bn_partials_bytes(M, K) bytes).  Returns the slice count for bn_train_stats_merge.  See ppy_conv2d_train_fwd_f32."""
    _dev(x.t, w_krsc, w_f16[0], w_f16[1], bias, y.t, partials)
    K, R, S, C = w_krsc.shape
    assert C == x.C and K == y.C and w_krsc.is_contiguous()
    n = ctypes.c_int(0)
    check(lib().ppy_conv2d_train_fwd_f32(x.ptr, x.ld, w_krsc.data_ptr(), w_f16[0].data_ptr(), w_f16[1].data_ptr(), bias.data_ptr(), y.ptr, y.ld,
                                         x.N, x.H, x.W, C, K, R, S, stride, pad, cfg, amax_in.data_ptr(), partials.data_ptr(),
                                         partials.numel() * partials.element_size(), ctypes.byref(n), _stream()), 'ppy_conv2d_train_fwd_f32')
    return n.value


def conv1x1_stats(x, w_f16, bias, Kout, variant, amax_in, partials):
    """The BatchNorm partials of a 1x1 convolution on the streaming kernel WITHOUT storing its output (frozen layers of the training
    step).  Returns the slice count for bn_train_stats_merge.  See ppy_conv1x1_stats_f32."""
    _dev(x.t, w_f16[0], w_f16[1], bias, partials)
    n = ctypes.c_int(0)
    check(lib().ppy_conv1x1_stats_f32(x.ptr, x.ld, w_f16[0].data_ptr(), w_f16[1].data_ptr(), bias.data_ptr(), x.N, x.H, x.W, x.C, Kout, variant,
                                      amax_in.data_ptr(), partials.data_ptr(), partials.numel() * partials.element_size(), ctypes.byref(n),
                                      _stream()), 'ppy_conv1x1_stats_f32')
    return n.value


def conv1x1_bn_apply(x, w_f16, bias, mean, invstd, gamma, beta, y, act=None, residual=None, variant=0, amax_in=None, amax_out=None):
    """1x1 convolution + BatchNorm on the given batch statistics + shortcut + activation in one launch of the streaming kernel
    (frozen layers of the training step).  See ppy_conv1x1_bn_apply_f32."""
    _dev(x.t, w_f16[0], w_f16[1], bias, mean, invstd, gamma, beta, y.t)
    check(lib().ppy_conv1x1_bn_apply_f32(x.ptr, x.ld, w_f16[0].data_ptr(), w_f16[1].data_ptr(), bias.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                         gamma.data_ptr(), beta.data_ptr(), None if residual is None else residual.ptr,
                                         0 if residual is None else residual.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, y.C, ACT[act], variant,
                                         _p(amax_in), _p(amax_out), _stream()), 'ppy_conv1x1_bn_apply_f32')


def conv2d_bn_partials_bytes(M, K):
    return int(lib().ppy_conv2d_bn_partials_bytes(M, K))


def bn_train_stats_merge(partials, slices, eps, momentum, mean, invstd, running_mean=None, running_var=None):
    """(n, mean, M2) triples [slices][C][3] of conv2d_train_fwd -> mean / invstd [C], running statistics updated in place.
    See ppy_bn_train_stats_merge_f32."""
    _dev(partials, mean, invstd, running_mean, running_var)
    check(lib().ppy_bn_train_stats_merge_f32(partials.data_ptr(), partials.numel() * partials.element_size(), slices, mean.numel(), eps, momentum,
                                             mean.data_ptr(), invstd.data_ptr(), _p(running_mean), _p(running_var), _stream()),
          'ppy_bn_train_stats_merge_f32')


def _bwd_ws(nbytes, device):
    return torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.float32, device=device)


def conv2d_dgrad(dy, w_krsc, dx, stride=1, pad=0, ws=None, cfg=-1, splitk=0, amax_dy=None):
    """dy: View [N,Ho,Wo,K]; w_krsc [K,R,S,C]; dx: View [N,H,W,C] (written).  See ppy_conv2d_dgrad_f32."""
    _dev(dy.t, w_krsc, dx.t)
    K, R, S, C = w_krsc.shape
    assert K == dy.C and C == dx.C and w_krsc.is_contiguous()
    need = int(lib().ppy_conv2d_dgrad_workspace_bytes(dx.N, dx.H, dx.W, C, K, R, S, stride, pad, cfg, splitk))
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = _bwd_ws(need, dx.t.device)
    check(lib().ppy_conv2d_dgrad_f32(dy.ptr, dy.ld, w_krsc.data_ptr(), dx.ptr, dx.ld, dx.N, dx.H, dx.W, C, K, R, S, stride, pad,
                                     cfg, splitk, _p(amax_dy), ws.data_ptr(), ws.numel() * ws.element_size(), _stream()), 'ppy_conv2d_dgrad_f32')
    return ws


PREP_SPLIT = 8      # PPY_PREP_SPLIT (include/ppyolo_hip.h)


class _WeightPrep(ctypes.Structure):      # PpyWeightPrep
    _fields_ = [('w', ctypes.c_void_p), ('fwd_planes', ctypes.c_void_p), ('fwd_scale', ctypes.c_void_p), ('dgrad_planes', ctypes.c_void_p),
                ('dgrad_scale', ctypes.c_void_p), ('dgrad_wt', ctypes.c_void_p), ('K', ctypes.c_int), ('R', ctypes.c_int), ('S', ctypes.c_int),
                ('C', ctypes.c_int), ('row0', ctypes.c_int), ('blk0', ctypes.c_int)]


class WeightPrepTable(object):
    """The operand forms of a set of trainable convolution weights, rebuilt by ONE call per step (ppy_train_prepare_weights_f16x2).
    entries: list of dicts with `w` ([K,R,S,C] fp32, contiguous, C % 32 == 0) and `dgrad` (bool).  After build(): per entry
    `planes`, `scale` (what split_weights_f16x2 returns) and, with dgrad, `dg_planes`, `dg_scale`, `dg_wt`."""

    def __init__(self, entries):
        self.entries = entries
        dev = entries[0]['w'].device
        rows = blks = 0
        arr = (_WeightPrep * len(entries))()
        for i, e in enumerate(entries):
            w = e['w']
            _dev(w)
            assert w.is_contiguous() and w.dtype == torch.float32 and w.dim() == 4 and w.shape[3] % 32 == 0
            K, R, S, C = w.shape
            Kp = (K + 31) // 32 * 32
            e['planes'] = torch.empty((2, K, R, S, C), dtype=torch.int16, device=dev)
            e['scale'] = torch.empty(K, dtype=torch.float32, device=dev)
            if e.get('dgrad'):
                e['dg_planes'] = torch.empty((2, C, R, S, Kp), dtype=torch.int16, device=dev)
                e['dg_scale'] = torch.empty(C, dtype=torch.float32, device=dev)
                e['dg_wt'] = torch.empty((C, R, S, Kp), dtype=torch.float32, device=dev)
            arr[i] = _WeightPrep(w.data_ptr(), e['planes'].data_ptr(), e['scale'].data_ptr(),
                                 e['dg_planes'].data_ptr() if e.get('dgrad') else None, e['dg_scale'].data_ptr() if e.get('dgrad') else None,
                                 e['dg_wt'].data_ptr() if e.get('dgrad') else None, K, R, S, C, rows, blks)
            rows += K
            blks += (C // 32) * PREP_SPLIT
        self.rows, self.blocks = rows, blks
        raw = bytes(bytearray(arr))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.colmax = torch.empty(max(1, blks * 32), dtype=torch.float32, device=dev)
        self.ptrs = [e['w'].data_ptr() for e in entries]

    def current(self):
        """The masters still live where the table says (a caller may have re-bound them)."""
        return all(e['w'].data_ptr() == q for e, q in zip(self.entries, self.ptrs))

    def build(self):
        check(lib().ppy_train_prepare_weights_f16x2(self.table.data_ptr(), len(self.entries), self.rows, self.blocks, self.colmax.data_ptr(),
                                                    self.colmax.numel() * 4, _stream()), 'ppy_train_prepare_weights_f16x2')


def conv2d_dgrad_prepared(dy, prep, dx, pad, ones, zeros, ws, cfg=-1, splitk=0, amax_dy=None):
    """conv2d_dgrad (stride 1, f16x2) on the flipped / transposed planes of a WeightPrepTable entry (`prep`)."""
    _dev(dy.t, dx.t, prep['dg_planes'])
    C, R, S, Kp = prep['dg_wt'].shape
    K = prep['w'].shape[0]
    assert K == dy.C and C == dx.C and amax_dy is not None
    need = int(lib().ppy_conv2d_dgrad_workspace_bytes(dx.N, dx.H, dx.W, C, K, R, S, 1, pad, cfg, splitk))
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = _bwd_ws(need, dx.t.device)
    check(lib().ppy_conv2d_dgrad_prepared_f32(dy.ptr, dy.ld, prep['dg_wt'].data_ptr(), prep['dg_planes'].data_ptr(), prep['dg_scale'].data_ptr(),
                                              ones.data_ptr(), zeros.data_ptr(), dx.ptr, dx.ld, dx.N, dx.H, dx.W, C, K, R, S, pad, cfg, splitk,
                                              _p(amax_dy), ws.data_ptr(), ws.numel() * ws.element_size(), _stream()),
          'ppy_conv2d_dgrad_prepared_f32')
    return ws


def conv2d_wgrad(x, dy, dw_krsc, stride=1, pad=0, ws=None, amax_x=None, amax_dy=None):
    """x: View [N,H,W,C]; dy: View [N,Ho,Wo,K]; dw_krsc [K,R,S,C] (written); amax_x / amax_dy (both or neither): tracked per-image
    maxima of the operands (amax_slots blocks) -> the f16x2 kernel.  See ppy_conv2d_wgrad_f32."""
    assert (amax_x is None) == (amax_dy is None)
    _dev(x.t, dy.t, dw_krsc)
    K, R, S, C = dw_krsc.shape
    assert K == dy.C and C == x.C and dw_krsc.is_contiguous() and dw_krsc.dtype == torch.float32
    need = int(lib().ppy_conv2d_wgrad_workspace_bytes(x.N, x.H, x.W, C, K, R, S, stride, pad))
    if need and (ws is None or ws.numel() * ws.element_size() < need):
        ws = _bwd_ws(need, x.t.device)
    check(lib().ppy_conv2d_wgrad_f32(x.ptr, x.ld, dy.ptr, dy.ld, dw_krsc.data_ptr(), x.N, x.H, x.W, C, K, R, S, stride, pad,
                                     _p(amax_x), _p(amax_dy), _p(ws), 0 if ws is None else ws.numel() * ws.element_size(), _stream()),
          'ppy_conv2d_wgrad_f32')
    return ws


# ---- training-step operators (csrc/train.hip, csrc/yolo_loss.hip) ---------------------------------------------------
def _ws_for(nbytes, ws, device):
    if ws is None or ws.numel() * ws.element_size() < nbytes:
        ws = _bwd_ws(nbytes, device)
    return ws


def bn_train_stats(x, eps, momentum, mean, invstd, running_mean=None, running_var=None, ws=None):
    """x: View; mean / invstd [C] written; running statistics updated in place.  See ppy_bn_train_stats_f32."""
    _dev(x.t, mean, invstd, running_mean, running_var)
    P = x.N * x.H * x.W
    ws = _ws_for(int(lib().ppy_bn_train_workspace_bytes(P, x.C)), ws, x.t.device)
    check(lib().ppy_bn_train_stats_f32(x.ptr, x.ld, P, x.C, float(eps), float(momentum), mean.data_ptr(), invstd.data_ptr(),
                                       _p(running_mean), _p(running_var), ws.data_ptr(), ws.numel() * 4, _stream()),
          'ppy_bn_train_stats_f32')
    return ws


def bn_train_apply(x, mean, invstd, gamma, beta, y, act=None, residual=None, amax_out=None):
    """amax_out: zeroed amax_slots(N) block that receives the per-image max|y| (operand scale of a following f16x2 convolution)."""
    _dev(x.t, mean, invstd, gamma, beta, y.t)
    check(lib().ppy_bn_train_apply_f32(x.ptr, x.ld, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                       None if residual is None else residual.ptr, 0 if residual is None else residual.ld, y.ptr,
                                       y.ld, x.N * x.H * x.W, x.C, ACT[act], x.H * x.W, _p(amax_out), _stream()), 'ppy_bn_train_apply_f32')


def bn_train_bwd(x, y, dy, mean, invstd, gamma, dx, dgamma, dbeta, act=None, ws=None, amax_dx=None):
    """amax_dx: zeroed amax_slots(N) block that receives the per-image max|dx| (operand scale of an f16x2 weight gradient)."""
    _dev(x.t, y.t, dy.t, mean, invstd, gamma, dx.t, dgamma, dbeta)
    P = x.N * x.H * x.W
    ws = _ws_for(int(lib().ppy_bn_train_workspace_bytes(P, x.C)), ws, x.t.device)
    check(lib().ppy_bn_train_bwd_f32(x.ptr, x.ld, y.ptr, y.ld, dy.ptr, dy.ld, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                     dx.ptr, dx.ld, dgamma.data_ptr(), dbeta.data_ptr(), P, x.C, ACT[act], x.H * x.W, _p(amax_dx),
                                     ws.data_ptr(), ws.numel() * 4, _stream()), 'ppy_bn_train_bwd_f32')
    return ws


def act_bwd(dy, y, dx, act):
    _dev(dy.t, y.t, dx.t)
    check(lib().ppy_act_bwd_f32(dy.ptr, dy.ld, y.ptr, y.ld, dx.ptr, dx.ld, dy.N * dy.H * dy.W, dy.C, ACT[act], _stream()),
          'ppy_act_bwd_f32')


def avgpool2x2_bwd(dy, dx):
    _dev(dy.t, dx.t)
    assert dy.H == dx.H // 2 and dy.W == dx.W // 2 and dy.C == dx.C
    check(lib().ppy_avgpool2x2_bwd_f32(dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, _stream()), 'ppy_avgpool2x2_bwd_f32')


def maxpool3x3s2_bwd(x, dy, dx):
    _dev(x.t, dy.t, dx.t)
    assert (dx.H, dx.W, dx.C) == (x.H, x.W, x.C) and dy.C == x.C
    check(lib().ppy_maxpool3x3s2_bwd_f32(x.ptr, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, x.N, x.H, x.W, x.C, _stream()), 'ppy_maxpool3x3s2_bwd_f32')


def zero_insert(dy, up, stride):
    """up[n, i*s, j*s] = dy[n, i, j], zeros elsewhere (see ppy_zero_insert_f32)."""
    _dev(dy.t, up.t)
    assert dy.C == up.C and dy.N == up.N
    check(lib().ppy_zero_insert_f32(dy.ptr, dy.ld, up.ptr, up.ld, dy.N, dy.H, dy.W, dy.C, up.H, up.W, stride, _stream()), 'ppy_zero_insert_f32')


def upsample2x_bwd(dy, dx, accumulate=False):
    _dev(dy.t, dx.t)
    assert dy.H == 2 * dx.H and dy.W == 2 * dx.W and dy.C == dx.C
    check(lib().ppy_upsample2x_bwd_f32(dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, int(bool(accumulate)), _stream()),
          'ppy_upsample2x_bwd_f32')


def spp_bwd(x, dy, dx, ws=None):
    """x: View [N,H,W,C] (the SPP input); dy: View [N,H,W,4C]; dx: View [N,H,W,C]."""
    _dev(x.t, dy.t, dx.t)
    assert dy.C == 4 * x.C and dx.C == x.C
    ws = _ws_for(int(lib().ppy_spp_bwd_workspace_bytes(x.N, x.H, x.W, x.C)), ws, x.t.device)
    check(lib().ppy_spp_bwd_f32(x.ptr, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, x.N, x.H, x.W, x.C, ws.data_ptr(), ws.numel() * 4,
                                _stream()), 'ppy_spp_bwd_f32')
    return ws


def dropblock_mask(mask, scale, keep_prob, seed, block_size=3, ws=None):
    """mask: [N,H,W,C] fp32 (written: 1 = keep); scale: [1] fp32 (numel / sum(mask))."""
    _dev(mask, scale)
    N, H, W, C = mask.shape
    ws = _ws_for(int(lib().ppy_dropblock_workspace_bytes(N, H, W, C)), ws, mask.device)
    check(lib().ppy_dropblock_mask_f32(mask.data_ptr(), scale.data_ptr(), N, H, W, C, block_size, float(keep_prob), int(seed),
                                       ws.data_ptr(), ws.numel() * 4, _stream()), 'ppy_dropblock_mask_f32')
    return ws


def dropblock_apply(x, mask, scale, y):
    _dev(x.t, mask, scale, y.t)
    assert mask.is_contiguous() and tuple(mask.shape) == (x.N, x.H, x.W, x.C)
    check(lib().ppy_dropblock_apply_f32(x.ptr, x.ld, mask.data_ptr(), scale.data_ptr(), y.ptr, y.ld, x.N * x.H * x.W, x.C, _stream()),
          'ppy_dropblock_apply_f32')


def sgd_momentum(param, grad, velocity, lr, momentum, weight_decay, first_step):
    _dev(param, grad, velocity)
    assert param.is_contiguous() and grad.is_contiguous() and velocity.is_contiguous() and param.numel() == grad.numel() == velocity.numel()
    check(lib().ppy_sgd_momentum_f32(param.data_ptr(), grad.data_ptr(), velocity.data_ptr(), param.numel(), float(lr), float(momentum),
                                     float(weight_decay), int(bool(first_step)), _stream()), 'ppy_sgd_momentum_f32')


def ema_update(shadow, param, step, ema_decay):
    """One ExponentialMovingAverage.update (reference model/EMA.py:29-44) over flat fp32 tensors; returns the decay used."""
    import numpy as np
    _dev(shadow, param)
    assert shadow.is_contiguous() and param.is_contiguous() and shadow.numel() == param.numel()
    decay = min(ema_decay, (1 + step) / (10 + step))
    check(lib().ppy_ema_update_f32(shadow.data_ptr(), param.data_ptr(), param.numel(), float(np.float32(decay)),
                                   float(np.float32(1 - decay)), _stream()), 'ppy_ema_update_f32')
    return decay


def add_inplace(dst, src):
    _dev(dst.t, src.t)
    assert dst.C == src.C
    check(lib().ppy_add_inplace_f32(dst.ptr, dst.ld, src.ptr, src.ld, dst.N * dst.H * dst.W, dst.C, _stream()), 'ppy_add_inplace_f32')


def upsample2x(x, y):
    _dev(x.t, y.t)
    assert y.H == 2 * x.H and y.W == 2 * x.W and y.C == x.C
    check(lib().ppy_upsample2x_f32(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, _stream()), 'ppy_upsample2x_f32')


def channel_sum(dy, out, ws=None):
    _dev(dy.t, out)
    P = dy.N * dy.H * dy.W
    ws = _ws_for(int(lib().ppy_bn_train_workspace_bytes(P, dy.C)), ws, dy.t.device)
    check(lib().ppy_channel_sum_f32(dy.ptr, dy.ld, P, dy.C, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), 'ppy_channel_sum_f32')
    return ws


def yolov3_loss(head_out, target, gt_box, anchors_px, num_classes, downsample, scale_x_y, ignore_thresh, iou_loss_weight, iou_aware,
                iou_aware_loss_weight, dout, loss6, accumulate=False, ws=None, amax_dout=None, iou_loss_square=True):
    """head_out / dout: View [N,S,S,*]; target [N,an,6+C,S,S]; gt_box [N,G,4]; loss6 [6] fp32.  See ppy_yolov3_loss_f32."""
    _dev(head_out.t, target, gt_box, dout.t, loss6)
    an = len(anchors_px)
    N, S = head_out.N, head_out.H
    assert head_out.H == head_out.W and target.is_contiguous() and gt_box.is_contiguous()
    assert tuple(target.shape) == (N, an, 6 + num_classes, S, S) and gt_box.shape[0] == N and gt_box.shape[2] == 4
    arr = (ctypes.c_float * (2 * an))(*[float(v) for a in anchors_px for v in a])
    ws = _ws_for(int(lib().ppy_yolov3_loss_workspace_bytes(N, S, an)), ws, head_out.t.device)
    check(lib().ppy_yolov3_loss_f32(head_out.ptr, head_out.ld, target.data_ptr(), gt_box.data_ptr(), gt_box.shape[1], arr, an, num_classes,
                                    N, S, int(downsample), float(scale_x_y), float(ignore_thresh), float(iou_loss_weight),
                                    int(bool(iou_loss_square)), int(bool(iou_aware)), float(iou_aware_loss_weight), dout.ptr, dout.ld, loss6.data_ptr(),
                                    int(bool(accumulate)), _p(amax_dout), ws.data_ptr(), ws.numel() * 4, _stream()), 'ppy_yolov3_loss_f32')
    return ws


def stem_conv(x_nchw, w_kcrs, scale, shift, y, act='relu', amax_out=None, mfma=False):
    """mfma: the bf16x3 MFMA form (K == 32; csrc/stem_pool.hip stem_conv_mfma_kernel) instead of the fp32 fma chain."""
    _dev(x_nchw, w_kcrs, scale, shift, y.t)
    N, C, H, W = x_nchw.shape
    assert C == 3 and x_nchw.is_contiguous() and w_kcrs.is_contiguous() and tuple(w_kcrs.shape[1:]) == (3, 3, 3)
    fn = lib().ppy_stem_conv3x3s2_nchw_x3_f32 if mfma else lib().ppy_stem_conv3x3s2_nchw_f32
    check(fn(x_nchw.data_ptr(), w_kcrs.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.ptr, y.ld, N, H, W, w_kcrs.shape[0],
             ACT[act], _p(amax_out), _stream()), 'ppy_stem_conv3x3s2_nchw%s_f32' % ('_x3' if mfma else ''))


def preprocess_images(images_u8, target_size, lut, out, swap_rb=True):
    """Decode.process_image on the device for a list of uint8 HWC device tensors (any sizes) -> out[i] = [3,S,S] float32.
    lut: device float32 [3,256] (ppyolo_hip.preprocess.normalisation_table)."""
    n = len(images_u8)
    _dev(lut, out, *images_u8)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n, 3, target_size, target_size)
    assert lut.dtype == torch.float32 and lut.is_contiguous() and tuple(lut.shape) == (3, 256)
    for im in images_u8:
        assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3 and im.stride(2) == 1 and im.stride(1) == 3
    check(lib().ppy_preprocess_u8_f32(
        n, (ctypes.c_void_p * n)(*[im.data_ptr() for im in images_u8]), (ctypes.c_int * n)(*[im.shape[0] for im in images_u8]),
        (ctypes.c_int * n)(*[im.shape[1] for im in images_u8]), (ctypes.c_int * n)(*[im.stride(0) for im in images_u8]),
        int(bool(swap_rb)), int(target_size), lut.data_ptr(), out.data_ptr(), _stream()), 'ppy_preprocess_u8_f32')


def maxpool3x3s2(x, y):
    _dev(x.t, y.t)
    check(lib().ppy_maxpool3x3s2_f32(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, _stream()), 'ppy_maxpool3x3s2_f32')


def avgpool2x2(x, y):
    _dev(x.t, y.t)
    check(lib().ppy_avgpool2x2_f32(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, _stream()), 'ppy_avgpool2x2_f32')


def spp(x, y5, y9, y13):
    _dev(x.t, y5.t)
    assert y5.ld == y9.ld == y13.ld
    check(lib().ppy_spp_f32(x.ptr, x.ld, y5.ptr, y9.ptr, y13.ptr, y5.ld, x.N, x.H, x.W, x.C, _stream()), 'ppy_spp_f32')


def dcn_out_hw(H, W, stride, pad):
    # reference model/custom_layers.py:567-568
    return (H + 2 * pad - 2) // stride, (W + 2 * pad - 2) // stride


def dcnv2_sample(x, offset_mask, cols, stride, pad):
    _dev(x.t, offset_mask.t, cols)
    Ho, Wo = offset_mask.H, offset_mask.W
    check(lib().ppy_dcnv2_sample_f32(x.ptr, x.ld, offset_mask.ptr, offset_mask.ld, cols.data_ptr(), x.N, x.H, x.W,
                                     x.C, Ho, Wo, stride, pad, _stream()), 'ppy_dcnv2_sample_f32')


def dcnv2_num_configs():
    """ids of the fused DCNv2 kernel: math scheme (0 exact fp32, 1 bf16x3, 2 f16x2) * tiles + tile"""
    return int(lib().ppy_dcnv2_num_configs())


DCN_TILES_4W = 6      # csrc/dcn_fused.hip: ids [0, 18) = scheme (fp32, bf16x3, f16x2) * 6 + four-wave tile; [18, ...) = eight-wave f16x2 tiles


def dcnv2_scheme(cfg):
    """'fp32' / 'bf16x3' / 'f16x2': the math scheme a fused-DCNv2 configuration id runs."""
    return ('fp32', 'bf16x3', 'f16x2')[min(cfg // DCN_TILES_4W, 2)]


def dcnv2_configs(math='f16x2'):
    """The configuration ids a plan in math mode `math` may use (the schemes up to its own)."""
    n = {'fp32': 1, 'bf16x3': 2}.get(math, 3) * DCN_TILES_4W
    return list(range(n)) + (list(range(3 * DCN_TILES_4W, dcnv2_num_configs())) if math == 'f16x2' else [])


def dcnv2_workspace_bytes(N, H, W, C, K, stride, pad, cfg=-1, splitk=0):
    return int(lib().ppy_dcnv2_workspace_bytes(N, H, W, C, K, stride, pad, cfg, splitk))


def dcnv2(x, w_krsc, scale, shift, offset_mask, y, stride, pad, act, ws, cfg=-1, splitk=0, w_x3=None, w_f16=None,
          amax_in=None, amax_out=None):
    _dev(x.t, w_krsc, offset_mask.t, y.t, ws)
    K = w_krsc.shape[0]
    check(lib().ppy_dcnv2_f32(x.ptr, x.ld, w_krsc.data_ptr(), _p(w_x3), None if w_f16 is None else w_f16[0].data_ptr(),
                              scale.data_ptr(), None if w_f16 is None else w_f16[1].data_ptr(), shift.data_ptr(), offset_mask.ptr,
                              offset_mask.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, K, stride, pad, ACT[act], cfg, splitk,
                              _p(amax_in), _p(amax_out), ws.data_ptr(), ws.numel() * ws.element_size(), _stream()), 'ppy_dcnv2_f32')


def dcnv2_backward(x, w_krsc, offset_mask, dy, dx, d_offset_mask, dw_krsc, stride, pad, ws=None):
    """Backward of dcnv2 (see ppy_dcnv2_backward_f32): x, dx Views [N,H,W,C]; offset_mask, d_offset_mask Views [N,Ho,Wo,27];
    dy View [N,Ho,Wo,K]; dw_krsc [K,3,3,C].  dx = sampling path only."""
    _dev(x.t, w_krsc, offset_mask.t, dy.t, dx.t, d_offset_mask.t, dw_krsc)
    K = w_krsc.shape[0]
    assert dw_krsc.is_contiguous() and tuple(dw_krsc.shape) == tuple(w_krsc.shape) and w_krsc.is_contiguous()
    need = int(lib().ppy_dcnv2_backward_workspace_bytes(x.N, x.H, x.W, x.C, K, stride, pad))
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = _bwd_ws(need, x.t.device)
    check(lib().ppy_dcnv2_backward_f32(x.ptr, x.ld, w_krsc.data_ptr(), offset_mask.ptr, offset_mask.ld, dy.ptr, dy.ld, dx.ptr, dx.ld,
                                       d_offset_mask.ptr, d_offset_mask.ld, dw_krsc.data_ptr(), x.N, x.H, x.W, x.C, K, stride, pad,
                                       ws.data_ptr(), ws.numel() * ws.element_size(), _stream()), 'ppy_dcnv2_backward_f32')
    return ws


def yolo_decode(head_out, anchors_px, downsample, num_classes, scale_x_y, iou_aware, iou_aware_factor, clip_bbox,
                im_size, boxes, box_offset, score_threshold, cand_key, cand_idx, cand_count, scores_dense=None):
    """head_out: View [N,S,S,nch]; boxes [N,M,4]; cand_* [N,cap] int32-storage; cand_count [N] int32."""
    _dev(head_out.t, im_size, boxes, cand_key, cand_idx, cand_count, scores_dense)
    A = len(anchors_px)
    flat = [float(v) for a in anchors_px for v in a]
    arr = (ctypes.c_float * len(flat))(*flat)
    assert head_out.H == head_out.W
    check(lib().ppy_yolo_decode_f32(head_out.ptr, head_out.ld, head_out.N, head_out.H, A, num_classes, arr,
                                    int(downsample), float(scale_x_y), int(bool(iou_aware)), float(iou_aware_factor),
                                    int(bool(clip_bbox)), im_size.data_ptr(), boxes.data_ptr(), boxes.shape[1],
                                    int(box_offset), float(score_threshold), cand_key.data_ptr(), cand_idx.data_ptr(),
                                    cand_count.data_ptr(), cand_key.shape[1], _p(scores_dense), _stream()),
          'ppy_yolo_decode_f32')


def yolo_decode_levels(head_outs, anchors_px, downsamples, num_classes, scale_x_y, iou_aware, iou_aware_factor, clip_bbox,
                       im_size, boxes, score_threshold, cand_key, cand_idx, cand_count):
    """All head levels in one launch.  head_outs: list of View [N,S,S,nch]; anchors_px: list (per level) of (w,h) lists.
    Level l writes box rows [sum_{k<l} S_k^2*A, ...) -- the reference's concatenation order (head.py:446-461)."""
    L = len(head_outs)
    _dev(*([h.t for h in head_outs] + [im_size, boxes, cand_key, cand_idx, cand_count]))
    A = len(anchors_px[0])
    assert all(len(a) == A for a in anchors_px) and all(h.H == h.W for h in head_outs)
    anchor_arrs = [(ctypes.c_float * (2 * A))(*[float(v) for a in lvl for v in a]) for lvl in anchors_px]
    offs, off = [], 0
    for h in head_outs:
        offs.append(off)
        off += h.H * h.W * A
    vp = ctypes.c_void_p
    check(lib().ppy_yolo_decode_levels_f32(
        L, (vp * L)(*[h.ptr for h in head_outs]), (ctypes.c_int * L)(*[h.ld for h in head_outs]),
        (ctypes.c_int * L)(*[h.H for h in head_outs]), (ctypes.c_int * L)(*[int(d) for d in downsamples]),
        (vp * L)(*[ctypes.cast(a, vp) for a in anchor_arrs]), (ctypes.c_int * L)(*offs), head_outs[0].N, A, num_classes,
        float(scale_x_y), int(bool(iou_aware)), float(iou_aware_factor), int(bool(clip_bbox)), im_size.data_ptr(),
        boxes.data_ptr(), boxes.shape[1], float(score_threshold), cand_key.data_ptr(), cand_idx.data_ptr(),
        cand_count.data_ptr(), cand_key.shape[1], _stream()), 'ppy_yolo_decode_levels_f32')


def nms_candidates(scores, score_threshold, cand_key, cand_idx, cand_count):
    _dev(scores, cand_key, cand_idx, cand_count)
    N, M, C = scores.shape
    assert scores.is_contiguous()
    check(lib().ppy_nms_candidates_f32(scores.data_ptr(), N, M, C, float(score_threshold), cand_key.data_ptr(),
                                       cand_idx.data_ptr(), cand_count.data_ptr(), cand_key.shape[1], _stream()),
          'ppy_nms_candidates_f32')


def matrix_nms_workspace(N, device):
    """Scratch tensor for matrix_nms on a batch of N images."""
    return torch.empty((int(lib().ppy_matrix_nms_workspace_bytes(N)) + 3) // 4, dtype=torch.float32, device=device)


def matrix_nms(boxes, num_classes, cand_key, cand_idx, cand_count, post_threshold, nms_top_k, keep_top_k,
               use_gaussian, gaussian_sigma, out_dets, out_count, out_keep, ws=None):
    _dev(boxes, cand_key, cand_idx, cand_count, out_dets, out_count, out_keep)
    N, M, _ = boxes.shape
    if ws is None:
        ws = matrix_nms_workspace(N, boxes.device)
    check(lib().ppy_matrix_nms_f32(boxes.data_ptr(), M, num_classes, cand_key.data_ptr(), cand_idx.data_ptr(),
                                   cand_count.data_ptr(), cand_key.shape[1], N, float(post_threshold),
                                   int(nms_top_k), int(keep_top_k), int(bool(use_gaussian)), float(gaussian_sigma),
                                   out_dets.data_ptr(), out_count.data_ptr(), out_keep.data_ptr(), ws.data_ptr(),
                                   ws.numel() * ws.element_size(), _stream()),
          'ppy_matrix_nms_f32')
