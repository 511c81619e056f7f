"""One training step of PP-YOLO on the MI355X (BASELINE config 5).

What the reference's `train.py:416-443` does per iteration -- `losses = model(images, None, False, gt_bbox, gt_class,
gt_score, targets)`, `all_loss.backward()`, `optimizer.step()` -- as a sequence of hand-written HIP kernels behind the C ABI:

  * training-mode forward of the WHOLE network: every convolution unfused from its BatchNorm, which runs on batch statistics
    (the reference never calls `.eval()`; `backbone.freeze()` only stops gradients: model/resnet_vd.py:174-200), DropBlock
    drawing a mask, CoordConv as a real concatenation (its two weight columns train);
  * YOLOv3Loss forward + backward in one kernel per head level (csrc/yolo_loss.hip);
  * backward through the head only (freeze_at = 5 in both configs): BatchNorm / LeakyReLU backward, conv wgrad and dgrad
    (csrc/conv_bwd.hip), nearest-upsample, SPP and DropBlock backward -- a tape recorded during the forward, replayed in
    reverse;
  * RCCL all-reduces of the gradients (one flat buffer, bucket by bucket as the backward finishes them) when several ranks train data-parallel, then
    SGD-momentum with the reference's parameter groups (weight decay on convolution weights only: custom_layers.py:167-215).

The convolutions run on the exact bf16x3 split (no tracked maxima needed); torch supplies memory, streams and
`torch.distributed` only.  There is no CPU path.
"""
import json
import os

import warnings

import torch

from . import ops as K
from ._lib import PPYoloHipError

NUM_FP32_CFGS = 31      # ids below: exact-fp32 MFMA tiles; the next nine: bf16x3 (csrc/conv_igemm.hip, conv_x3.hip)
TRAIN_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_gfx950_train.json')
TRAIN_TABLE_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_gfx950_train_f16x2.json')
NUM_X3_F16_FIRST, NUM_X3_F16_LAST = 40, 66      # ids of the f16x2 tiles with 2 / 3 / 4 LDS stages (csrc/conv_x3.hip)


class Act(object):
    """NHWC activation: channel slice [coff, coff + C) of a buffer [N, H, W, ld]; `g` = its gradient (an Act) once a
    consumer has produced one; `req` = whether anything upstream wants that gradient."""
    __slots__ = ('t', 'coff', 'C', 'g', 'req', 'amax', 'coordp')

    def __init__(self, t, coff=0, C=None, req=False, amax=None):
        self.t, self.coff, self.C, self.g, self.req = t, coff, (t.shape[3] - coff if C is None else C), None, req
        self.coordp = 0           # > 0: channels [C, coordp) of the buffer already hold a CoordConv's coordinates + zero padding (new_coord)
        self.amax = amax          # tracked per-image max|.| (ops.amax_slots block) or None: the operand scale of an f16x2 convolution

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def H(self):
        return self.t.shape[1]

    @property
    def W(self):
        return self.t.shape[2]

    def view(self):
        return K.View(self.t, self.coff, self.C)

    def slice(self, coff, C):
        return Act(self.t, self.coff + coff, C, self.req, self.amax)

    def dense_nchw(self):
        return self.t[..., self.coff:self.coff + self.C].permute(0, 3, 1, 2).contiguous()


def _r32(c):
    return (c + 31) // 32 * 32


class ModelSettings(object):
    """What the training step reads from a configuration object (config/ppyolo_2x.py), taken from the MODEL instead -- as the
    reference's train.py builds it: backbone(**cfg.backbone), YOLOv3Head(yolo_loss=YOLOv3Loss(iou_loss=..., ...), ...)."""

    def __init__(self, model):
        bb, hd = model.backbone, model.head
        yl = hd.yolo_loss
        if yl is None:
            raise PPYoloHipError('the head holds no loss settings: build it with yolo_loss=YOLOv3Loss(...) (reference train.py:241-249)')
        self.backbone_type = type(bb).__name__
        self.backbone = dict(freeze_at=bb.freeze_at, feature_maps=list(bb.feature_maps))
        self.head = dict(anchors=hd.anchors, anchor_masks=hd.anchor_masks, num_classes=hd.num_classes, downsample=hd.downsample,
                         conv_block_num=hd.conv_block_num, coord_conv=hd.coord_conv, spp=hd.use_spp, keep_prob=hd.keep_prob,
                         drop_block=bool(hd.drop_block), iou_aware=hd.iou_aware,
                         # head.set_dropblock(is_test=True): the DropBlock modules stay in the layer lists, as identities
                         drop_active=self.drop_active(hd))
        if hd.block_size != 3:
            raise PPYoloHipError('DropBlock block_size %r: the mask kernel implements the configurations\' 3' % (hd.block_size,))
        if bool(hd.iou_aware) != (yl._iou_aware_loss is not None):
            raise PPYoloHipError('head.iou_aware and YOLOv3Loss(iou_aware_loss=...) disagree')
        self.yolo_loss = dict(scale_x_y=yl.scale_x_y, ignore_thresh=yl._ignore_thresh)
        self.iou_loss = dict(loss_weight=yl._iou_loss._loss_weight, loss_square=yl._iou_loss.loss_square)
        self.iou_aware_loss = dict(loss_weight=yl._iou_aware_loss._loss_weight if yl._iou_aware_loss is not None else 0.0)
        self.optimizerBuilder = dict(optimizer=dict(momentum=0.9), regularizer=dict(factor=0.0))      # (unused: torch.optim steps)
        self.use_ema = False


    @staticmethod
    def drop_active(hd):
        dropping = [m for blk in hd.detection_blocks for m in blk.layers if type(m).__name__ == 'DropBlock']
        return bool(dropping) and not all(m.is_test for m in dropping)


class TrainStep(object):
    def __init__(self, model, cfg=None, world_size=1, external_optimizer=False, seed_rank=None):
        """cfg: a configuration object (config/ppyolo_2x.py), or None = read the settings from the model's own objects.
        external_optimizer: the parameters live in the MODULE and something else (torch.optim, the reference's loop) updates
        them: they are re-read at every forward, step() / sgd() / the fused EMA are off -- see loss_dict()."""
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise PPYoloHipError('the training step needs the model on a ROCm device (got %s); there is no CPU path' % dev)
        cfg = ModelSettings(model) if cfg is None else cfg
        self.freeze_at = int(cfg.backbone.get('freeze_at', 5))
        if not 0 <= self.freeze_at <= 5:
            raise PPYoloHipError('freeze_at = %d: 0..5 (the reference configurations use 5: the head trains)' % self.freeze_at)
        if any(float(v) != 1.0 for v in cfg.backbone.get('lr_mult_list', [1.0])):
            raise PPYoloHipError('lr_mult_list other than ones is not implemented (one learning rate for all groups)')
        self.model, self.cfg, self.dev, self.world = model, cfg, dev, world_size
        self.external = bool(external_optimizer)
        self.sd = model.state_dict()                       # tensors alias the module's parameters / buffers
        want = [k for k, _ in model.named_parameters() if self._stage_of(k) > self.freeze_at]
        if self.external:
            got = [k for k, q in model.named_parameters() if q.requires_grad]
            if got != want:
                odd = sorted(set(got) ^ set(want))
                raise PPYoloHipError('the tensors that require gradients must be exactly the stages above freeze_at = %d and the head '
                                     '(call model.backbone.freeze(), reference train.py:264); %d differ, e.g. %s'
                                     % (self.freeze_at, len(odd), odd[0]))
        self.train_keys = want
        self._wcache = {}
        self._const = {}
        self._coord_bufs = {}
        self._prep, self._prep_done = None, False      # ops.WeightPrepTable of the trainable convolutions (built after the first step)
        self.ws = torch.empty(96 << 20, dtype=torch.float32, device=dev)     # conv split-K / dgrad / wgrad / reductions
        # The weight gradient of a head convolution has no consumer before the optimizer: it runs on a SECOND stream beside the data
        # gradient chain (its own workspace; the operands are kept alive until the join at the end of the backward).  Same kernels,
        # same results bit for bit; PPYOLO_HIP_TRAIN_WGRAD_STREAM=0 puts it back in line.
        self._wgrad_side = os.environ.get('PPYOLO_HIP_TRAIN_WGRAD_STREAM', '1') == '1'
        # frozen 1x1 layers on the streaming kernel: BatchNorm from the convolution's own epilogue, no raw tensor (conv_unit)
        # (PPYOLO_HIP_TRAIN_BN_EPILOGUE: 0 = off, 1 = the layers the table puts on the streaming kernel, 2 (default) = those and every
        # frozen C = 128 1x1 layer the kernel accepts, whatever tile the table names: 11.82 -> 11.58 -> 11.48 ms on the R50vd-608 step)
        self.bn_epilogue = os.environ.get('PPYOLO_HIP_TRAIN_BN_EPILOGUE', '2') in ('1', '2')
        self.bn_epilogue_all = os.environ.get('PPYOLO_HIP_TRAIN_BN_EPILOGUE', '2') == '2'
        self._wstream = torch.cuda.Stream(device=dev) if self._wgrad_side else None
        self._ws_side = None
        self._wkeep = []
        self._wpending = False
        # ... and so do the optimizer step, the EMA update and the re-split of the updated weights (sgd, _prepare_weights): the next
        # step's frozen layers do not read a trainable parameter, so its forward starts while they run; the first use of a trainable
        # parameter (weight() / param()) or a reader outside the step (sync_to_model, grads) waits for them (_await_params).
        # PPYOLO_HIP_TRAIN_ASYNC_TAIL=0: in line.
        self._async_tail = self._wgrad_side and os.environ.get('PPYOLO_HIP_TRAIN_ASYNC_TAIL', '1') == '1'
        self._params_pending = False
        self.steps_done = 0
        self.momentum = cfg.optimizerBuilder['optimizer']['momentum']
        self.weight_decay = cfg.optimizerBuilder['regularizer']['factor']
        self.gflat = None
        self.G, self.V, self.P = {}, {}, {}
        self.use_ema = bool(getattr(cfg, 'use_ema', True))          # reference config/ppyolo_2x.py:92-94
        self.ema_decay = float(getattr(cfg, 'ema_decay', 0.9998))
        self.ema_steps = 0
        self.masks = None
        self.seed = 0
        # DropBlock draws differ between data-parallel ranks (the reference's processes each own a torch.rand stream) and
        # between runs with another PPYOLO_HIP_SEED
        rank = seed_rank if seed_rank is not None else (
            torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0)
        self.seed_base = (int(os.environ.get('PPYOLO_HIP_SEED', '0')) * 0x2545F491 + rank * 0x5851F42D) & 0xFFFFFFFFFFFF
        self.acts = None
        self.flops = 0                      # algorithmic convolution FLOPs (2 * MAC) of the last forward + backward
        # tile configurations: the measured bf16x3 table of the inference path knows the backbone's shapes; the head's own
        # (CoordConv channels padded to 32, the data gradients' transposed geometries) are in tuned_gfx950_train.json,
        # written by autotune() below.  Keys are the geometry the forward kernel sees.
        from .engine import tuned_table
        self._tuned = dict(tuned_table('bf16x3'))
        if os.path.exists(TRAIN_TABLE):
            with open(TRAIN_TABLE) as fh:
                self._tuned.update(json.load(fh))
        # forward convolutions on the f16x2 kernels (3 MFMA products instead of 6) where the input's maximum is tracked -- by
        # bn_train_apply for every normalised activation, propagated through concatenations / pooling / DropBlock here;
        # PPYOLO_HIP_TRAIN_MATH=bf16x3 keeps every convolution on the exact bf16 split
        self.f16 = os.environ.get('PPYOLO_HIP_TRAIN_MATH', 'f16x2') == 'f16x2'
        # PPYOLO_HIP_TRAIN_MATH=fp32 (bench.py's value_fp32_exact leg; with PPY_WGRAD_FP32=1 and PPY_DGRAD_FP32=1 in the environment of
        # the process): every convolution, data gradient and weight gradient on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
        self.fp32 = os.environ.get('PPYOLO_HIP_TRAIN_MATH', 'f16x2') == 'fp32'
        self._tuned_f = dict(tuned_table('f16x2')) if self.f16 else {}
        if self.f16 and os.path.exists(TRAIN_TABLE_F16):
            with open(TRAIN_TABLE_F16) as fh:
                self._tuned_f.update(json.load(fh))
        self._amax_arena, self._amax_next = None, 0
        # Gradient buckets go out as asynchronous all-reduces DURING the backward.  With backend nccl (= RCCL) that puts RCCL's fp32
        # sum kernels beside this library's 16-bit-MFMA kernels on the same CUs -- the co-residence under which a packed-fp32
        # instruction form misreads.  librccl's gfx950 code holds 945 v_pk_*_f32, none in that form
        # (tools/rccl_pk_scan.py -> profiles/r03_rccl_pk_scan.txt), but the pair has never executed on hardware (no multi-GPU
        # box): under nccl the overlap is therefore OPT-IN (PPYOLO_HIP_TRAIN_OVERLAP=1) and the default is one collective after
        # the backward, when no MFMA kernel of this rank is in flight; other backends (gloo: host reductions) overlap by default.
        ov = os.environ.get('PPYOLO_HIP_TRAIN_OVERLAP')
        nccl = world_size > 1 and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_backend() == 'nccl'
        self.overlap = ((ov == '1') if nccl else (ov != '0')) and not self.external
        self._buckets, self._pending, self._works, self._reduced = None, {}, [], []
        self.tune = False                   # True: measure shapes the tables do not know while stepping (autotune())
        self.fuse_stats = os.environ.get('PPYOLO_HIP_TRAIN_FUSE_STATS', '1') == '1'      # BatchNorm statistics from the conv epilogue
        self._bn_part = None
        self._measured = {}
        self._nbt = []                      # keys of the BatchNorm step counters touched by this forward (bumped in one launch)
        self._nbt_flat, self._nbt_keys = None, None
        # Round 5: with the whole backbone frozen (freeze_at = 5, the reference's configurations) its training-mode forward reads no
        # trainable parameter, so the NEXT batch's backbone can run on a third stream beside THIS batch's head forward / loss /
        # backward (prefetch_backbone, step(..., next_x=...)): same kernels on the same inputs in the same order per tensor --
        # bit-identical losses, gradients and running statistics -- with its own workspace, BatchNorm partials and (two alternating)
        # blocks of tracked-maximum slots.  PPYOLO_HIP_TRAIN_PREFETCH=0 ignores next_x.
        self._prefetch_on = os.environ.get('PPYOLO_HIP_TRAIN_PREFETCH', '1') == '1'
        self._bstream = None
        self._pref = None
        self._own_backbone_done = None
        self._coord_slot = 2                # which of a layer's coordinate-ready buffers new_coord hands out (0 / 1: prefetched backbones)
        self._b_res = None                  # [ws, bn_part, [arena0, arena1], which]

    @staticmethod
    def _stage_of(key):
        """Backbone stage (1..5) a state_dict key belongs to; 6 = the head."""
        return int(key[len('backbone.stage')]) if key.startswith('backbone.stage') else 6

    # ---- constants / buffers -------------------------------------------------------------------------------------
    def _vec(self, name, n, val):
        key = (name, n)
        if key not in self._const:
            self._const[key] = torch.full((n,), val, dtype=torch.float32, device=self.dev)
        return self._const[key]

    def new(self, N, H, W, C, ld=None, req=False, zero=False):
        ld = C if ld is None else ld
        t = (torch.zeros if zero else torch.empty)((N, H, W, ld), dtype=torch.float32, device=self.dev)
        return Act(t, 0, C, req)

    def new_coord(self, tag, N, H, W, C, req=False):
        """An activation whose only consumer is a CoordConv: a PERSISTENT buffer [N, H, W, r32(C + 2)] per layer whose channels
        [C, ...) hold x_range, y_range and the zero padding (written once); the producer fills [0, C) and coord_concat() hands
        the whole buffer to the convolution -- no concatenation copy per step (it was 12 x 27 us).  A step's forward and
        backward finish inside one call (loss_dict snapshots the gradients), so a layer's buffer is free again at the next
        step."""
        Cp = _r32(C + 2)
        # (a prefetched backbone writes the NEXT batch's tensor while this batch's head still reads its own: two alternating buffers
        # on that path, a third for the plain loop -- self._coord_slot, set by prefetch_backbone)
        key = (tag, N, H, W, C, self._coord_slot)
        t = self._coord_bufs.get(key)
        if t is None:
            t = torch.zeros((N, H, W, Cp), dtype=torch.float32, device=self.dev)
            t[..., C] = (torch.arange(0, W, dtype=torch.float32, device=self.dev) / (W - 1) * 2.0 - 1).view(1, 1, W)
            t[..., C + 1] = (torch.arange(0, H, dtype=torch.float32, device=self.dev) / (H - 1) * 2.0 - 1).view(1, H, 1)
            self._coord_bufs[key] = t
        a = Act(t, 0, C, req)
        a.coordp = Cp
        return a

    def new_amax(self, N):
        """A zeroed block of per-image maximum slots (one arena, zeroed once per step)."""
        n = N * K.AMAX_FLOATS_PER_IMAGE
        if self._amax_arena is None or self._amax_next + n > self._amax_arena.numel():
            self._amax_arena = torch.zeros(max(384 * n, 1 << 16), dtype=torch.float32, device=self.dev)
            self._amax_next = 0
        a = self._amax_arena[self._amax_next:self._amax_next + n]
        self._amax_next += n
        return a

    # ---- parameters in kernel layout ---------------------------------------------------------------------------------
    def weight(self, key, coord=False):
        """-> dict(krsc, planes, Cin): the convolution weight `key` ([K, C, R, S] in the state_dict) as KRSC, padded to a
        multiple of 32 input channels behind a CoordConv; trainable weights keep a MASTER copy here (updated by SGD, written
        back by sync_to_model) and get their bf16 planes re-split every step."""
        ent = self._wcache.get(key)
        if ent is not None and ent['trainable']:
            self._await_params()
        if ent is None:
            w = self.sd[key].detach().float()
            Kout, Cin, R, S = w.shape
            Cp = _r32(Cin) if (coord or Cin % 32) else Cin
            krsc = torch.zeros((Kout, R, S, Cp), dtype=torch.float32, device=self.dev)
            krsc[..., :Cin] = w.permute(0, 2, 3, 1)
            ent = dict(krsc=krsc, planes=None, f16=None, Cin=Cin, trainable=key in self.train_keys)
            self._wcache[key] = ent
        if ent.get('prep') is not None:    # f16x2 planes rebuilt for all trainable weights at the start of the step (_prepare_weights)
            ent['planes'] = None           # (bf16x3 planes, should a launch want them: split on demand from the current master)
            return ent
        if ent['trainable'] or (ent['planes'] is None and ent['f16'] is None):
            ent['planes'] = None       # bf16x3 planes: split on demand (_planes) -- a layer on the f16x2 kernels never needs them
            if self.f16:      # (planes, per-channel epilogue scale with the weight scale folded in) for a unit scale
                ent['f16'] = K.split_weights_f16x2(ent['krsc'], self._vec('one', ent['krsc'].shape[0], 1.0))
            else:
                self._planes(ent)
        return ent

    @staticmethod
    def _planes(ent):
        """The three bf16 planes of a weight entry (bf16x3 kernels), split when first asked for after an update."""
        if ent['planes'] is None:
            ent['planes'] = K.split_weights_bf16x3(ent['krsc'])
        return ent['planes']

    def _alloc_flat(self):
        """Parameters, gradients, velocities and EMA shadows of ALL trainable tensors in flat buffers, in kernel layout:
        convolution weights first (the weight-decay group), then biases and BatchNorm scales / offsets -- so that SGD is two
        launches, the EMA one, and data-parallel ranks average every gradient with ONE all-reduce.  The kernels read the
        parameters through views into `pflat` from now on; sync_to_model() writes them back into the module."""
        # (the reference decays conv_offset's bias like a weight: custom_layers.py:189-194)
        convs = [k for k in self.train_keys if k in self._wcache] + [k for k in self.train_keys if k.endswith('.conv_offset.bias')]
        rest = [k for k in self.train_keys if k not in convs]
        offs, total = {}, 0
        for k in convs + rest:
            shp = tuple(self._wcache[k]['krsc'].shape) if k in self._wcache else tuple(self.sd[k].shape)
            n = 1
            for d in shp:
                n *= d
            offs[k] = (total, n, shp)
            total += (n + 63) // 64 * 64
            if k == convs[-1]:
                self.n_decay = total                       # [0, n_decay): weight decay applies
        z = lambda: torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.gflat, self.vflat, self.pflat = z(), z(), z()
        for k, (o, n, shp) in offs.items():
            self.G[k] = self.gflat[o:o + n].view(shp)
            self.V[k] = self.vflat[o:o + n].view(shp)
            self.P[k] = self.pflat[o:o + n].view(shp)
            if k in self._wcache:
                self.P[k].copy_(self._wcache[k]['krsc'])
                self._wcache[k]['krsc'] = self.P[k]
            else:
                self.P[k].copy_(self.sd[k])
        self.sflat = self.pflat.clone() if self.use_ema else None        # EMA.register(): shadow = parameters

    def _pull_params(self):
        """external_optimizer: the module's parameters are the masters -- bring the kernel-layout copies up to date."""
        self.sd = self.model.state_dict()      # (a caller may have rebound a parameter's storage, as the reference's EMA.apply does)
        for k in self.train_keys:
            src = self.sd[k].detach()
            ent = self._wcache.get(k)
            if ent is not None:
                ent['krsc'][..., :ent['Cin']].copy_(src.permute(0, 2, 3, 1))
            elif k in self.P:
                self.P[k].copy_(src)

    def _await_params(self):
        """The current stream waits for the optimizer tail of the previous step (sgd / EMA / weight planes on the second stream)."""
        if self._params_pending:
            torch.cuda.current_stream(self.dev).wait_stream(self._wstream)
            self._params_pending = False

    def param(self, key):
        """A bias / BatchNorm scale or offset as the kernels should read it: the flat master copy once it exists."""
        if key in self.P:
            self._await_params()
            return self.P[key]
        return self.sd.get(key)

    # ---- forward ops ---------------------------------------------------------------------------------------------------
    def coord_concat(self, x):
        """CoordConv.__call__ (reference model/custom_layers.py:261-272) as a real concatenation, zero-padded to a multiple
        of 32 channels for the implicit GEMM: [x, x_range, y_range, 0 ...]."""
        Cp = _r32(x.C + 2)
        if x.coordp == Cp and x.coff == 0:               # produced into a coordinate-ready buffer (new_coord): nothing to copy
            return Act(x.t, 0, Cp, x.req, None if x.amax is None else torch.clamp_min(x.amax, 1.0))
        key = ('coord', x.H, x.W, Cp - x.C)
        if key not in self._const:
            xr = torch.arange(0, x.W, dtype=torch.float32, device=self.dev) / (x.W - 1) * 2.0 - 1
            yr = torch.arange(0, x.H, dtype=torch.float32, device=self.dev) / (x.H - 1) * 2.0 - 1
            g = torch.zeros((1, x.H, x.W, Cp - x.C), dtype=torch.float32, device=self.dev)
            g[0, :, :, 0] = xr.view(1, x.W)
            g[0, :, :, 1] = yr.view(x.H, 1)
            self._const[key] = g
        # one launch: [x | x_range, y_range, zeros]
        out = Act(torch.cat((x.t[..., x.coff:x.coff + x.C], self._const[key].expand(x.N, -1, -1, -1)), dim=3), 0, Cp, x.req,
                  None if x.amax is None else torch.clamp_min(x.amax, 1.0))          # (the coordinates lie in [-1, 1])
        return out

    def _choose(self, key, run, chunks, f16=False):
        """(tile configuration, split-K) of a convolution launch: the tables; measured on the spot when `self.tune`.  f16: the
        launch has the operands of the f16x2 kernels (split fp16 weights, tracked input maximum) -- its entries carry ':f'."""
        tab = self._tuned_f if f16 else self._tuned
        tkey = key + ':f' if f16 else key
        ent = tab.get(tkey)
        # PPYOLO_HIP_TRAIN_RETUNE=1: measure the f16x2 geometries again even where the table has an entry (new candidate kernels)
        again = f16 and os.environ.get('PPYOLO_HIP_TRAIN_RETUNE', '0') == '1' and tkey not in self._measured
        if (ent is None or again) and self.tune:
            ids = list(range(NUM_FP32_CFGS, NUM_FP32_CFGS + 9))                 # the nine bf16x3 tiles
            if f16:
                ids = list(range(NUM_X3_F16_FIRST, NUM_X3_F16_LAST + 1))        # the nine f16x2 tiles x {2, 3, 4} LDS stages
                ids += list(range(K.ws_first_cfg(), K.ws_first_cfg() + min(9, K.ws_num_cfgs())))      # ... and with specialised waves (csrc/conv_ws.hip;
                # not the k-parity tiles ws + 9..12 of round 6: they do not emit the BatchNorm statistics the training forward takes from the epilogue)
            best = None
            for cfg_id in ids:
                for splitk in (1, 2, 3, 4, 6, 8):
                    if splitk > 1 and chunks // splitk < 4:
                        continue
                    try:
                        run(cfg_id, splitk)
                    except PPYoloHipError:
                        continue
                    ms = None
                    for _ in range(2):
                        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        st.record()
                        for _ in range(4):
                            run(cfg_id, splitk)
                        en.record()
                        en.synchronize()
                        t = st.elapsed_time(en) / 4
                        ms = t if ms is None else min(ms, t)
                    if best is None or ms < best[2]:
                        best = [cfg_id, splitk, round(ms, 4)]
            if best is not None:
                ent = tab[tkey] = self._measured[tkey] = best
        return (ent[0], ent[1]) if ent else (-1, 0)

    def autotune(self, x_nchw, gt_box, targets, path=None):
        """One forward + backward with every convolution geometry the tables do not know measured on the device (nine
        bf16x3 tiles x split-K); `path`: where to write what was measured (the committed tuned_gfx950_train.json)."""
        self.tune = True
        try:
            self.forward_backward(x_nchw, gt_box, targets)
        finally:
            self.tune = False
        if path:       # bf16x3 geometries -> path, f16x2 ones (':f') -> path with '_f16x2' before the extension
            for suffix, sel in (('', False), ('_f16x2', True)):
                part = {k: v for k, v in self._measured.items() if k.endswith(':f') == sel}
                if part:
                    root, ext = os.path.splitext(path)
                    with open(root + suffix + ext, 'w') as fh:
                        json.dump(part, fh, indent=0, sort_keys=True)
        return dict(self._measured)

    def conv_unit(self, prefix, x, stride=1, act=None, res=None, coord=False, out=None, coord_out=False):
        """Conv2dUnit.forward in training mode (reference model/custom_layers.py:243-253): conv -> BatchNorm on batch
        statistics -> activation; records its backward when its parameters train."""
        sd = self.sd
        if prefix + '.conv.dcn_weight' in sd:
            return self._dcn_unit(prefix, x, stride, act)
        wkey = prefix + '.conv.weight'
        ent = self.weight(wkey, coord)
        trainable = ent['trainable']
        xin = self.coord_concat(x) if coord else x
        krsc = ent['krsc']
        Kout, R, S, Cp = krsc.shape
        if xin.C != Cp:
            raise PPYoloHipError('%s: input has %d channels, the weight %d' % (prefix, xin.C, Cp))
        pad = (R - 1) // 2
        Ho, Wo = K.conv_out_hw(xin.H, xin.W, R, S, stride, pad)
        bias = self.param(prefix + '.conv.bias')
        has_bn = prefix + '.bn.weight' in sd
        raw_box = []

        def get_raw():          # (allocated on first use: a layer whose BatchNorm is applied from the convolution's epilogue never stores it)
            if not raw_box:
                raw_box.append(self.new(xin.N, Ho, Wo, Kout, ld=_r32(Kout) if not has_bn else None, req=trainable, zero=not has_bn))
            return raw_box[0]
        one, b0 = self._vec('one', Kout, 1.0), bias if bias is not None else self._vec('zero', Kout, 0.0)
        use_f16 = self.f16 and xin.amax is not None and ent['f16'] is not None

        def run(cfg_id, splitk):
            K.conv2d_bn_act(xin.view(), krsc, one, b0, get_raw().view(), stride, pad, None, cfg=cfg_id, splitk=splitk, ws=self.ws,
                            w_x3=None if (use_f16 or self.fp32) else self._planes(ent), w_f16=ent['f16'] if use_f16 else None,
                            amax_in=xin.amax if use_f16 else None)
        key = 'conv:N%d:H%d:W%d:C%d:K%d:R%d:s%d' % (xin.N, xin.H, xin.W, Cp, Kout, R, stride)
        cfg_id, splitk = (-1, 0) if self.fp32 else self._choose(key, run, R * S * Cp // 32, use_f16)
        kp0 = K.ws_first_cfg() + 9
        if kp0 <= cfg_id < kp0 + 7:
            # a k-parity tile (round 6; the inference table's entry for this geometry -- the training tables fall back on it): those
            # kernels do not emit the BatchNorm statistics this forward takes from the epilogue; the same tile with one consumer group
            cfg_id = K.ws_first_cfg() + (0, 1, 2, 3, 1, 2, 3)[cfg_id - kp0]
        # BatchNorm statistics from the convolution's epilogue (the f16x2 kernels, one split): saves the
        # statistics kernel's pass over the raw output
        slices = 0
        s_first = K.stream_first_cfg()
        # Frozen 1x1 layers on the streaming kernel (the HBM-bound conv3 / shortcut layers of stage 2): the raw output is never
        # stored -- one launch for the statistics, one that applies the BatchNorm to its own accumulators (ops.conv1x1_bn_apply)
        epi = (self.bn_epilogue and has_bn and use_f16 and self.fuse_stats and not trainable and not coord and splitk == 1
               and s_first <= cfg_id < s_first + 2 and (R, S, stride) == (1, 1, 1))
        if (self.bn_epilogue_all and not epi and has_bn and use_f16 and self.fuse_stats and not trainable and not coord and (R, S, stride) == (1, 1, 1)
                and Cp == 128 and Kout % 128 == 0 and (Kout // 128) & (Kout // 128 - 1) == 0 and Kout // 128 <= 16 and xin.H * xin.W >= 32):
            epi, cfg_id, splitk = True, s_first, 1          # (the C = 128 layers whatever tile the table names: measured +0.8 %)
        if epi:
            need = K.conv2d_bn_partials_bytes(xin.N * Ho * Wo, Kout) // 4
            if self._bn_part is None or self._bn_part.numel() < need:
                self._bn_part = torch.empty(need, dtype=torch.float32, device=self.dev)
            slices = K.conv1x1_stats(xin.view(), ent['f16'], b0, Kout, cfg_id - s_first, xin.amax, self._bn_part)
        elif has_bn and use_f16 and self.fuse_stats and splitk == 1 and cfg_id >= NUM_X3_F16_FIRST:      # (every f16x2 kernel family)
            need = K.conv2d_bn_partials_bytes(xin.N * Ho * Wo, Kout) // 4
            if self._bn_part is None or self._bn_part.numel() < need:
                self._bn_part = torch.empty(need, dtype=torch.float32, device=self.dev)
            slices = K.conv2d_train_fwd(xin.view(), krsc, ent['f16'], b0, get_raw().view(), stride, pad, cfg_id, xin.amax, self._bn_part)
        else:
            run(cfg_id, splitk)
        self.flops += 2 * xin.N * Ho * Wo * Kout * R * S * ent['Cin']
        raw = None if epi else get_raw()
        if not has_bn:
            y = raw
            mean = invstd = None
        else:
            mean = torch.empty(Kout, dtype=torch.float32, device=self.dev)
            invstd = torch.empty(Kout, dtype=torch.float32, device=self.dev)
            if slices:
                K.bn_train_stats_merge(self._bn_part, slices, 1e-5, 0.1, mean, invstd, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'])
            else:
                K.bn_train_stats(raw.view(), 1e-5, 0.1, mean, invstd, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'], self.ws)
            self._nbt.append(prefix + '.bn.num_batches_tracked')
            y = out if out is not None else (self.new_coord(prefix, xin.N, Ho, Wo, Kout) if coord_out else self.new(xin.N, Ho, Wo, Kout))
            y.req = trainable
            y.amax = self.new_amax(xin.N) if self.f16 else None
            if epi:
                K.conv1x1_bn_apply(xin.view(), ent['f16'], b0, mean, invstd, self.param(prefix + '.bn.weight'), self.param(prefix + '.bn.bias'),
                                   y.view(), act, None if res is None else res.view(), cfg_id - s_first, xin.amax, y.amax)
            else:
                K.bn_train_apply(raw.view(), mean, invstd, self.param(prefix + '.bn.weight'), self.param(prefix + '.bn.bias'), y.view(), act,
                                 None if res is None else res.view(), y.amax)
        if trainable:
            def bwd_unit():
                self._conv_unit_bwd(prefix, x, xin, raw, y, mean, invstd, act, stride, pad, ent, res)
                self._grads_done(prefix)
            self.tape.append(bwd_unit)
        if self.acts is not None:          # debugging / tests: activations (and, after the backward, their gradients) by layer
            self.acts[prefix] = y
        return y

    def _prepare_weights(self):
        """Once per step, in front of the first convolution: the f16x2 operand planes of every trainable convolution weight (forward
        layout, and the flipped / transposed one of its data gradient) in three launches (ops.WeightPrepTable) -- the optimizer changed
        all of them.  The table is built after the first step, when the layers that ran on the f16x2 kernels are known."""
        if self._prep_done:
            return
        self._prep_done = True
        if self._prep is not None:
            if self._prep.current():
                if self._params_pending:      # behind the optimizer step on the second stream (the planes' readers wait: weight())
                    with torch.cuda.stream(self._wstream):
                        self._prep.build()
                else:
                    self._prep.build()
            else:                       # somebody re-bound a master copy: back to the per-layer splits, rebuild the table after this step
                for e in self._prep.entries:
                    self._wcache[e['key']]['prep'] = None
                self._prep = None

    def _build_prep(self):
        ents = [(k, e) for k, e in self._wcache.items() if e['trainable'] and e.get('f16') is not None and e['krsc'].shape[3] % 32 == 0
                and k in self.P]
        if not (self.f16 and ents) or os.environ.get('PPYOLO_HIP_TRAIN_PREP', '1') != '1':
            return
        entries = [dict(key=k, w=e['krsc'], dgrad=bool(e.get('dgrad_f16'))) for k, e in ents]
        self._prep = K.WeightPrepTable(entries)
        for (k, e), pe in zip(ents, entries):
            e['prep'] = pe
            e['f16'] = (pe['planes'], pe['scale'])

    def _dgrad(self, d_raw, krsc, dxin, stride, pad, cfg_id=-1, splitk=0, f16=False, ent=None):
        """Data gradient of a convolution; stride > 1 as the stride-1 data gradient of the zero-inserted output gradient.
        f16: on the f16x2 kernels, scaled by the tracked maxima of d_raw."""
        amax = d_raw.amax if f16 else None
        if stride == 1 and f16 and ent is not None:
            ent['dgrad_f16'] = True
            pe = ent.get('prep')
            if pe is not None and pe.get('dgrad'):
                C = krsc.shape[3]
                K.conv2d_dgrad_prepared(d_raw.view(), pe, dxin.view(), pad, self._vec('one', C, 1.0), self._vec('zero', C, 0.0), self.ws,
                                        cfg=cfg_id, splitk=splitk, amax_dy=amax)
                return
        if stride == 1:
            K.conv2d_dgrad(d_raw.view(), krsc, dxin.view(), 1, pad, self.ws, cfg=cfg_id, splitk=splitk, amax_dy=amax)
            return
        R = krsc.shape[1]
        H1, W1 = dxin.H + 2 * pad - R + 1, dxin.W + 2 * pad - R + 1
        up = self.new(d_raw.N, H1, W1, d_raw.C, ld=_r32(d_raw.C), zero=d_raw.C % 32 != 0)
        K.zero_insert(d_raw.view(), up.view(), stride)
        K.conv2d_dgrad(up.view(), krsc, dxin.view(), 1, pad, self.ws, amax_dy=amax)       # (zeros do not raise the maximum)

    def _wgrad(self, xin, d_raw, dw, stride, pad, amax_x, amax_dy):
        """Weight gradient of a convolution; with the side stream, issued there behind everything queued so far."""
        if not self._wgrad_side:
            K.conv2d_wgrad(xin.view(), d_raw.view(), dw, stride, pad, self.ws, amax_x, amax_dy)
            return
        if self._ws_side is None:
            self._ws_side = torch.empty_like(self.ws)
        main = torch.cuda.current_stream(self.dev)
        self._wstream.wait_stream(main)
        with torch.cuda.stream(self._wstream):
            K.conv2d_wgrad(xin.view(), d_raw.view(), dw, stride, pad, self._ws_side, amax_x, amax_dy)
        self._wkeep.append((xin.t, d_raw.t))          # (the allocator must not hand these out again before the join)
        self._wpending = True

    def _join_wgrad(self):
        """The current stream waits for the weight gradients issued on the side stream."""
        if self._wpending:
            torch.cuda.current_stream(self.dev).wait_stream(self._wstream)
            self._wpending = False
        self._wkeep = []

    def _conv_unit_bwd(self, prefix, x, xin, raw, y, mean, invstd, act, stride, pad, ent, res=None):
        dy = y.g
        if dy is None:
            raise PPYoloHipError('%s: no gradient reached this layer' % prefix)
        sd = self.sd
        if res is not None:
            # y = act(bn(conv) + res): the gradient in front of the activation goes to both branches
            dz = self.new(y.N, y.H, y.W, y.C)
            K.act_bwd(dy.view(), y.view(), dz.view(), act)
            if res.req:
                self.accum(res, dz)
            dy, act = dz, None
        f16 = self.f16 and xin.amax is not None       # weight gradient on the f16x2 kernel: both operands' maxima are tracked
        if mean is not None:
            d_raw = self.new(raw.N, raw.H, raw.W, raw.C)
            d_raw.amax = self.new_amax(raw.N) if f16 else None
            K.bn_train_bwd(raw.view(), y.view(), dy.view(), mean, invstd, self.param(prefix + '.bn.weight'), d_raw.view(),
                           self.G[prefix + '.bn.weight'], self.G[prefix + '.bn.bias'], act, self.ws, d_raw.amax)
        else:
            d_raw = dy
            if f16 and d_raw.amax is None:
                d_raw.amax = K.amax_slots(d_raw.t)      # (the loss gradient of an output convolution: three small tensors per step)
            K.channel_sum(dy.view(), self.G[prefix + '.conv.bias'], self.ws)
        self._wgrad(xin, d_raw, self.G[prefix + '.conv.weight'], stride, pad,
                    xin.amax if f16 and d_raw.amax is not None else None, d_raw.amax if f16 else None)
        unit = 2 * raw.N * raw.H * raw.W * raw.C * ent['krsc'].shape[1] * ent['krsc'].shape[2] * ent['Cin']
        self.flops += unit
        if x.req:
            self.flops += unit
            dxin = self.new(xin.N, xin.H, xin.W, xin.C)
            Kk, R = ent['krsc'].shape[0], ent['krsc'].shape[1]

            df16 = self.f16 and d_raw.amax is not None
            if stride == 1:
                def run(cfg_id, splitk):
                    self._dgrad(d_raw, ent['krsc'], dxin, 1, pad, cfg_id, splitk, df16, ent)
                # the data gradient runs the forward kernel on the transposed geometry: C' = K rounded up to 32, K' = C
                run(*self._choose('conv:N%d:H%d:W%d:C%d:K%d:R%d:s1' % (raw.N, raw.H, raw.W, _r32(Kk), xin.C, R), run, R * R * _r32(Kk) // 32,
                                  df16))
            else:
                self._dgrad(d_raw, ent['krsc'], dxin, stride, pad, f16=df16)
            self.accum(x, dxin.slice(0, x.C))

    def _dcn_unit(self, prefix, x, stride, act):
        """DCNv2 inside a backbone unit: offsets / masks from conv_offset, deformable contraction, BatchNorm on batch statistics
        (reference model/custom_layers.py:551-677); records its backward when stage 5 trains (freeze_at < 5)."""
        sd = self.sd
        co = self.weight(prefix + '.conv.conv_offset.weight')
        trainable = co['trainable']
        Ho, Wo = K.dcn_out_hw(x.H, x.W, stride, 1)
        om = self.new(x.N, Ho, Wo, 27, ld=32, zero=True)
        K.conv2d_bn_act(x.view(), co['krsc'], self._vec('one', 27, 1.0), self.param(prefix + '.conv.conv_offset.bias'), om.view(), stride, 1,
                        None, ws=self.ws, w_x3=None if self.fp32 else self._planes(co))
        w = self.weight(prefix + '.conv.dcn_weight')
        Kout = w['krsc'].shape[0]
        raw = self.new(x.N, Ho, Wo, Kout)
        use_f16 = self.f16 and x.amax is not None and w['f16'] is not None
        ent = (self._tuned_f if use_f16 else self._tuned).get('dcnf:N%d:H%d:W%d:C%d:K%d:R3:s%d%s' % (x.N, x.H, x.W, x.C, Kout, stride,
                                                                                                ':f' if use_f16 else ''))
        K.dcnv2(x.view(), w['krsc'], self._vec('one', Kout, 1.0), self._vec('zero', Kout, 0.0), om.view(), raw.view(), stride, 1, None,
                self.ws, cfg=ent[0] if (ent and not self.fp32) else -1, splitk=ent[1] if (ent and not self.fp32) else 0,
                w_x3=None if (use_f16 or self.fp32) else self._planes(w), w_f16=w['f16'] if use_f16 else None,
                amax_in=x.amax if use_f16 else None)
        self.flops += 2 * x.N * Ho * Wo * (Kout * 9 * x.C + 27 * 9 * x.C)
        mean = torch.empty(Kout, dtype=torch.float32, device=self.dev)
        invstd = torch.empty(Kout, dtype=torch.float32, device=self.dev)
        K.bn_train_stats(raw.view(), 1e-5, 0.1, mean, invstd, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'], self.ws)
        self._nbt.append(prefix + '.bn.num_batches_tracked')
        y = self.new(x.N, Ho, Wo, Kout, req=trainable)
        y.amax = self.new_amax(x.N) if self.f16 else None
        K.bn_train_apply(raw.view(), mean, invstd, self.param(prefix + '.bn.weight'), self.param(prefix + '.bn.bias'), y.view(), act, None,
                         y.amax)

        def bwd():
            if y.g is None:
                raise PPYoloHipError('%s: no gradient reached this layer' % prefix)
            d_raw = self.new(raw.N, raw.H, raw.W, raw.C)
            K.bn_train_bwd(raw.view(), y.view(), y.g.view(), mean, invstd, self.param(prefix + '.bn.weight'), d_raw.view(),
                           self.G[prefix + '.bn.weight'], self.G[prefix + '.bn.bias'], act, self.ws)
            dxs = self.new(x.N, x.H, x.W, x.C)
            d_om = self.new(x.N, Ho, Wo, 27, ld=32, zero=True)
            K.dcnv2_backward(x.view(), w['krsc'], om.view(), d_raw.view(), dxs.view(), d_om.view(), self.G[prefix + '.conv.dcn_weight'],
                             stride, 1, self.ws)
            # conv_offset: a plain 3x3 convolution with bias, same stride, whose output gradient is d_om
            K.conv2d_wgrad(x.view(), d_om.view(), self.G[prefix + '.conv.conv_offset.weight'], stride, 1, self.ws)
            K.channel_sum(d_om.view(), self.G[prefix + '.conv.conv_offset.bias'], self.ws)
            self.flops += 2 * 2 * x.N * Ho * Wo * (Kout * 9 * x.C + 27 * 9 * x.C)
            if x.req:
                self.accum(x, dxs)
                dxo = self.new(x.N, x.H, x.W, x.C)
                self._dgrad(d_om, co['krsc'], dxo, stride, 1)
                self.accum(x, dxo)
            self._grads_done(prefix)
        if trainable:
            self.tape.append(bwd)
        return y

    def accum(self, x, g):
        if x.g is None:
            x.g = g
        else:
            K.add_inplace(x.g.view(), g.view())

    # ---- backbone (stages 1 .. freeze_at forward only; the stages above record their backward like the head) -----------------
    def _stem(self, x_nchw):
        sd, p = self.sd, 'backbone.stage1_conv1_1'
        ent = self.weight(p + '.conv.weight')                  # (KRSC master, 3 channels padded to 32; the stem kernel reads KCRS)
        trainable = ent['trainable']
        N, _, H, W = x_nchw.shape
        Ho, Wo = K.conv_out_hw(H, W, 3, 3, 2, 1)
        Kout = ent['krsc'].shape[0]
        raw = self.new(N, Ho, Wo, Kout)
        K.stem_conv(x_nchw, ent['krsc'][..., :3].permute(0, 3, 1, 2).contiguous(), self._vec('one', Kout, 1.0), self._vec('zero', Kout, 0.0),
                    raw.view(), None)
        mean = torch.empty(Kout, dtype=torch.float32, device=self.dev)
        invstd = torch.empty(Kout, dtype=torch.float32, device=self.dev)
        K.bn_train_stats(raw.view(), 1e-5, 0.1, mean, invstd, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], self.ws)
        self._nbt.append(p + '.bn.num_batches_tracked')
        y0 = self.new(N, Ho, Wo, Kout, req=trainable)
        y0.amax = self.new_amax(N) if self.f16 else None
        K.bn_train_apply(raw.view(), mean, invstd, self.param(p + '.bn.weight'), self.param(p + '.bn.bias'), y0.view(), 'relu', None, y0.amax)
        if trainable:                                          # freeze_at = 0: the first convolution's weight gradient (no data gradient: the image)
            def bwd():
                d_raw = self.new(N, Ho, Wo, Kout)
                K.bn_train_bwd(raw.view(), y0.view(), y0.g.view(), mean, invstd, self.param(p + '.bn.weight'), d_raw.view(),
                               self.G[p + '.bn.weight'], self.G[p + '.bn.bias'], 'relu', self.ws)
                x4 = torch.zeros((N, H, W, 4), dtype=torch.float32, device=self.dev)
                x4[..., :3] = x_nchw.permute(0, 2, 3, 1)
                dw3 = torch.empty((Kout, 3, 3, 3), dtype=torch.float32, device=self.dev)
                K.conv2d_wgrad(K.View(x4, 0, 3), d_raw.view(), dw3, 2, 1, self.ws)
                g = self.G[p + '.conv.weight']
                g.zero_()
                g[..., :3] = dw3
                self._grads_done(p)
            self.tape.append(bwd)
        y = self.conv_unit('backbone.stage1_conv1_2', y0, 1, 'relu')
        y = self.conv_unit('backbone.stage1_conv1_3', y, 1, 'relu')
        Hp, Wp = K.conv_out_hw(y.H, y.W, 3, 3, 2, 1)
        o = self.new(N, Hp, Wp, y.C, req=y.req)
        o.amax = y.amax                                        # (a maximum of inputs)
        K.maxpool3x3s2(y.view(), o.view())
        if y.req:
            def pool_bwd():
                g = self.new(y.N, y.H, y.W, y.C)
                K.maxpool3x3s2_bwd(y.view(), o.g.view(), g.view())
                self.accum(y, g)
            self.tape.append(pool_bwd)
        return o

    def _avgpool(self, x):
        o = self.new(x.N, x.H // 2, x.W // 2, x.C, req=x.req)
        o.amax = x.amax                                        # (an average of inputs)
        K.avgpool2x2(x.view(), o.view())
        if x.req:
            def bwd():
                g = self.new(x.N, x.H, x.W, x.C)
                K.avgpool2x2_bwd(o.g.view(), g.view())
                self.accum(x, g)
            self.tape.append(bwd)
        return o

    def _bottleneck(self, p, x, stride, has_proj, is_first, coord_out=False):
        # reference model/resnet_vd.py:48-57, :81-87   (coord_out: the block's output feeds a CoordConv -- the head's first layer)
        y = self.conv_unit(p + '.conv1', x, 1, 'relu')
        y = self.conv_unit(p + '.conv2', y, stride, 'relu')
        if has_proj:
            s = self.conv_unit(p + '.conv4', x, stride, None) if is_first else self.conv_unit(p + '.conv4', self._avgpool(x), 1, None)
        else:
            s = x
        return self.conv_unit(p + '.conv3', y, 1, 'relu', res=s, coord_out=coord_out)             # relu(bn(conv3) + shortcut)

    def _basic(self, p, x, stride, is_first):
        # reference model/resnet_vd.py:256-267
        y = self.conv_unit(p + '.conv1', x, stride, 'relu')
        if stride == 2 or is_first:
            s = self.conv_unit(p + '.conv3', x, stride, None) if is_first else self.conv_unit(p + '.conv3', self._avgpool(x), 1, None)
        else:
            s = x
        return self.conv_unit(p + '.conv2', y, 1, 'relu', res=s)

    def backbone(self, x_nchw):
        cfg = self.cfg
        x = self._stem(x_nchw)
        feats = {}
        if cfg.backbone_type == 'Resnet50Vd':
            for stage, nblk in ((2, 3), (3, 4), (4, 6), (5, 3)):
                for b in range(nblk):
                    p = 'backbone.stage%d_%d' % (stage, b)
                    # round 6: the deepest feature map is read by ONE layer, the head's first CoordConv (reference model/head.py:178-182):
                    # it is produced into that layer's coordinate-ready buffer, so no step concatenates [C5 | x_range, y_range] any more
                    c5 = stage == 5 and b == nblk - 1 and bool(cfg.head.get('coord_conv', True)) and 5 in cfg.backbone['feature_maps']
                    x = self._bottleneck(p, x, 1 if (stage == 2 or b > 0) else 2, b == 0, stage == 2, coord_out=c5)
                feats[stage] = x
        else:
            for stage in (2, 3, 4, 5):
                for b in range(2):
                    p = 'backbone.stage%d_%d' % (stage, b)
                    x = self._basic(p, x, 2 if (b == 0 and stage > 2) else 1, b == 0 and stage == 2)
                feats[stage] = x
        return [feats[s] for s in (2, 3, 4, 5) if s in cfg.backbone['feature_maps']]

    # ---- head (trains) --------------------------------------------------------------------------------------------------------
    def drop_block(self, x, keep_prob, coord_tag=None):
        """DropBlock in training mode (reference model/custom_layers.py:303-342)."""
        if self.masks is not None:                         # parity tests: the reference's own mask ([N, C, H, W], 1 = keep)
            m = self.masks.pop(0).to(self.dev).permute(0, 2, 3, 1).contiguous()
            scale = torch.tensor([float(m.numel()) / float(m.sum())], dtype=torch.float32, device=self.dev)
        else:
            self.seed += 1
            seed = (self.seed_base + self.seed * 0x9E3779B1 + self.steps_done * 7919) & 0xFFFFFFFFFFFF
            if self._wgrad_side:
                # the mask depends on the seed only: drawn on the second stream, beside whatever the main stream still has queued
                # (the host runs ahead of the device); allocated there too, so that no block the main stream has just freed --
                # and may still be reading -- is written early
                if self._ws_side is None:
                    self._ws_side = torch.empty_like(self.ws)
                main = torch.cuda.current_stream(self.dev)
                with torch.cuda.stream(self._wstream):
                    m = torch.empty((x.N, x.H, x.W, x.C), dtype=torch.float32, device=self.dev)
                    scale = torch.empty(1, dtype=torch.float32, device=self.dev)
                    K.dropblock_mask(m, scale, keep_prob, seed, ws=self._ws_side)
                main.wait_stream(self._wstream)
                m.record_stream(main)
                scale.record_stream(main)
            else:
                m = torch.empty((x.N, x.H, x.W, x.C), dtype=torch.float32, device=self.dev)
                scale = torch.empty(1, dtype=torch.float32, device=self.dev)
                K.dropblock_mask(m, scale, keep_prob, seed, ws=self.ws)
        y = self.new(x.N, x.H, x.W, x.C, req=True) if coord_tag is None else self.new_coord(coord_tag, x.N, x.H, x.W, x.C, req=True)
        y.amax = None if x.amax is None else x.amax * scale     # y = x * mask * scale, mask in {0, 1}
        K.dropblock_apply(x.view(), m, scale, y.view())

        def bwd():
            g = self.new(x.N, x.H, x.W, x.C)
            K.dropblock_apply(y.g.view(), m, scale, g.view())
            self.accum(x, g)
        self.tape.append(bwd)
        return y

    def detection_block(self, p, x, hcfg, is_first):
        """DetectionBlock.__call__ (reference model/head.py:146-231); layer indices as in the state_dict keys."""
        nblk, coord = hcfg.get('conv_block_num', 2), hcfg.get('coord_conv', True)
        use_spp, drop, keep = hcfg.get('spp', True), hcfg.get('drop_block', True), hcfg.get('keep_prob', 0.9)
        active = hcfg.get('drop_active', True)
        idx = 0
        # (coord_out / coord_tag: the tensor's one consumer is a CoordConv -> produced into that layer's coordinate-ready buffer)
        for j in range(nblk):
            last = j == nblk - 1
            drop_here = drop and active and ((j == 0 and not is_first) or (last and is_first))
            if use_spp and is_first and j == 1:
                Cw = self.sd['%s.layers.%d.conv.weight' % (p, idx + 1)].shape[0]
                wide = self.new(x.N, x.H, x.W, 4 * Cw, req=True)
                slot0 = wide.slice(0, Cw)
                self.conv_unit('%s.layers.%d' % (p, idx + 1), x, 1, 'leaky', coord=coord, out=slot0)
                K.spp(slot0.view(), wide.slice(Cw, Cw).view(), wide.slice(2 * Cw, Cw).view(), wide.slice(3 * Cw, Cw).view())
                wide.amax = slot0.amax                             # (max-pooled copies of slot 0)

                def spp_bwd(wide=wide, slot0=slot0, Cw=Cw):
                    g = self.new(wide.N, wide.H, wide.W, Cw)
                    K.spp_bwd(slot0.view(), wide.g.view(), g.view(), self.ws)
                    self.accum(slot0, g)
                self.tape.append(spp_bwd)
                x = self.conv_unit('%s.layers.%d' % (p, idx + 3), wide, 1, 'leaky')
                x = self.conv_unit('%s.layers.%d' % (p, idx + 4), x, 1, 'leaky', coord_out=coord and not drop_here)
                idx += 5
            else:
                x = self.conv_unit('%s.layers.%d' % (p, idx + 1), x, 1, 'leaky', coord=coord)
                x = self.conv_unit('%s.layers.%d' % (p, idx + 2), x, 1, 'leaky', coord_out=coord and not drop_here)
                idx += 3
            if drop and j == 0 and not is_first:
                x = self.drop_block(x, keep, '%s.drop%d' % (p, j) if coord else None) if active else x
                idx += 1
        if drop and is_first:
            x = self.drop_block(x, keep, '%s.drop_last' % p if coord else None) if active else x
            idx += 1
        route = self.conv_unit('%s.layers.%d' % (p, idx + 1), x, 1, 'leaky', coord=coord, coord_out=coord)
        tip = self.conv_unit('%s.tip_layers.1' % p, route, 1, 'leaky', coord=coord)
        return route, tip

    def head(self, feats):
        """YOLOv3Head._get_outputs (reference model/head.py:381-398)."""
        hcfg = self.cfg.head
        n_lvl = len(hcfg['anchor_masks'])
        blocks = feats[::-1][:n_lvl]
        outs, route = [], None
        for i, feat in enumerate(blocks):
            if i > 0:
                Cr = route.C
                wide = (self.new_coord('head.route%d' % i, feat.N, feat.H, feat.W, Cr + feat.C, req=True) if hcfg.get('coord_conv', True)
                        else self.new(feat.N, feat.H, feat.W, Cr + feat.C, req=True))
                up = wide.slice(0, Cr)
                K.upsample2x(route.view(), up.view())
                wide.t[..., Cr:Cr + feat.C].copy_(feat.t[..., feat.coff:feat.coff + feat.C])
                wide.amax = None if (route.amax is None or feat.amax is None) else torch.maximum(route.amax, feat.amax)

                def up_bwd(route=route, wide=wide, Cr=Cr, feat=feat):
                    g = self.new(route.N, route.H, route.W, Cr)
                    K.upsample2x_bwd(wide.g.slice(0, Cr).view(), g.view())
                    self.accum(route, g)
                    if feat.req:           # the backbone stage behind this feature map trains (freeze_at < 5)
                        gf = self.new(feat.N, feat.H, feat.W, feat.C)
                        gf.t.copy_(wide.g.t[..., wide.g.coff + Cr:wide.g.coff + Cr + feat.C])
                        self.accum(feat, gf)
                self.tape.append(up_bwd)
                blk = wide
            else:
                blk = feat
            route, tip = self.detection_block('head.detection_blocks.%d' % i, blk, hcfg, i == 0)
            outs.append(self.conv_unit('head.yolo_output_convs.%d' % i, tip, 1, None))
            if i < n_lvl - 1:
                route = self.conv_unit('head.upsample_layers.%d' % (2 * i), route, 1, 'leaky')
        return outs

    # ---- the step ------------------------------------------------------------------------------------------------------------
    def forward_backward(self, x_nchw, gt_box, targets, dropblock_masks=None, inject_douts=None):
        """Forward + loss + backward.  -> the six loss terms (device tensor [6]: loss_xy, loss_wh, loss_obj, loss_cls, loss_iou,
        loss_iou_aware).  `dropblock_masks` / `inject_douts` are test hooks: the reference's own DropBlock masks, and a loss
        gradient computed elsewhere (the L1 terms of the loss have kinks, so two fp32 evaluations of the network that differ by
        1e-3 can disagree on a sign there; the loss kernel is checked on identical inputs in tests/test_gpu_train_ops.py)."""
        if not x_nchw.is_cuda:
            raise PPYoloHipError('the training step needs ROCm device tensors; there is no CPU path')
        self.tape = []
        self.flops = 0
        self._prep_done = False          # a step that raised after _prepare_weights() must not leave stale f16x2 planes behind
        if self._amax_arena is not None:
            self._amax_arena.zero_()
            self._amax_next = 0
        self.masks = list(dropblock_masks) if dropblock_masks is not None else None
        pref, self._pref = self._pref, None
        with torch.no_grad():
            if self.external:
                self._pull_params()
            self._prepare_weights()
            self._own_backbone_done = None
            if pref is not None and pref['x'] is x_nchw and pref['key'] == self._pref_key(x_nchw):
                # the backbone of this batch ran on the side stream during the previous step
                main = torch.cuda.current_stream(self.dev)
                main.wait_event(pref['event'])
                feats = pref['feats']
                for f in feats:
                    f.t.record_stream(main)          # (allocated under the side stream: not to be reused before the head has read it)
                self._nbt = list(pref['nbt']) + self._nbt
                self.flops += pref['flops']
            else:
                if pref is not None:
                    # a prefetch that is not used (another tensor, or the same one modified since): its backbone has already
                    # updated the BatchNorm running statistics once (no counter bump) -- the plain loop never would, so say so.
                    # The main-stream backbone below also read-modify-writes them: order it behind the side stream's.
                    warnings.warn('TrainStep: a prefetched backbone forward was discarded (the batch passed to step() is not the '
                                  'tensor given as next_x, or it was modified in between); its BatchNorm running-statistics '
                                  'update stays applied', RuntimeWarning)
                    torch.cuda.current_stream(self.dev).wait_event(pref['event'])
                feats = self.backbone(x_nchw.float().contiguous())
                # this step ran its own backbone on the main stream: the NEXT batch's prefetched backbone must not start before it
                # has finished (both update the same running statistics and, on step 0, the frozen-weight caches are made here)
                self._own_backbone_done = torch.cuda.Event()
                self._own_backbone_done.record(torch.cuda.current_stream(self.dev))
        return self.head_loss_backward(feats, gt_box, targets, inject_douts)

    @staticmethod
    def _pref_key(x):
        return (x.data_ptr(), tuple(x.shape), x._version)

    def prefetch_backbone(self, x_next, ready=None):
        """Enqueue the frozen backbone's training-mode forward of the NEXT batch on the side stream (freeze_at = 5 only; else a
        no-op).  `ready`: an event after which x_next may be read (default: everything issued on the current stream so far --
        call this BEFORE issuing the current batch's work if x_next is ready earlier; step(..., next_x=) does).  The next
        forward_backward(x_next, ...) picks the features up if x_next is still the same tensor, unmodified."""
        if not self._prefetch_on or self.freeze_at != 5 or self.external or self.tune or self.acts is not None:
            return False
        if self.world > 1 and not self.overlap:
            # data-parallel ranks under RCCL: the gradient all-reduce is kept clear of this rank's MFMA kernels unless the overlap is
            # opted into (PPYOLO_HIP_TRAIN_OVERLAP=1, see __init__) -- a prefetched backbone would run beside it
            return False
        if self._bstream is None:
            self._bstream = torch.cuda.Stream(device=self.dev)
            self._b_res = [torch.empty_like(self.ws), None, [None, None], 0]
        main = torch.cuda.current_stream(self.dev)
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(main)
        res = self._b_res
        res[3] ^= 1
        saved = (self.ws, self._bn_part, self._amax_arena, self._amax_next, self._nbt, self.flops, self.tape)
        self._coord_slot = res[3]
        self._bstream.wait_event(ready)
        try:
            with torch.cuda.stream(self._bstream), torch.no_grad():
                self.ws, self._bn_part = res[0], res[1]
                self._amax_arena, self._amax_next = res[2][res[3]], 0
                if self._amax_arena is not None:
                    self._amax_arena.zero_()
                self._nbt, self.flops, self.tape = [], 0, []
                feats = self.backbone(x_next.float().contiguous())
                assert not self.tape, 'prefetch_backbone: a frozen backbone records no backward'
                ev = torch.cuda.Event()
                ev.record(self._bstream)
                x_next.record_stream(self._bstream)          # (the allocator must not hand its block out while the side stream reads it)
                # the tensor itself is kept: identity + version decide whether the features are served (an address can be reused)
                self._pref = dict(x=x_next, key=self._pref_key(x_next), feats=feats, event=ev, nbt=self._nbt, flops=self.flops)
                res[1], res[2][res[3]] = self._bn_part, self._amax_arena
        finally:
            self.ws, self._bn_part, self._amax_arena, self._amax_next, self._nbt, self.flops, self.tape = saved
            self._coord_slot = 2
        return True

    def head_loss_backward(self, feats, gt_box, targets, inject_douts=None):
        """Head forward on the given backbone features (list of Act, shallowest first), loss, backward, -> loss terms [6]."""
        cfg, hcfg = self.cfg, self.cfg.head
        try:
            return self._head_loss_backward(feats, gt_box, targets, inject_douts, cfg, hcfg)
        finally:
            self.tape = []
            self._prep_done = False
            self._join_wgrad()           # (also after an exception: nothing of this step stays queued behind freed operands)
            # BatchNorm running statistics / counters were written through raw pointers (no autograd version bump) and the
            # counters may have been re-bound: inference executors folded from the old values are stale
            plans = getattr(self.model, '_plans', None)
            if plans is not None:
                plans.mark_dirty()

    def _head_loss_backward(self, feats, gt_box, targets, inject_douts, cfg, hcfg):
        with torch.no_grad():
            self._prepare_weights()
            outs = self.head(feats)
            if self._nbt:      # BatchNorm's num_batches_tracked of every layer this forward normalised: one launch
                self._bump_counters()
            if self.gflat is None:
                self._alloc_flat()
            loss6 = torch.zeros(6, dtype=torch.float32, device=self.dev)
            iou_aware = bool(hcfg.get('iou_aware', False))
            for i, out in enumerate(outs):
                anchors = [hcfg['anchors'][m] for m in hcfg['anchor_masks'][i]]
                dout = Act(torch.zeros_like(out.t), 0, out.C)
                if self.f16 and inject_douts is None:
                    dout.amax = self.new_amax(out.N)          # the loss kernel records max|dout| per image
                K.yolov3_loss(out.view(), targets[i].float().contiguous(), gt_box.float().contiguous(), anchors, hcfg['num_classes'],
                              hcfg['downsample'][i], cfg.yolo_loss['scale_x_y'], cfg.yolo_loss['ignore_thresh'], cfg.iou_loss['loss_weight'],
                              iou_aware, cfg.iou_aware_loss['loss_weight'] if iou_aware else 0.0, dout.view(), loss6, accumulate=i > 0,
                              ws=self.ws, amax_dout=dout.amax, iou_loss_square=cfg.iou_loss.get('loss_square', True))
                if inject_douts is not None:
                    dout.t[..., :out.C].copy_(inject_douts[i].to(self.dev).permute(0, 2, 3, 1))
                out.g = dout
            for fn in reversed(self.tape):
                fn()
            self._join_wgrad()
        self.outs = outs
        if self._prep is None and self.gflat is not None:
            self._build_prep()
        return loss6

    def _bump_counters(self):
        """num_batches_tracked += 1 for the BatchNorm layers of this forward.  The 76 counters of a R50vd are 76 zero-dimensional
        int64 buffers; torch._foreach_add_ on them turned out to be 72 tiny device copies per step (0.35 ms: rocprofv3,
        tools/probes/train_copy_probe.py).  They become views into ONE flat tensor (the modules' buffers are re-bound to the views,
        values kept; state_dict() keeps returning one 0-dim tensor per layer) and the bump is one `add_`."""
        keys, self._nbt = self._nbt, []
        if self._nbt_keys != keys or any(self.sd[k].data_ptr() != self._nbt_flat[i].data_ptr() for i, k in enumerate(keys)):
            try:
                flat = torch.stack([self.sd[k].detach().to(torch.int64) for k in keys])
                for i, k in enumerate(keys):
                    mod_name, leaf = k.rsplit('.', 1)
                    self.model.get_submodule(mod_name)._buffers[leaf] = flat[i]
                self.sd = self.model.state_dict()
                self._nbt_flat, self._nbt_keys = flat, keys
            except (AttributeError, KeyError):          # a model without that module tree: the counters one by one
                torch._foreach_add_([self.sd[k] for k in keys], 1)
                return
        self._nbt_flat.add_(1)

    # ---- data parallelism: gradient averaging overlapped with the backward --------------------------------------------------------
    @staticmethod
    def _bucket_of(key):
        """Gradient bucket of a parameter: a detection block, the head's output / transition convolutions, a backbone stage --
        the units in which the backward finishes its gradients (last layers first)."""
        q = key.split('.')
        if q[0] == 'backbone':
            return q[1][:6]                      # 'stage5'
        return '.'.join(q[:3]) if q[1] == 'detection_blocks' else 'head.tail'

    def _make_buckets(self):
        """{bucket: [pending unit count, [(start, end) ranges of gflat]]}: the keys of a bucket are (nearly) contiguous in
        both parameter groups of the flat layout, so a bucket is two or three ranges."""
        base = self.gflat.data_ptr()
        spans, units = {}, {}
        for k in self.train_keys:
            b = self._bucket_of(k)
            o = (self.G[k].data_ptr() - base) // 4
            spans.setdefault(b, []).append((o, o + (self.G[k].numel() + 63) // 64 * 64))
            units.setdefault(b, set()).add(self._unit_of(k))
        self._buckets = {}
        for b, iv in spans.items():
            iv.sort()
            merged = [list(iv[0])]
            for a, e in iv[1:]:
                if a <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], e)
                else:
                    merged.append([a, e])
            self._buckets[b] = dict(units=units[b], ranges=[(a, min(e, self.gflat.numel())) for a, e in merged])

    @staticmethod
    def _unit_of(key):
        """The Conv2dUnit prefix a parameter key belongs to ('....conv.weight' / '.bn.bias' / '.conv.conv_offset.bias' ...)."""
        for tail in ('.conv.conv_offset.weight', '.conv.conv_offset.bias', '.conv.dcn_weight', '.conv.weight', '.conv.bias', '.bn.weight',
                     '.bn.bias'):
            if key.endswith(tail):
                return key[:-len(tail)]
        return key

    def _grads_done(self, prefix):
        """All gradients of unit `prefix` are written: when that completes a bucket, start its all-reduce while the rest of
        the backward runs (the collective is queued behind the kernels issued so far and proceeds on RCCL's own stream)."""
        if self.world <= 1 or not self.overlap or self.gflat is None:
            return
        if self._buckets is None:
            self._make_buckets()
        if not self._pending:
            self._pending = {b: set(v['units']) for b, v in self._buckets.items()}
        b = self._bucket_of(prefix + '.conv.weight')
        left = self._pending.get(b)
        if left is None:
            return
        left.discard(prefix)
        if not left:
            del self._pending[b]
            if self._wpending:           # the bucket's weight gradients may still be running on the side stream
                torch.cuda.current_stream(self.dev).wait_stream(self._wstream)
            for a, e in self._buckets[b]['ranges']:
                self._works.append(torch.distributed.all_reduce(self.gflat[a:e], async_op=True))
            self._reduced.append(b)

    def all_reduce(self):
        """Data-parallel ranks: average ALL gradients (RCCL over xGMI; BatchNorm stays per GPU, like the reference's
        'sync_bn' -> 'bn' alias, model/custom_layers.py:28-29).  Buckets whose gradients were complete during the backward are
        already in flight (_grads_done); whatever is left goes in one collective; PPYOLO_HIP_TRAIN_OVERLAP=0: one collective
        for everything, after the backward."""
        if self.world > 1:
            if self._works:
                rest = [r for b, v in self._buckets.items() if b not in self._reduced for r in v['ranges']]
                for a, e in rest:
                    self._works.append(torch.distributed.all_reduce(self.gflat[a:e], async_op=True))
                for w in self._works:
                    w.wait()
            else:
                # not overlapped: the collective is ordered behind the backward's kernels on the current stream, and the SGD
                # launches behind it -- RCCL's reduction never runs beside this rank's MFMA kernels
                torch.distributed.all_reduce(self.gflat)
            self._works, self._reduced, self._pending = [], [], {}
            self.gflat.mul_(1.0 / self.world)

    def sgd(self, lr):
        """optimizer.step() of train.py:442 with the reference's parameter groups (custom_layers.py:167-215: weight decay on
        convolution weights, none on biases and BatchNorm scales / offsets; every head layer has lr multiplier 1), then
        ema.update() (train.py:443-444, model/EMA.py:29-44)."""
        first = self.steps_done == 0
        nd, n = self.n_decay, self.pflat.numel()

        def tail():
            K.sgd_momentum(self.pflat[:nd], self.gflat[:nd], self.vflat[:nd], lr, self.momentum, self.weight_decay, first)
            if n > nd:
                K.sgd_momentum(self.pflat[nd:], self.gflat[nd:], self.vflat[nd:], lr, self.momentum, 0.0, first)
            if self.use_ema:
                K.ema_update(self.sflat, self.pflat, self.ema_steps, self.ema_decay)
        if self._async_tail:
            self._await_params()
            self._wstream.wait_stream(torch.cuda.current_stream(self.dev))      # the gradients are complete (and averaged)
            with torch.cuda.stream(self._wstream):
                tail()
            self._params_pending = True
        else:
            tail()
        self.steps_done += 1
        if self.use_ema:
            self.ema_steps += 1

    def step(self, x_nchw, gt_box, targets, lr, dropblock_masks=None, next_x=None):
        """One iteration of the reference's loop (train.py:416-443).  next_x: the NEXT iteration's images, if the loader already has
        them (round 5): with the backbone frozen their backbone forward runs beside this iteration's head, loss and backward
        (prefetch_backbone) and the next step(next_x, ...) finds it done -- results are those of the plain loop, bit for bit."""
        if self.external:
            raise PPYoloHipError('external_optimizer: the caller owns the update (loss.backward(); optimizer.step())')
        ready = None
        if next_x is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.dev))      # next_x is ready here, before this step's kernels are issued
        loss6 = self.forward_backward(x_nchw, gt_box, targets, dropblock_masks)
        if next_x is not None:
            # the early event only when this step's backbone was served from the prefetch: otherwise the main stream has just run a
            # backbone itself (step 0, or a key mismatch) and the side stream waits for THAT (round-5 advisor, medium)
            self.prefetch_backbone(next_x, ready if self._own_backbone_done is None else self._own_backbone_done)
        self.all_reduce()
        self.sgd(lr)
        return loss6

    # ---- views for tests / checkpoints --------------------------------------------------------------------------------------
    def grads(self, flat=None):
        """Gradients in the state_dict's own layouts ([K, C, R, S] convolution weights) -- of the last forward_backward, or
        read from `flat`, a copy of the flat gradient buffer taken after an earlier one (then the results are views of it
        wherever no re-layout is needed)."""
        out = {}
        if self.gflat is None:
            raise PPYoloHipError('grads(): no step has run yet (the flat gradient buffer is allocated by the first one)')
        base = self.gflat.data_ptr()
        for k in self.train_keys:
            g = self.G[k]
            if flat is not None:
                o = (g.data_ptr() - base) // 4
                g = flat[o:o + g.numel()].view(g.shape)
            if k in self._wcache:
                out[k] = g[..., :self._wcache[k]['Cin']].permute(0, 3, 1, 2).contiguous()
            else:
                out[k] = g if flat is not None else g.clone()
        return out

    def sync_to_model(self, ema=False):
        """Write the trained parameters (kept flat, in kernel layout, during training) back into the module; `ema=True` writes
        the EMA shadows instead -- what the reference evaluates and saves after ema.apply() (train.py:476-500)."""
        self._await_params()
        if self._bstream is not None:       # a prefetched backbone forward may still be writing BatchNorm running statistics
            torch.cuda.current_stream(self.dev).wait_stream(self._bstream)
        src = self.sflat if ema else self.pflat
        if ema and src is None:
            raise PPYoloHipError('EMA is off (cfg.use_ema)')
        if self.gflat is not None:
            base = self.pflat.data_ptr()
            for k in self.train_keys:
                v = self.P[k]
                o = (v.data_ptr() - base) // 4
                t = src[o:o + v.numel()].view(v.shape)
                if k in self._wcache:
                    self.sd[k].copy_(t[..., :self._wcache[k]['Cin']].permute(0, 3, 1, 2))
                else:
                    self.sd[k].copy_(t)
        if hasattr(self.model, '_plans'):
            self.model._plans.clear()


class _FinishedBackward(torch.autograd.Function):
    """The loss terms as autograd leaves-with-history: by the time forward() returns, the HIP kernels have already produced
    d(sum of the terms)/d(parameter) for every trainable tensor; backward() hands those to autograd (scaled by the upstream
    gradient, which must be the same for every term -- the reference sums them, train.py:429-434)."""

    @staticmethod
    def forward(ctx, ts, loss6, n_terms, *params):
        ctx.ts = ts
        # the gradients of THIS forward: the step's flat buffer is overwritten by the next training forward (gradient
        # accumulation `(model(a) + model(b)).backward()`, a validation-loss forward before the backward), so keep a copy --
        # one device-to-device move of 92.6 MB (R50vd head), ~40 us
        ctx.gsnap = ts.gflat.clone()
        return tuple(loss6[i].clone() for i in range(n_terms))

    @staticmethod
    def backward(ctx, *gs):
        ts = ctx.ts
        g = [float(v) for v in gs if v is not None]
        if not g or any(v != g[0] for v in g) or len(g) != len(gs):
            raise PPYoloHipError('the HIP training step differentiates the SUM of the loss terms (reference train.py:429-440); '
                                 'got upstream gradients %s' % g)
        grads = ts.grads(ctx.gsnap)
        ctx.gsnap = None
        out = tuple(grads[k] if g[0] == 1.0 else grads[k] * g[0] for k in ts.train_keys)
        return (None, None, None) + out


LOSS_NAMES = ('loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou', 'loss_iou_aware')      # reference model/losses.py:231-240


def loss_dict(model, images, gt_box, targets):
    """`PPYOLO.forward(images, None, False, gt_box, gt_label, gt_score, targets)` of the reference (model/ppyolo.py:19-25):
    {name: scalar tensor}, whose sum the caller back-propagates and steps with its own torch optimizer.  Training forward
    (BatchNorm on batch statistics everywhere, DropBlock), loss and the backward through the head run here, as HIP
    kernels; gt_label / gt_score do not enter the reference's loss either (model/losses.py:113-117)."""
    ts = getattr(model, '_train_bridge', None)
    if ts is not None and (ts.dev != next(model.parameters()).device or ts.freeze_at != int(model.backbone.freeze_at)):
        ts = None                     # moved / re-frozen since: the cached kernel-layout weights describe another model
    if ts is None:
        ts = TrainStep(model, None, external_optimizer=True)
        object.__setattr__(model, '_train_bridge', ts)
    else:                             # head.set_dropblock(is_test=...) may have been called since (eval between iterations)
        ts.cfg.head['drop_active'] = ModelSettings.drop_active(model.head)
    if isinstance(targets, (list, tuple)):
        targets = [t if torch.is_tensor(t) else torch.as_tensor(t) for t in targets]
    dev = ts.dev
    images = images if torch.is_tensor(images) else torch.as_tensor(images)
    gt_box = gt_box if torch.is_tensor(gt_box) else torch.as_tensor(gt_box)
    loss6 = ts.forward_backward(images.to(dev), gt_box.to(dev), [t.to(dev) for t in targets])
    n = 6 if ts.cfg.head.get('iou_aware', False) else 5
    params = [p for k, p in model.named_parameters() if k in set(ts.train_keys)]
    with torch.enable_grad():
        terms = _FinishedBackward.apply(ts, loss6, n, *params)
    return {LOSS_NAMES[i]: terms[i] for i in range(n)}


def lr_at(iter_id, cfg):
    """calc_lr of the reference (train.py:172-188): linear warm-up, then piecewise decay."""
    lrc = cfg.learningRate
    base, gamma, miles = lrc['base_lr'], lrc['PiecewiseDecay']['gamma'], lrc['PiecewiseDecay']['milestones']
    for i in range(len(miles), 0, -1):
        if iter_id >= miles[i - 1]:
            return base * gamma ** i
    steps, start = lrc['LinearWarmup']['steps'], lrc['LinearWarmup']['start_factor']
    if iter_id <= steps:
        return base * (start + (1.0 - start) / steps * iter_id)
    return base
