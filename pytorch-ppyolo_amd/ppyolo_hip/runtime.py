"""Model-level glue: module tree -> Plan -> HipExecutor, cached per input shape."""
import os

import torch

from .engine import Builder, HipExecutor
from ._lib import PPYoloHipError


def build_plan(model, N, H, W, device, with_head=True, skeleton=False):
    """Walk `model` (a model.ppyolo.PPYOLO) into a kernel plan for inputs [N,3,H,W].  `skeleton`: shapes only, parameter
    data untouched (engine.Builder)."""
    b = Builder(N, H, W, device, skeleton=skeleton)
    head = model.head if with_head else None
    slots = head.make_out_slots(model.backbone.feature_maps) if head is not None else None
    feats = model.backbone.emit(b, slots)
    b.plan.feats = feats
    if head is not None:
        outs = head.emit(b, feats)
        b.plan.head_outs = outs
        b.plan.decode = head.decode_params(outs)
    return b.plan


class PlanCache(object):
    """Plans (and their device buffers / hipGraphs) keyed by (N, H, W, device, lane)."""

    def __init__(self, model):
        self._model = model
        self._ex = {}
        self.blob = None               # weight owner loaded from a native blob (ppyolo_hip/blob.py), if any
        self.generation = 0
        self._sig = None               # parameter signature the executors' folded weights were derived from
        self._tensors = None
        self._dirty = False            # set by writers that bypass autograd's version counters (mark_dirty)
        # the check costs ~0.14 ms of host time per forward (468 tensors): a server whose weights are final may switch it off
        self.pinned = os.environ.get('PPYOLO_HIP_PIN_WEIGHTS', '0') == '1'
        self.use_graph = os.environ.get('PPYOLO_HIP_GRAPH', '1') != '0'
        # plain forward() of an even batch >= 4 as two half-batch lanes on two streams (round 4 experiment, see run_split)
        self.split_forward = os.environ.get('PPYOLO_HIP_FORWARD_SPLIT', '0') == '1'
        self._split = {}
        self.autotune = os.environ.get('PPYOLO_HIP_AUTOTUNE', '0') == '1'

    def clear(self):
        self._ex = {}
        self._split = {}
        self.blob = None
        self._sig = None
        self._tensors = None
        self.generation += 1          # InFlight lanes built from older executors (older weights) are stale

    def _signature(self):
        """Changes whenever a parameter or buffer is written in place (optimizer.step(), EMA.apply() / restore() through
        `p.copy_`, BatchNorm statistics of a training forward) or rebound (`p.data = ...`, as the reference's EMA does,
        model/EMA.py:46-58): the tensors' autograd version counters and storage addresses."""
        ts = self._tensors
        if ts is None:          # (walking the module tree costs ~1 ms: once per clear(); rebinding `.data` keeps the Parameter objects)
            m = self._model
            ts = self._tensors = [t for t in list(m.parameters()) + list(m.buffers()) if t.device.type != 'meta']
        # order-sensitive (the caching allocator hands freed addresses to OTHER parameters across EMA.apply() / restore(), and a
        # sum / XOR of the addresses is blind to such a permutation)
        return hash(tuple([(t._version, t.data_ptr()) for t in ts]))

    def mark_dirty(self):
        """Explicit invalidation for writers the signature cannot see: the HIP training kernels update BatchNorm running
        statistics through raw pointers (no autograd version bump), and TrainStep re-binds the num_batches_tracked buffers."""
        self._dirty = True

    def check_current(self):
        """The executors hold COPIES of the folded / re-laid / split weights: drop them when the parameters they were derived
        from have been modified since (the reference's loop trains, then evaluates with `ema.apply(); eval; ema.restore()`
        every eval_iter iterations, train.py:481-499 -- each evaluation must see the weights of ITS iteration).  A loaded
        native blob is such a copy too."""
        if self.pinned:
            return
        if self._dirty:
            self._dirty = False
            self._tensors = None           # buffers may have been re-bound
            if self._ex or self.blob is not None:
                self.clear()
        sig = self._signature()
        if self._sig is not None and sig != self._sig and (self._ex or self.blob is not None):
            self.clear()
        self._sig = sig

    def executor(self, x, lane=0, multi_stream=None):
        """Lane 0 belongs to `PPYOLO.forward` / `forward_padded`; lanes > 0 are further executors of the same shape (own
        activations, scratch and graph) that InFlight uses to keep several batches on the device at once -- InFlight never
        touches lane 0, so a plain `model(x)` call cannot overwrite a batch that is still in flight.  Whichever executor
        of a shape is built first holds the folded / split weights; the others share them (read-only)."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise PPYoloHipError('PPYOLO.forward needs a ROCm device tensor [N,3,H,W]; the MI355X path has no '
                                 'CPU fallback (got %s)' % (getattr(x, 'device', type(x)),))
        if self._model.training:
            raise PPYoloHipError('call model.eval() first: only the inference path is implemented '
                                 '(reference demo.py:92)')
        N, C, H, W = x.shape
        if C != 3:
            raise PPYoloHipError('expected NCHW input with 3 channels')
        self.check_current()
        key = (N, H, W, str(x.device), lane)
        ex = self._ex.get(key)
        if ex is None:
            from .engine import math_mode
            with torch.no_grad():
                # the folded / split weights are the same for every input shape and lane: the first executor on a device
                # (or a loaded blob) owns them, all others are built from a shape-only plan and share its tensors
                owner = next((e for k, e in self._ex.items() if k[3] == key[3] and e.math == math_mode()), None)
                if owner is None and self.blob is not None and self.blob.math == math_mode() \
                        and str(self.blob.device) == str(x.device):
                    owner = self.blob
                plan = build_plan(self._model, N, H, W, x.device, skeleton=owner is not None)
                ex = HipExecutor(plan, x.device, use_graph=self.use_graph, multi_stream=multi_stream, share=owner)
                if self.autotune:
                    ex.run() if not self.use_graph else ex._launch_all()
                    ex.autotune()
            self._ex[key] = ex
        return ex

    def run_split(self, x, im_size):
        """`forward` of one batch as TWO half-batch lanes (own executor, hipGraph and stream each) joined on the caller's
        stream: the second half's wide layers fill the CUs the first half's narrow tail leaves idle -- what InFlight does
        across batches, inside one call.  -> an object with out_dets / out_count / out_keep of the whole batch (the two
        executors write their halves of ONE set of result tensors)."""
        N = x.shape[0]
        h = N // 2
        key = (tuple(x.shape), str(x.device))
        sp = self._split.get(key)
        self.check_current()
        if sp is None or sp.generation != self.generation:
            class _Split(object):
                pass
            sp = _Split()
            sp.generation = self.generation
            exs = [self.executor(x[:h], lane=('split', k), multi_stream=False) for k in range(2)]
            kk = exs[0].out_dets.shape[1]
            sp.out_dets = torch.zeros((N, kk, 6), dtype=torch.float32, device=x.device)
            sp.out_count = torch.zeros((N,), dtype=torch.int32, device=x.device)
            sp.out_keep = torch.zeros((N, kk), dtype=torch.int32, device=x.device)
            for k, ex in enumerate(exs):      # (before the first run: the graphs capture these addresses)
                ex.out_dets, ex.out_count, ex.out_keep = sp.out_dets[k * h:(k + 1) * h], sp.out_count[k * h:(k + 1) * h], sp.out_keep[k * h:(k + 1) * h]
                ex.invalidate_graph()
            sp.lanes = [(ex, torch.cuda.Stream(device=x.device)) for ex in exs]
            self._split[key] = sp
        cur = torch.cuda.current_stream(x.device)
        for k, (ex, st) in enumerate(sp.lanes):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ex.set_inputs(x[k * h:(k + 1) * h], im_size[k * h:(k + 1) * h])
                ex.run()
        for _, st in sp.lanes:
            cur.wait_stream(st)
        x.record_stream(sp.lanes[0][1])
        x.record_stream(sp.lanes[1][1])
        return sp

    @staticmethod
    def unpack(ex):
        """Reference result convention (model/matrix_nms.py:113, :134, :147-151): per image a
        [K,6] tensor, or [[-1]*6] when nothing survives."""
        counts = ex.out_count.cpu().tolist()        # the one host sync of the API
        out = []
        for i, k in enumerate(counts):
            out.append(ex.out_dets[i, :max(k, 1)].clone())
        return out


class Ticket(object):
    """One submitted batch.  `result()` blocks until it is done and returns the reference's list of [K,6] tensors;
    `padded()` returns the device-resident (dets, count, keep_idx) of the lane -- valid until the lane is reused,
    i.e. until `depth` more batches have been submitted."""

    def __init__(self, lane):
        self._lane = lane
        self._open = True

    def padded(self):
        ex = self._lane.ex
        self._lane.done.synchronize()
        return ex.out_dets, ex.out_count, ex.out_keep

    def result(self):
        if not self._open:
            raise PPYoloHipError('ticket already collected (its lane may hold a newer batch)')
        self._lane.done.synchronize()
        out = PlanCache.unpack(self._lane.ex)
        self._open = False
        self._lane.ticket = None
        return out


def lane_cu_masks(spec, depth, ncu):
    """Host logic of the lanes' CU partition (no device): `spec` = one term per lane, '|'-separated, a term = 'm<mod>:<lo>-<hi>'
    (the CUs whose mask bit i has lo <= i % mod <= hi), 'h:<hex>,<hex>,...' (raw 32-bit mask words) or 'all'.  On MI355X bit i of a stream's CU mask is CU i // 8 of XCD
    i % 8 (tools/probes/cu_mask_probe.hip), so 'm8:0-3|m8:4-7' gives each of two lanes four whole XCDs, 'm16:0-7|m16:8-15' half
    of the CUs of every XCD.  Named forms: 'xcd' = whole XCDs dealt out evenly over the lanes, 'half' = an equal share of every
    XCD's CUs.  -> list of `depth` masks, each a list of 32-bit words (None = no mask); '' / '0' / 'off' -> all None."""
    if not spec or spec in ('0', 'off', 'none'):
        return [None] * depth
    if depth < 1 or ncu < 1:
        raise PPYoloHipError('lane_cu_masks: depth and ncu must be positive')
    if spec == 'xcd':
        if 8 % depth:
            raise PPYoloHipError('PPYOLO_HIP_LANE_CUS=xcd needs a lane count that divides 8 (got %d)' % depth)
        per = 8 // depth
        spec = '|'.join('m8:%d-%d' % (k * per, (k + 1) * per - 1) for k in range(depth))
    elif spec == 'half':
        per_xcd = max(1, ncu // 8)
        if per_xcd % depth:
            raise PPYoloHipError('PPYOLO_HIP_LANE_CUS=half: %d CUs per XCD do not divide over %d lanes' % (per_xcd, depth))
        per = per_xcd // depth
        spec = '|'.join('m%d:%d-%d' % (8 * per_xcd, 8 * k * per, 8 * (k + 1) * per - 1) for k in range(depth))
    terms = spec.split('|')
    if len(terms) != depth:
        raise PPYoloHipError('PPYOLO_HIP_LANE_CUS=%r names %d lanes, InFlight has %d' % (spec, len(terms), depth))
    masks = []
    for t in terms:
        if t == 'all':
            masks.append(None)
            continue
        if t.startswith('h:'):          # raw 32-bit words, hexadecimal, lowest CUs first
            try:
                words = [int(w, 16) & 0xffffffff for w in t[2:].split(',')]
            except ValueError:
                raise PPYoloHipError('PPYOLO_HIP_LANE_CUS: cannot read term %r' % t)
            if not any(words):
                raise PPYoloHipError('PPYOLO_HIP_LANE_CUS: term %r enables no CU' % t)
            masks.append(words)
            continue
        try:
            mod, rng = t[1:].split(':')
            lo, hi = rng.split('-')
            mod, lo, hi = int(mod), int(lo), int(hi)
            assert t[0] == 'm' and mod > 0 and 0 <= lo <= hi < mod
        except (ValueError, AssertionError):
            raise PPYoloHipError('PPYOLO_HIP_LANE_CUS: cannot read term %r (want m<mod>:<lo>-<hi> or all)' % t)
        words = [0] * ((ncu + 31) // 32)
        for i in range(ncu):
            if lo <= i % mod <= hi:
                words[i // 32] |= 1 << (i % 32)
        if not any(words):
            raise PPYoloHipError('PPYOLO_HIP_LANE_CUS: term %r enables no CU of %d' % (t, ncu))
        masks.append(words)
    return masks


class _MaskedStream(object):
    """A lane's HIP stream restricted to part of the chip (ppy_lane_stream_create), wrapped for torch."""

    def __init__(self, device, words):
        import ctypes
        from ._lib import lib, check
        self._lib, self._check = lib, check
        ptr = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * len(words))(*words)
        with torch.cuda.device(device):
            check(lib().ppy_lane_stream_create(ctypes.byref(ptr), arr, len(words)), 'ppy_lane_stream_create')
        self.ptr = ptr.value
        self.words = list(words)
        self.stream = torch.cuda.ExternalStream(self.ptr, device=device)
        # the HIP stream lives as long as the torch object that names it (callers keep `lane.stream`, not the lane: bench.py)
        self.stream._ppy_owner = self

    def __del__(self):
        try:
            if self.ptr:
                self._lib().ppy_lane_stream_destroy(self.ptr)
                self.ptr = None
        except Exception:      # interpreter shutdown
            pass


class _Lane(object):
    def __init__(self, ex, device, cu_mask=None, priority=0):
        self.ex = ex
        self._masked = None
        if cu_mask is None:
            self.stream = torch.cuda.Stream(device=device, priority=priority)
        else:
            self._masked = _MaskedStream(device, cu_mask)
            self.stream = self._masked.stream
        self.cu_mask = cu_mask
        self.done = torch.cuda.Event()
        self.ticket = None


class InFlight(object):
    """Keep `depth` batches on the device at once: lane k has its own executor (activations, scratch, hipGraph) and its
    own stream, and consecutive `submit`s go round-robin over the lanes, so the head of batch i+1 fills the CUs that
    the narrow tail layers of batch i (19x19 maps, decode, Matrix-NMS) leave idle.  Measured on MI355X, 8 images
    per batch: R50vd-608 1460 -> 1818 img/s, R18vd-416 7350 -> 10380 img/s at depth 2 (depth 3 is slower again).
    Results are those of `model.forward` bit for bit (tests/test_gpu_model.py::test_in_flight_matches_forward).

        pipe = InFlight(model, depth=2)
        t0 = pipe.submit(x0, im0); t1 = pipe.submit(x1, im1)
        preds0 = t0.result(); t2 = pipe.submit(x2, im2); ...
    """

    def __init__(self, model, depth=2, cu_masks=None):
        if depth < 1:
            raise PPYoloHipError('InFlight depth must be >= 1')
        self._model = model
        self.depth = depth
        self._lanes = {}
        self._generation = model._plans.generation
        self._next = 0
        # CU partition of the lanes (round 5): PPYOLO_HIP_LANE_CUS, see lane_cu_masks
        self.cu_spec = os.environ.get('PPYOLO_HIP_LANE_CUS', '') if cu_masks is None else cu_masks
        self._masks = None

    def _lane(self, x, k):
        self._model._plans.check_current()                           # parameters modified in place since the lanes were built
        if self._generation != self._model._plans.generation:      # load_state_dict / .to(): weights changed
            if any(lane.ticket is not None for lane in self._lanes.values()):
                raise PPYoloHipError('the model changed while batches were in flight: collect their tickets first')
            self._lanes = {}
            self._generation = self._model._plans.generation
        key = (tuple(x.shape), str(x.device), k)
        lane = self._lanes.get(key)
        if lane is None:
            # one executor alone keeps its forked graph; executors that overlap each other run single-branch graphs
            # lanes 1..depth: lane 0 is the executor of model.forward, which must stay free to be called beside open tickets
            ex = self._model._plans.executor(x, lane=k + 1, multi_stream=None if self.depth == 1 else False)
            if self._masks is None:
                ncu = torch.cuda.get_device_properties(x.device).multi_processor_count
                self._masks = lane_cu_masks(self.cu_spec, self.depth, ncu)
            # PPYOLO_HIP_LANE_PRIORITY="-1,0": HIP stream priorities of the lanes (lower = dispatched first); an experiment of round 5
            pr = [int(v) for v in os.environ.get('PPYOLO_HIP_LANE_PRIORITY', '').split(',') if v.strip()]
            lane = _Lane(ex, x.device, self._masks[k], pr[k] if k < len(pr) else 0)
            self._lanes[key] = lane
        return lane

    def submit(self, x, im_size):
        """Enqueue one batch (device tensors, as for `forward`); returns a Ticket at once."""
        k = self._next
        lane = self._lane(x, k)
        if lane.ticket is not None:
            raise PPYoloHipError('all %d lanes hold uncollected batches: call result() on the oldest ticket first'
                                 % self.depth)
        self._next = (k + 1) % self.depth
        lane.stream.wait_stream(torch.cuda.current_stream(x.device))        # x / im_size are ready on the caller's stream
        with torch.cuda.stream(lane.stream):
            lane.ex.set_inputs(x, im_size)
            lane.ex.run()
            lane.done.record(lane.stream)
        x.record_stream(lane.stream)
        lane.ticket = Ticket(lane)
        return lane.ticket

    def lanes(self, x):
        """The (executor, stream) pairs for this input shape -- for callers that keep their inputs resident in the
        executors and drive the replay themselves (bench.py).  These executors are InFlight's own (never the one
        `model.forward` uses)."""
        return [(self._lane(x, k).ex, self._lane(x, k).stream) for k in range(self.depth)]


def run_backbone(backbone, x):
    """Backbone.forward on an NCHW device tensor -> list of NCHW feature maps (API glue for
    the per-stage parity tests; the fused model path never converts layouts)."""
    if not x.is_cuda:
        raise PPYoloHipError('HIP path needs a ROCm device tensor; there is no CPU fallback')
    N, C, H, W = x.shape
    b = Builder(N, H, W, x.device)
    with torch.no_grad():
        feats = backbone.emit(b, None)
    ex = HipExecutor(b.plan, x.device, use_graph=False)
    ex.set_inputs(x.float())
    ex.run()
    return [ex.view(f).dense().permute(0, 3, 1, 2).contiguous() for f in feats]
