"""`PPYOLO` -- drop-in for the reference's `model.ppyolo.PPYOLO` (model/ppyolo.py:14-29).

Same constructor and `forward(x, im_size, eval=True, ...)` signature and return type (a list
with one `[K, 6]` tensor per image: label, score, x0, y0, x1, y1; `[[-1]*6]` when nothing is
detected).  The forward runs the MI355X plan (hand-written HIP kernels behind the C ABI);
torch tensors are storage only.  There is no CPU fallback: CPU inputs raise.
"""
import torch

from ppyolo_hip.runtime import InFlight, PlanCache


class PPYOLO(torch.nn.Module):
    def __init__(self, backbone, head):
        super(PPYOLO, self).__init__()
        self.backbone = backbone
        self.head = head
        self._plans = PlanCache(self)

    def forward(self, x, im_size, eval=True, gt_box=None, gt_label=None, gt_score=None, targets=None):
        if not eval:
            # the reference's training call (train.py:428): a dict of loss terms whose sum can be .backward()ed -- forward, loss
            # and backward all ran as HIP kernels by the time this returns; backward() hands the finished gradients to autograd
            from ppyolo_hip.train import loss_dict
            return loss_dict(self, x, gt_box, targets)
        ex = self._run(x, im_size)
        return self._plans.unpack(ex)

    def _run(self, x, im_size):
        """One forward on the current stream -> the object holding out_dets / out_count / out_keep."""
        if self._plans.split_forward and x.shape[0] >= 4 and x.shape[0] % 2 == 0:
            return self._plans.run_split(x, im_size)
        ex = self._plans.executor(x)
        ex.set_inputs(x, im_size)
        ex.run()
        return ex

    def add_param_group(self, param_groups, base_lr, base_wd):      # reference model/ppyolo.py:27-29
        self.backbone.add_param_group(param_groups, base_lr, base_wd)
        self.head.add_param_group(param_groups, base_lr, base_wd)

    def forward_padded(self, x, im_size):
        """Device-resident result without the host sync `forward` needs to build its
        variable-length list: (dets [N,keep_top_k,6] padded with -1, count [N] int32,
        keep_idx [N,keep_top_k] int32 = box*num_classes + class)."""
        ex = self._run(x, im_size)
        return ex.out_dets, ex.out_count, ex.out_keep

    def in_flight(self, depth=2, cu_masks=None):
        """Throughput mode (not in the reference): a submit/collect pipeline that keeps `depth` batches on the device
        at once -- see ppyolo_hip.runtime.InFlight.  cu_masks: how the lanes share the chip's CUs (default: the
        PPYOLO_HIP_LANE_CUS environment switch; runtime.lane_cu_masks)."""
        return InFlight(self, depth, cu_masks)

    # ---- native weight blob (ppyolo_hip/blob.py) ----
    def save_native_blob(self, path):
        """Write the folded / re-laid / pre-split weights of the CURRENT parameters, in the current PPYOLO_HIP_MATH mode,
        to `path` (the model must be on a ROCm device, in eval mode).  Returns the file size."""
        from ppyolo_hip import blob
        dev = next(self.parameters()).device
        ex = next(iter(self._plans._ex.values()), None)
        if ex is None:      # any input shape yields the same weights: a small one
            ex = self._plans.executor(torch.zeros((1, 3, 64, 64), dtype=torch.float32, device=dev))
        return blob.save(ex, path, blob.fingerprint(self.state_dict()))

    def load_native_blob(self, path, verify=True):
        """After `load_state_dict` (the blob is a cache of what is derived from the checkpoint, checked against the
        parameters the model holds unless verify=False): executors built from now on upload nothing but this file's data
        region and build their plans shape-only."""
        from ppyolo_hip import blob
        dev = next(self.parameters()).device
        if dev.type == 'meta':
            raise RuntimeError('a model built on the meta device has no parameters to verify against: use attach_native_blob')
        own = blob.load(path, dev, blob.fingerprint(self.state_dict()) if verify else None)
        self._plans.clear()
        self._plans.blob = own
        return own.nbytes

    def attach_native_blob(self, path, device='cuda'):
        """Inference-only start WITHOUT the checkpoint: a model whose modules were constructed under
        `torch.device('meta')` (shapes only, no initialisation, no .pt read) takes all its weights from the blob.  Its
        `state_dict()` holds no data -- use `load_state_dict` + `load_native_blob` when the parameters themselves are needed."""
        from ppyolo_hip import blob
        own = blob.load(path, device, None)
        self._plans.clear()
        self._plans.blob = own
        return own.nbytes

    # any change of parameters / device invalidates the folded weights held by the plans
    # (in-place updates -- optimizer.step(), EMA.apply() / restore(), a training forward's BatchNorm statistics -- are noticed
    # by PlanCache.check_current() at the next forward)
    def load_state_dict(self, *a, **k):
        r = super(PPYOLO, self).load_state_dict(*a, **k)
        self._plans.clear()
        self.__dict__.pop('_train_bridge', None)      # its kernel-layout copies of the FROZEN weights were built from the old ones
        return r

    def _apply(self, fn, *a, **k):
        r = super(PPYOLO, self)._apply(fn, *a, **k)
        if hasattr(self, '_plans'):
            self._plans.clear()
        self.__dict__.pop('_train_bridge', None)
        return r

    def pin_weights(self, on=True):
        """Serving: the weights are final -- skip the per-forward check for modified parameters (~0.14 ms of host time)."""
        self._plans.pinned = bool(on)
