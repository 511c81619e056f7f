"""Model-level glue: module tree -> Plan -> HipExecutor, cached per input shape."""
import os

import torch

from .engine import Builder, HipExecutor
from ._lib import PPYoloHipError


def build_plan(model, N, H, W, device, with_head=True):
    """Walk `model` (a model.ppyolo.PPYOLO) into a kernel plan for inputs [N,3,H,W]."""
    b = Builder(N, H, W, device)
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
    """Plans (and their device buffers / hipGraphs) keyed by (N, H, W, device)."""

    def __init__(self, model):
        self._model = model
        self._ex = {}
        self.use_graph = os.environ.get('PPYOLO_HIP_GRAPH', '1') != '0'
        self.autotune = os.environ.get('PPYOLO_HIP_AUTOTUNE', '0') == '1'

    def clear(self):
        self._ex = {}

    def executor(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise PPYoloHipError('PPYOLO.forward needs a ROCm device tensor [N,3,H,W]; the MI355X path has no '
                                 'CPU fallback (got %s)' % (getattr(x, 'device', type(x)),))
        if self._model.training:
            raise PPYoloHipError('call model.eval() first: only the inference path is implemented '
                                 '(reference demo.py:92)')
        N, C, H, W = x.shape
        if C != 3:
            raise PPYoloHipError('expected NCHW input with 3 channels')
        key = (N, H, W, str(x.device))
        ex = self._ex.get(key)
        if ex is None:
            with torch.no_grad():
                plan = build_plan(self._model, N, H, W, x.device)
                ex = HipExecutor(plan, x.device, use_graph=self.use_graph)
                if self.autotune:
                    ex.run() if not self.use_graph else ex._launch_all()
                    ex.autotune()
            self._ex[key] = ex
        return ex

    @staticmethod
    def unpack(ex):
        """Reference result convention (model/matrix_nms.py:113, :134, :147-151): per image a
        [K,6] tensor, or [[-1]*6] when nothing survives."""
        counts = ex.out_count.cpu().tolist()        # the one host sync of the API
        out = []
        for i, k in enumerate(counts):
            out.append(ex.out_dets[i, :max(k, 1)].clone())
        return out


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
