import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pytorch-ppyolo_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # a native crash (GPU memory fault -> abort) leaves its Python stack in gpurun_out/ instead of vanishing
    try:
        import faulthandler
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        config._ppy_fault_log = open(os.path.join(d, 'pytest_fault.log'), 'w')
        faulthandler.enable(file=config._ppy_fault_log, all_threads=True)
    except Exception:
        pass
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    def _load(name):
        return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    return _load


def build_model(cfg, seed=0, device='cpu'):
    """This package's PPYOLO with the deterministic synthetic weights (same tensors the
    goldens were generated with)."""
    import torch
    from config import select_backbone, select_head
    from model.ppyolo import PPYOLO
    from ppyolo_hip import synth
    bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
    hd = select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed)
    m.load_state_dict(sd, strict=True)
    m.eval()
    hd.set_dropblock(is_test=True)
    if device != 'cpu':
        m = m.to(device)
    return m, sd


def build_train_model(cfg, seed=0, device='cpu'):
    """The model as the reference's train.py:238-264 builds it -- loss settings, YOLOv3Head(is_train=True),
    backbone.freeze() -- with the deterministic synthetic weights; left in nn.Module's training mode."""
    from config import select_backbone, select_head, select_loss
    from model.ppyolo import PPYOLO
    from ppyolo_hip import synth
    bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
    iou_loss = select_loss(cfg.iou_loss_type)(**cfg.iou_loss)
    iou_aware = select_loss(cfg.iou_aware_loss_type)(**cfg.iou_aware_loss) if cfg.head['iou_aware'] else None
    yolo_loss = select_loss(cfg.yolo_loss_type)(iou_loss=iou_loss, iou_aware_loss=iou_aware, **cfg.yolo_loss)
    hd = select_head(cfg.head_type)(yolo_loss=yolo_loss, is_train=True, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, seed=seed), strict=True)
    bb.freeze()
    return m.to(device) if device != 'cpu' else m


@pytest.fixture(scope='session')
def model_shapes():
    def _shapes(cfg):
        from config import select_backbone, select_head
        from model.ppyolo import PPYOLO
        bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
        hd = select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
        return {k: tuple(v.shape) for k, v in PPYOLO(bb, hd).state_dict().items()}
    return _shapes


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests():
    """Executors of a finished test (captured hipGraphs, their memory pools, side streams) are destroyed HERE, with the device
    idle -- not whenever the cyclic garbage collector happens to run inside a later test's launch loop (round 4: the collector
    fired in the middle of an autotune sweep and the runtime aborted in a graph destructor; the same test passes alone)."""
    yield
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()


@pytest.fixture(scope='session', autouse=True)
def _drain_gpu_at_exit():
    """Captured hipGraphs and their memory pools are released while the HIP runtime is still alive (an interpreter
    that tears the runtime down first can die in the graphs' destructors)."""
    yield
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
