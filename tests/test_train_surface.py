"""The training-loop surface of the reference (train.py:238-286, :416-444) -- loss settings objects, backbone.freeze(),
model.add_param_group, forward(eval=False) -- host logic, no GPU: the optimizer groups against the reference's own
(golden g14), and loud failures where the HIP path cannot run."""
import numpy as np
import pytest
import torch

from conftest import build_train_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config, select_loss, select_optimizer


@pytest.mark.parametrize('tag,cfgc', [('r18vd', PPYOLO_r18vd_Config), ('r50vd', PPYOLO_2x_Config)])
def test_param_groups_match_the_reference(golden, tag, cfgc):
    g = golden('g14_train_loop')
    cfg = cfgc()
    m = build_train_model(cfg)
    assert m.training and not m.head.detection_blocks[0].layers[-3 if tag == 'r50vd' else 0].is_test      # DropBlock draws
    groups = []
    m.add_param_group(groups, cfg.learningRate['base_lr'], cfg.optimizerBuilder['regularizer']['factor'])
    names = {id(q): k for k, q in m.named_parameters()}
    assert [names[id(gr['params'][0])] for gr in groups] == [str(v) for v in g[tag + '.group_names']]
    got = np.array([[gr['lr'], gr['base_lr'], gr['weight_decay']] for gr in groups], np.float64)
    assert np.array_equal(got, g[tag + '.group_lr_wd'])
    # the frozen backbone contributes nothing; every head tensor is in exactly one group
    assert all(k.startswith('head.') for k in g[tag + '.group_names'])
    assert len(groups) == sum(1 for k, q in m.named_parameters() if k.startswith('head.')) == sum(q.requires_grad for q in m.parameters())
    opt = select_optimizer(cfg.optimizerBuilder['optimizer']['type'])(groups, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    assert isinstance(opt, torch.optim.SGD) and len(opt.param_groups) == len(groups)


def test_freeze_follows_freeze_at():
    from config import select_backbone
    cfg = PPYOLO_2x_Config()
    for fa in (0, 2, 5):
        bb = select_backbone(cfg.backbone_type)(**dict(cfg.backbone, freeze_at=fa))
        bb.freeze()
        frozen = {k.split('.')[0] for k, q in bb.named_parameters() if not q.requires_grad}
        stages = {int(k[5]) for k in frozen}
        assert stages == set(range(1, fa + 1)), (fa, stages)
        assert all(q.requires_grad == (int(k[5]) > fa) for k, q in bb.named_parameters())


def test_unsupported_settings_are_refused():
    with pytest.raises(NotImplementedError):
        select_loss('IouLoss')(ciou_term=True)
    assert select_loss('IouLoss')(loss_square=False).loss_square is False        # 1 - iou: in the fused kernel since round 5
    with pytest.raises(NotImplementedError):
        select_loss('YOLOv3Loss')(iou_loss=None)
    il = select_loss('IouLoss')(loss_weight=2.5)
    yl = select_loss('YOLOv3Loss')(iou_loss=il, iou_aware_loss=select_loss('IouAwareLoss')(), ignore_thresh=0.7, scale_x_y=1.05)
    assert (yl._iou_loss._loss_weight, yl._iou_aware_loss._loss_weight, yl._ignore_thresh, yl.scale_x_y) == (2.5, 1.0, 0.7, 1.05)
    with pytest.raises(NotImplementedError):
        yl(None)


def test_training_forward_has_no_cpu_path():
    from ppyolo_hip._lib import PPYoloHipError
    cfg = PPYOLO_r18vd_Config()
    m = build_train_model(cfg)
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(PPYoloHipError):
        m(x, None, False, torch.zeros(1, 50, 4), None, None, [torch.zeros(1, 3, 86, 2, 2), torch.zeros(1, 3, 86, 4, 4)])
    with pytest.raises(NotImplementedError):
        m.head.get_loss(None, None, None, None, None)
