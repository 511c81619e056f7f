"""Factories (reference config/get_model.py:16-40)."""


def select_backbone(name):
    from model.resnet_vd import Resnet50Vd, Resnet18Vd
    return {'Resnet50Vd': Resnet50Vd, 'Resnet18Vd': Resnet18Vd}.get(name)


def select_head(name):
    from model.head import YOLOv3Head
    return {'YOLOv3Head': YOLOv3Head}.get(name)


def select_loss(name):
    from model.losses import YOLOv3Loss
    from model.iou_losses import IouLoss, IouAwareLoss
    return {'YOLOv3Loss': YOLOv3Loss, 'IouLoss': IouLoss, 'IouAwareLoss': IouAwareLoss}.get(name)


def select_optimizer(name):
    import torch
    return {'Momentum': torch.optim.SGD, 'Adam': torch.optim.Adam, 'SGD': torch.optim.SGD}.get(name)
