"""Factories (reference config/get_model.py:16-24).  Loss / optimizer selectors belong to
training and are out of scope."""


def select_backbone(name):
    from model.resnet_vd import Resnet50Vd, Resnet18Vd
    return {'Resnet50Vd': Resnet50Vd, 'Resnet18Vd': Resnet18Vd}.get(name)


def select_head(name):
    from model.head import YOLOv3Head
    return {'YOLOv3Head': YOLOv3Head}.get(name)
