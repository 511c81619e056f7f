"""CPU oracle for ONE TRAINING STEP of PP-YOLO (BASELINE config 5)  --  TEST INFRASTRUCTURE ONLY.

Restates, as plain PyTorch-CPU ops driven by a state_dict, what the reference's `train.py:416-443` does up to
`all_loss.backward()`:

  * the training-mode forward: the eval forward of oracle/ppyolo_oracle.py with BatchNorm on BATCH statistics everywhere
    (the reference never calls `.eval()`; `backbone.freeze()` -- model/resnet_vd.py:174-200, custom_layers.py:142-164 --
    only clears `requires_grad`) and DropBlock drawing its mask (custom_layers.py:303-342);
  * `YOLOv3Loss._get_fine_grained_loss` (model/losses.py:121-253) with `IouLoss` / `IouAwareLoss`
    (model/iou_losses.py:39-246) and the ignore mask of `_calc_obj_loss` (losses.py:296-356);
  * the backward pass: torch autograd over these ops, which is what the reference's own backward is.

Pinned to tests/golden/g12_train_*.npz -- loss terms, head outputs, d loss / d head outputs, gradient digests of every
trainable tensor and updated BatchNorm statistics produced by the reference itself (tools/make_goldens.py g12).
Nothing in the product may import this file.
"""
import numpy as np
import torch

from . import ppyolo_oracle as orc


# ----------------------------------------------------------------------------
# training-mode forward
# ----------------------------------------------------------------------------
def trainable_keys(sd, freeze_at=5):
    """Parameters that receive gradients: backbone.freeze() stops stages 1 .. freeze_at (model/resnet_vd.py:174-200;
    freeze_at = 5 in both configurations, config/ppyolo_2x.py:101: the whole backbone), the stages above and the whole head
    train.  Buffers never do."""
    def stage(k):
        return int(k[len('backbone.stage')]) if k.startswith('backbone.stage') else 6
    return [k for k in sd if stage(k) > freeze_at and not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]


def forward_train(sd, cfg, x):
    """-> (head outputs, state): `state` is a copy of `sd` whose head parameters are autograd leaves and whose BatchNorm
    running statistics have been updated by this forward (momentum 0.1, unbiased variance -- torch.nn.BatchNorm2d)."""
    state = {k: v.clone() for k, v in sd.items()}
    for k in trainable_keys(sd, cfg.backbone.get('freeze_at', 5)):
        state[k].requires_grad_(True)
    orc.TRAIN_MODE[0] = True
    try:
        if cfg.backbone_type == 'Resnet50Vd':
            feats = orc.resnet50vd(state, x, cfg.backbone['feature_maps'])
        else:
            feats = orc.resnet18vd(state, x, cfg.backbone['feature_maps'])
        outs = orc.head_outputs(state, feats, cfg.head)
    finally:
        orc.TRAIN_MODE[0] = False
    for k in state:
        if k.endswith('num_batches_tracked'):
            state[k] += 1
    return outs, state


# ----------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------
def split_output(out, an_num, num_classes):
    """losses.py:255-270: [N, an*(5+C), S, S] -> x, y, w, h, obj [N,an,S,S] and cls [N,an,S,S,C]."""
    N, _, S, _ = out.shape
    o = out.reshape(N, an_num, 5 + num_classes, S, S)
    return o[:, :, 0], o[:, :, 1], o[:, :, 2], o[:, :, 3], o[:, :, 4], o[:, :, 5:].permute(0, 1, 3, 4, 2)


def split_target(t):
    """losses.py:272-294: target [N, an, 6+C, S, S] -> tx, ty, tw, th, tscale, tobj, tcls."""
    return t[:, :, 0], t[:, :, 1], t[:, :, 2], t[:, :, 3], t[:, :, 4], t[:, :, 5], t[:, :, 6:].permute(0, 1, 3, 4, 2)


def bbox_transform(dcx, dcy, dw, dh, anchors, downsample, is_gt, scale_x_y, eps=1e-10):
    """IouLoss._bbox_transform -- iou_losses.py:135-190: encoded (x, y, w, h) -> corner boxes in units of the input
    image side.  `rows` runs along the LAST axis (grid x), `cols` along H; anchors = [w0, h0, w1, h1, ...] of this level."""
    N, A, S, _ = dcx.shape
    gx = torch.arange(S, dtype=torch.float32).view(1, 1, 1, S).expand(N, A, S, S)
    gy = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1).expand(N, A, S, S)
    if is_gt:
        cx, cy = (dcx + gx) / S, (dcy + gy) / S
    else:
        sx, sy = torch.sigmoid(dcx), torch.sigmoid(dcy)
        if abs(scale_x_y - 1.0) > eps:
            sx = scale_x_y * sx - 0.5 * (scale_x_y - 1)
            sy = scale_x_y * sy - 0.5 * (scale_x_y - 1)
        cx, cy = (sx + gx) / S, (sy + gy) / S
    aw = torch.tensor([float(a) for a in anchors[0::2]]).view(1, A, 1, 1)
    ah = torch.tensor([float(a) for a in anchors[1::2]]).view(1, A, 1, 1)
    pw = (torch.exp(dw) * aw) / (S * downsample)
    ph = (torch.exp(dh) * ah) / (S * downsample)
    return cx - 0.5 * pw, cy - 0.5 * ph, cx + 0.5 * pw, cy + 0.5 * ph


def iou_pairs(pred, gt, eps=1e-10):
    """IouLoss._iou (ciou_term=False) -- iou_losses.py:74-96: element-wise IoU of matching boxes, `+ eps` in the union."""
    x1, y1, x2, y2 = pred
    x1g, y1g, x2g, y2g = gt
    x2, y2 = torch.max(x1, x2), torch.max(y1, y2)
    iw = torch.clamp(torch.min(x2, x2g) - torch.max(x1, x1g), min=0)
    ih = torch.clamp(torch.min(y2, y2g) - torch.max(y1, y1g), min=0)
    inter = iw * ih
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter + eps
    return inter / union


def iou_loss(x, y, w, h, tx, ty, tw, th, anchors, downsample, scale_x_y, weight=2.5, square=True):
    """IouLoss.__call__ -- iou_losses.py:39-72: (1 - iou^2) * loss_weight."""
    k = iou_pairs(bbox_transform(x, y, w, h, anchors, downsample, False, scale_x_y),
                  bbox_transform(tx, ty, tw, th, anchors, downsample, True, scale_x_y))
    return ((1. - k * k) if square else (1. - k)) * weight


def iou_aware_loss(ioup, x, y, w, h, tx, ty, tw, th, anchors, downsample, scale_x_y, weight=1.0):
    """IouAwareLoss.__call__ -- iou_losses.py:206-246.  Faithful to the reference including its reduction: the
    cross-entropy term is summed over the LAST axis (grid x) and broadcast back (`.sum(-1).unsqueeze(-1)`, :241-242)."""
    k = iou_pairs(bbox_transform(x, y, w, h, anchors, downsample, False, scale_x_y),
                  bbox_transform(tx, ty, tw, th, anchors, downsample, True, scale_x_y))
    return (k * (0 - torch.log(ioup + 1e-9))).sum(-1).unsqueeze(-1) * weight


def train_boxes(out, anchors_px, stride, num_classes, scale_x_y):
    """paddle_yolo_box with im_size = 1, clip_bbox=False -- losses.py:22-83: boxes in (anchor, h, w) order (:56-60),
    in units of the input image side."""
    N, _, S, _ = out.shape
    A = len(anchors_px)
    o = out.permute(0, 2, 3, 1).reshape(N, S, S, A, 5 + num_classes)
    gx = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1, 1).expand(N, S, S, A, 1)
    gy = torch.arange(S, dtype=torch.float32).view(1, S, 1, 1, 1).expand(N, S, S, A, 1)
    xy = (scale_x_y * torch.sigmoid(o[..., 0:2]) + torch.cat([gx, gy], -1) - (scale_x_y - 1.0) * 0.5) * stride
    wh = torch.exp(o[..., 2:4]) * torch.tensor(np.asarray(anchors_px, dtype=np.float32))
    xyxy = torch.cat([xy - wh / 2, xy + wh / 2], -1).permute(0, 3, 1, 2, 4).reshape(N, A * S * S, 4)
    ones = torch.ones((N, 1, 2), dtype=torch.float32)
    return torch.cat([xyxy[:, :, 0:2] / S / stride * ones, xyxy[:, :, 2:4] / S / stride * ones], -1)


def obj_loss(out, obj, tobj, gt_box, anchors, num_classes, downsample, ignore_thresh, scale_x_y):
    """YOLOv3Loss._calc_obj_loss -- losses.py:296-356: a prediction whose best IoU with any ground-truth box exceeds
    `ignore_thresh` is neither positive nor negative."""
    N, _, S, _ = out.shape
    an = len(anchors) // 2
    boxes = train_boxes(out, np.reshape(np.array(anchors), (-1, 2)).astype(np.float32), downsample, num_classes, scale_x_y)
    best = []
    for pred, gt in zip(boxes, gt_box):
        g = torch.cat([gt[:, 0:1] - gt[:, 2:3] / 2., gt[:, 1:2] - gt[:, 3:4] / 2.,
                       gt[:, 0:1] + gt[:, 2:3] / 2., gt[:, 1:2] + gt[:, 3:4] / 2.], 1)
        best.append(orc.pairwise_iou(pred, g).max(-1)[0])
    iou_mask = (torch.stack(best) <= ignore_thresh).float().reshape(N, an, S, S).detach()
    noobj = (1.0 - (tobj > 0.).float()) * iou_mask
    s = torch.sigmoid(obj)
    pos = (tobj * (0 - torch.log(s + 1e-9))).sum((1, 2, 3))
    neg = (noobj * (0 - torch.log(1 - s + 1e-9))).sum((1, 2, 3))
    return pos, neg


def yolov3_loss(outs, targets, gt_box, cfg):
    """YOLOv3Loss._get_fine_grained_loss -- losses.py:121-253, with the loss objects train.py:241-248 builds from the
    config.  Returns the reference's dict of loss terms (each the batch mean, summed over the levels)."""
    hcfg, lcfg = cfg.head, cfg.yolo_loss
    C = hcfg['num_classes'] if 'num_classes' in hcfg else 80
    iou_aware = bool(hcfg.get('iou_aware', False))
    scale_x_y = lcfg.get('scale_x_y', 1.)
    names = ['loss_xy', 'loss_wh', 'loss_obj', 'loss_cls', 'loss_iou'] + (['loss_iou_aware'] if iou_aware else [])
    tot = {k: 0.0 for k in names}
    for i, (out, tgt) in enumerate(zip(outs, targets)):
        mask = hcfg['anchor_masks'][i]
        anchors = [v for m in mask for v in hcfg['anchors'][m]]        # YOLOv3Head.mask_anchors (head.py:298-303)
        ds = hcfg['downsample'][i]
        an = len(mask)
        ioup = None
        if iou_aware:                                                  # _split_ioup (losses.py:243-253)
            ioup, out = torch.sigmoid(out[:, :an]), out[:, an:]
        x, y, w, h, obj, cls = split_output(out, an, C)
        tx, ty, tw, th, tscale, tobj, tcls = split_target(tgt)
        ts = tscale * tobj
        if abs(scale_x_y - 1.0) < 1e-10:
            sx, sy = torch.sigmoid(x), torch.sigmoid(y)
            lx = (tx * (0 - torch.log(sx + 1e-9)) + (1 - tx) * (0 - torch.log(1 - sx + 1e-9))) * ts
            ly = (ty * (0 - torch.log(sy + 1e-9)) + (1 - ty) * (0 - torch.log(1 - sy + 1e-9))) * ts
        else:                                                          # Grid Sensitive: L1 on the decoded offset
            lx = torch.abs(scale_x_y * torch.sigmoid(x) - 0.5 * (scale_x_y - 1.0) - tx) * ts
            ly = torch.abs(scale_x_y * torch.sigmoid(y) - 0.5 * (scale_x_y - 1.0) - ty) * ts
        lx, ly = lx.sum((1, 2, 3)), ly.sum((1, 2, 3))
        lw = (torch.abs(w - tw) * ts).sum((1, 2, 3))
        lh = (torch.abs(h - th) * ts).sum((1, 2, 3))
        li = iou_loss(x, y, w, h, tx, ty, tw, th, anchors, ds, scale_x_y, cfg.iou_loss['loss_weight'],
                      cfg.iou_loss.get('loss_square', True)) * ts
        tot['loss_iou'] = tot['loss_iou'] + li.sum((1, 2, 3)).mean()
        if iou_aware:
            la = iou_aware_loss(ioup, x, y, w, h, tx, ty, tw, th, anchors, ds, scale_x_y, cfg.iou_aware_loss['loss_weight']) * tobj
            tot['loss_iou_aware'] = tot['loss_iou_aware'] + la.sum((1, 2, 3)).mean()
        pos, neg = obj_loss(out, obj, tobj, gt_box, anchors, C, ds, lcfg['ignore_thresh'], scale_x_y)
        sc = torch.sigmoid(cls)
        lc = (tcls * (0 - torch.log(sc + 1e-9)) + (1 - tcls) * (0 - torch.log(1 - sc + 1e-9))).sum(4)
        lc = (lc * tobj).sum((1, 2, 3))
        tot['loss_xy'] = tot['loss_xy'] + (lx + ly).mean()
        tot['loss_wh'] = tot['loss_wh'] + (lw + lh).mean()
        tot['loss_obj'] = tot['loss_obj'] + (pos + neg).mean()
        tot['loss_cls'] = tot['loss_cls'] + lc.mean()
    return tot


def train_step(sd, cfg, x, gt_box, targets, rng_seed=None):
    """One forward + backward.  -> dict(losses, all_loss, outs, douts, grads {key: tensor}, state)."""
    if rng_seed is not None:
        torch.manual_seed(rng_seed)
    outs, state = forward_train(sd, cfg, x)
    for o in outs:
        o.retain_grad()
    losses = yolov3_loss(outs, targets, gt_box, cfg)
    all_loss = 0.0
    for k in losses:                                                   # train.py:428-432: plain sum of the terms
        all_loss = all_loss + losses[k]
    all_loss.backward()
    grads = {k: v.grad for k, v in state.items() if v.requires_grad}
    return dict(losses=losses, all_loss=all_loss, outs=outs, douts=[o.grad for o in outs], grads=grads,
                state={k: v.detach() for k, v in state.items()})


def ema_update(shadow, param, step, ema_decay=0.9998):
    """ExponentialMovingAverage.update for one tensor -- reference model/EMA.py:29-44: decay warms up as (1+t)/(10+t);
    numpy evaluates `decay * old + (1 - decay) * new` in float32 (the Python floats are weak scalars)."""
    decay = min(ema_decay, (1 + step) / (10 + step))
    old, new = np.asarray(shadow, dtype=np.float32), np.asarray(param, dtype=np.float32)
    return decay * old + (1 - decay) * new, decay
