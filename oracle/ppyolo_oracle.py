"""CPU oracle for the PP-YOLO inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional, state_dict-driven restatement (plain PyTorch-CPU fp32
ops, the same ATen operators in the same order the reference issues them) of
`model.ppyolo.PPYOLO.forward(x, im_size)` in eval mode of
miemie2013/Pytorch-PPYOLO.  It exists to CHECK the HIP path; nothing in the
product (`pytorch-ppyolo_amd/`) may import it.  Allowed importers: `tests/`,
`__graft_entry__.smoke()`, `bench.py`'s `cpu_baseline` leg, `tools/make_goldens.py`.

Pinning: the reference repo ships no tests / golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF,
generated in the build container by `tools/make_goldens.py` (imports
/root/reference, commits `tests/golden/*.npz`).  `tests/test_oracle_golden.py`
replays every fixture through this file.

Third-party arithmetic: the reference's math is torch ATen (MKLDNN conv, sort); it
pins no torch version (README "Pytorch1.x"); this image has torch 2.10.0+rocm7.0.

Every function cites the reference file:line it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# Training-mode switch, used by oracle/train_oracle.py only.  The reference trains in nn.Module's default training mode
# (train.py never calls .eval() before its loop; `backbone.freeze()` only stops gradients): every BatchNorm2d then
# normalises with BATCH statistics and updates its running statistics (momentum 0.1), and DropBlock draws a mask.
TRAIN_MODE = [False]


def drop_block_train(x, block_size=3, keep_prob=0.9):
    """DropBlock.__call__ with is_test=False -- reference model/custom_layers.py:303-342: gamma = H^2 (1-keep) /
    (bs^2 (H-bs+1)^2) from the HEIGHT only; seeds = rand(shape) < gamma; mask = 1 - max_pool(seeds, bs, stride 1, pad 1);
    out = x * mask * numel / mask.sum().  Draws from torch's global RNG exactly once, like the reference."""
    h = torch.tensor([float(x.shape[2])], dtype=torch.float32).reshape(1, 1, 1, 1)
    bs = torch.zeros((1, 1, 1, 1), dtype=torch.float32) + block_size
    gamma = (torch.pow(h, 2) * (1 - keep_prob)) / (torch.pow(bs, 2) * torch.pow(h - bs + 1, 2))
    p = gamma.repeat(x.shape)
    seeds = (torch.rand(x.shape) < p).float()
    mask = 1.0 - F.max_pool2d(seeds, (block_size, block_size), stride=1, padding=1)
    return x * mask * float(x.numel()) / mask.sum()


# ----------------------------------------------------------------------------
# layer primitives
# ----------------------------------------------------------------------------
def conv_unit(sd, prefix, x, stride=1, act=None):
    """Conv2dUnit.forward -- reference model/custom_layers.py:243-253 (ctor :65-139).

    conv (pad=(k-1)//2, optional bias) -> eval BatchNorm2d(eps=1e-5) if present ->
    relu | leaky(0.1) | none.  DCN variant dispatches to `dcnv2`.
    """
    if prefix + '.conv.dcn_weight' in sd:
        y = dcnv2(sd, prefix + '.conv', x, stride=stride)
    else:
        w = sd[prefix + '.conv.weight']
        b = sd.get(prefix + '.conv.bias')
        y = F.conv2d(x, w, b, stride=stride, padding=(w.shape[2] - 1) // 2)
    if prefix + '.bn.weight' in sd:
        y = F.batch_norm(y, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'],
                         sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias'],
                         TRAIN_MODE[0], 0.1, 1e-5)
    if act == 'relu':
        y = F.relu(y)
    elif act == 'leaky':
        y = F.leaky_relu(y, 0.1)
    elif act is not None:
        raise NotImplementedError(act)
    return y


def dcnv2_sample(x, offset, mask, stride, padding, k=3):
    """Deformable bilinear sampling of DCNv2.forward -- reference
    model/custom_layers.py:565-662.

    Returns the modulated samples as [N, oH, oW, k*k, C].  Follows the reference's
    arithmetic: sampling happens on a zero-padded copy of size (H+2p+1, W+2p+1),
    positions are clamped to [0, H+2p-1], and the image index is folded into the
    float row coordinate (y + n*(H+2p+1)) BEFORE floor, so the fractional part for
    images n>0 carries that fp32 rounding (:626-633, :650-651).
    """
    N, C, H, W = x.shape
    oW = (W + 2 * padding - (k - 1)) // stride
    oH = (H + 2 * padding - (k - 1)) // stride
    Hp, Wp = H + 2 * padding + 1, W + 2 * padding + 1
    xp = torch.zeros((N, C, Hp, Wp), dtype=x.dtype)
    xp[:, :, padding:padding + H, padding:padding + W] = x
    # window origin (in padded coordinates) + tap position inside the window
    oy = (torch.arange(oH, dtype=torch.float32) * stride + padding).view(1, oH, 1, 1)
    ox = (torch.arange(oW, dtype=torch.float32) * stride + padding).view(1, 1, oW, 1)
    half = (k - 1) // 2
    ty = (torch.arange(k, dtype=torch.float32) - half).view(k, 1).repeat(1, k).reshape(1, 1, 1, k * k)
    tx = (torch.arange(k, dtype=torch.float32) - half).view(1, k).repeat(k, 1).reshape(1, 1, 1, k * k)
    off = offset.permute(0, 2, 3, 1).reshape(N, oH, oW, k * k, 2)      # (y, x) interleaved
    py = (oy + ty) + off[..., 0]
    px = (ox + tx) + off[..., 1]
    py = torch.clamp(py, 0.0, H + padding * 2 - 1.0)
    px = torch.clamp(px, 0.0, W + padding * 2 - 1.0)
    row0 = (torch.arange(N, dtype=torch.float32) * Hp).view(N, 1, 1, 1)
    py = py + row0                                  # image index folded into the row coordinate
    y1 = torch.floor(py)
    x1 = torch.floor(px)
    lh = py - y1
    lw = px - x1
    hh = 1 - lh
    hw = 1 - lw
    flat = xp.permute(0, 2, 3, 1).reshape(N * Hp * Wp, C)
    iy1 = y1.long()
    ix1 = x1.long()

    def corner(iy, ix):
        return flat[(iy * Wp + ix).reshape(-1)].reshape(N, oH, oW, k * k, C)

    v1 = corner(iy1, ix1)
    v2 = corner(iy1, ix1 + 1)
    v3 = corner(iy1 + 1, ix1)
    v4 = corner(iy1 + 1, ix1 + 1)
    w1 = (hh * hw).unsqueeze(-1)
    w2 = (hh * lw).unsqueeze(-1)
    w3 = (lh * hw).unsqueeze(-1)
    w4 = (lh * lw).unsqueeze(-1)
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    val = val * mask.permute(0, 2, 3, 1).unsqueeze(-1)
    return val


def dcnv2(sd, prefix, x, stride=1, padding=1):
    """DCNv2.forward -- reference model/custom_layers.py:551-677.

    conv_offset (3x3, bias) -> 18 offset channels (y,x interleaved per tap) + 9 mask
    logits (sigmoid) (:558-561); sampling (see dcnv2_sample); contraction with
    dcn_weight in (c, kh, kw) K-order as a 1x1 conv (:661-676); no bias.
    """
    wd = sd[prefix + '.dcn_weight']
    k = wd.shape[2]
    om = F.conv2d(x, sd[prefix + '.conv_offset.weight'], sd[prefix + '.conv_offset.bias'],
                  stride=stride, padding=padding)
    offset = om[:, :2 * k * k]
    mask = torch.sigmoid(om[:, 2 * k * k:])
    val = dcnv2_sample(x, offset, mask, stride, padding, k)        # [N,oH,oW,kk,C]
    N, oH, oW, KK, C = val.shape
    cols = val.reshape(N, oH, oW, k, k, C).permute(0, 1, 2, 5, 3, 4).reshape(N, oH, oW, C * k * k)
    cols = cols.permute(0, 3, 1, 2)
    out = F.conv2d(cols, wd.reshape(wd.shape[0], C * k * k, 1, 1), stride=1)
    if prefix + '.dcn_bias' in sd:
        out = out + sd[prefix + '.dcn_bias'].view(1, -1, 1, 1)
    return out


def coord_concat(x):
    """CoordConv.__call__ -- reference model/custom_layers.py:261-272: append x_range
    (varies along W) then y_range (varies along H), both in [-1, 1]."""
    b, _, h, w = x.shape
    xr = torch.arange(0, w, dtype=torch.float32) / (w - 1) * 2.0 - 1
    yr = torch.arange(0, h, dtype=torch.float32) / (h - 1) * 2.0 - 1
    xr = xr.view(1, 1, 1, w).repeat(b, 1, h, 1)
    yr = yr.view(1, 1, h, 1).repeat(b, 1, 1, w)
    return torch.cat([x, xr, yr], dim=1)


def spp(x):
    """SPP.__call__ ('asc' order) -- reference model/custom_layers.py:281-290."""
    return torch.cat([x, F.max_pool2d(x, 5, 1, 2), F.max_pool2d(x, 9, 1, 4),
                      F.max_pool2d(x, 13, 1, 6)], dim=1)


# ----------------------------------------------------------------------------
# backbones
# ----------------------------------------------------------------------------
def _stem(sd, x):
    # reference model/resnet_vd.py:100-103, :133-136
    x = conv_unit(sd, 'backbone.stage1_conv1_1', x, 2, 'relu')
    x = conv_unit(sd, 'backbone.stage1_conv1_2', x, 1, 'relu')
    x = conv_unit(sd, 'backbone.stage1_conv1_3', x, 1, 'relu')
    return F.max_pool2d(x, 3, 2, 1)


def _bottleneck(sd, p, x, stride, has_proj, is_first):
    """ConvBlock / IdentityBlock -- reference model/resnet_vd.py:48-57, :81-87
    (stride on the 3x3, `downsample_in3x3=True` :19-20; vd shortcut = AvgPool2d(2,2)
    then 1x1 unless first stage :29-33)."""
    y = conv_unit(sd, p + '.conv1', x, 1, 'relu')
    y = conv_unit(sd, p + '.conv2', y, stride, 'relu')
    y = conv_unit(sd, p + '.conv3', y, 1, None)
    if has_proj:
        s = x
        if not is_first:
            s = F.avg_pool2d(s, 2, 2, 0)
            s = conv_unit(sd, p + '.conv4', s, 1, None)
        else:
            s = conv_unit(sd, p + '.conv4', s, stride, None)
    else:
        s = x
    return F.relu(y + s)


def resnet50vd(sd, x, feature_maps=(3, 4, 5)):
    """Resnet50Vd.forward -- reference model/resnet_vd.py:132-168."""
    x = _stem(sd, x)
    feats = {}
    for stage, nblk in ((2, 3), (3, 4), (4, 6), (5, 3)):
        for b in range(nblk):
            p = 'backbone.stage%d_%d' % (stage, b)
            if b == 0:
                x = _bottleneck(sd, p, x, 1 if stage == 2 else 2, True, stage == 2)
            else:
                x = _bottleneck(sd, p, x, 1, False, False)
        feats[stage] = x
    return [feats[s] for s in (2, 3, 4, 5) if s in feature_maps]


def _basic(sd, p, x, stride, is_first):
    """BasicBlock.forward -- reference model/resnet_vd.py:256-267."""
    y = conv_unit(sd, p + '.conv1', x, stride, 'relu')
    y = conv_unit(sd, p + '.conv2', y, 1, None)
    if stride == 2 or is_first:
        s = x
        if not is_first:
            s = F.avg_pool2d(s, 2, 2, 0)
            s = conv_unit(sd, p + '.conv3', s, 1, None)
        else:
            s = conv_unit(sd, p + '.conv3', s, stride, None)
    else:
        s = x
    return F.relu(y + s)


def resnet18vd(sd, x, feature_maps=(4, 5)):
    """Resnet18Vd.forward -- reference model/resnet_vd.py:302-330."""
    x = _stem(sd, x)
    feats = {}
    for stage in (2, 3, 4, 5):
        for b in range(2):
            p = 'backbone.stage%d_%d' % (stage, b)
            x = _basic(sd, p, x, 2 if (b == 0 and stage > 2) else 1, b == 0 and stage == 2)
        feats[stage] = x
    return [feats[s] for s in (2, 3, 4, 5) if s in feature_maps]


# ----------------------------------------------------------------------------
# head
# ----------------------------------------------------------------------------
def detection_block(sd, p, x, conv_block_num, is_first, coord, use_spp, drop_block=True):
    """DetectionBlock.__call__ -- reference model/head.py:146-231.  Layer indices inside
    `layers` count the parameter-less CoordConv / SPP / DropBlock entries, exactly as
    the reference's ModuleList does (that is what the state_dict keys encode)."""
    cc = (lambda t: coord_concat(t)) if coord else (lambda t: t)
    idx = 0
    for j in range(conv_block_num):
        x = conv_unit(sd, '%s.layers.%d' % (p, idx + 1), cc(x), 1, 'leaky')
        idx += 2
        if use_spp and is_first and j == 1:
            x = spp(x)
            x = conv_unit(sd, '%s.layers.%d' % (p, idx + 1), x, 1, 'leaky')
            x = conv_unit(sd, '%s.layers.%d' % (p, idx + 2), x, 1, 'leaky')
            idx += 3
        else:
            x = conv_unit(sd, '%s.layers.%d' % (p, idx), x, 1, 'leaky')
            idx += 1
        if drop_block and j == 0 and not is_first:
            idx += 1                       # DropBlock(is_test) == identity (custom_layers.py:304-305)
            if TRAIN_MODE[0]:
                x = drop_block_train(x)
    if drop_block and is_first:
        idx += 1
        if TRAIN_MODE[0]:
            x = drop_block_train(x)
    route = conv_unit(sd, '%s.layers.%d' % (p, idx + 1), cc(x), 1, 'leaky')
    tip = conv_unit(sd, '%s.tip_layers.1' % p, cc(route), 1, 'leaky')
    return route, tip


def head_outputs(sd, feats, hcfg):
    """YOLOv3Head._get_outputs -- reference model/head.py:381-398 (deepest level first;
    concat order [upsampled route, backbone feature] :391)."""
    n_lvl = len(hcfg['anchor_masks'])
    blocks = feats[::-1][:n_lvl]
    outs, route = [], None
    for i, blk in enumerate(blocks):
        if i > 0:
            blk = torch.cat([route, blk], dim=1)
        route, tip = detection_block(sd, 'head.detection_blocks.%d' % i, blk,
                                     hcfg.get('conv_block_num', 2), i == 0,
                                     hcfg.get('coord_conv', True), hcfg.get('spp', True),
                                     hcfg.get('drop_block', True))
        outs.append(conv_unit(sd, 'head.yolo_output_convs.%d' % i, tip, 1, None))
        if i < n_lvl - 1:
            route = conv_unit(sd, 'head.upsample_layers.%d' % (2 * i), route, 1, 'leaky')
            route = F.interpolate(route, scale_factor=2, mode='nearest')
    return outs


def iou_aware_score(out, an_num, num_classes, factor):
    """get_iou_aware_score -- reference model/head.py:83-141: first `an_num` channels
    are IoU logits; obj' = sigmoid(obj)^(1-f) * sigmoid(ioup)^f, re-encoded as a logit
    by _de_sigmoid (clamp to [1e-7, 1e7] twice, :97-109)."""
    ioup = torch.sigmoid(out[:, :an_num])
    rest = out[:, an_num:]
    step = rest.shape[1] // an_num
    eps = 1e-7
    parts = []
    for m in range(an_num):
        parts.append(rest[:, step * m:step * m + 4])
        obj = torch.sigmoid(rest[:, step * m + 4:step * m + 5])
        new = torch.pow(obj, 1 - factor) * torch.pow(ioup[:, m:m + 1], factor)
        new = torch.clamp(new, eps, 1 / eps)
        new = 1.0 / new - 1.0
        new = torch.clamp(new, eps, 1 / eps)
        parts.append(-torch.log(new))
        parts.append(rest[:, step * m + 5:step * m + 5 + num_classes])
    return torch.cat(parts, dim=1)


def yolo_box(out, anchors, stride, num_classes, scale_x_y, im_size, clip_bbox=True):
    """yolo_box -- reference model/head.py:21-80.  Box order (h, w, anchor); grid x is
    the column index; rescale `/ S / stride * (im_w, im_h)`; clip x0<0 -> x0*0 (keeps
    -0.0), x1>im_w -> im_w (:68-77)."""
    o = out.permute(0, 2, 3, 1)
    N, S = o.shape[0], o.shape[1]
    A = len(anchors)
    o = o.reshape(N, S, S, A, 5 + num_classes)
    gx = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1, 1).repeat(1, S, 1, 1, 1)
    gy = torch.arange(S, dtype=torch.float32).view(1, S, 1, 1, 1).repeat(1, 1, S, 1, 1)
    grid = torch.cat([gx, gy], dim=-1).repeat(N, 1, 1, A, 1)
    xy = (scale_x_y * torch.sigmoid(o[..., 0:2]) + grid - (scale_x_y - 1.0) * 0.5) * stride
    wh = torch.exp(o[..., 2:4]) * torch.tensor(np.asarray(anchors, dtype=np.float32))
    xyxy = torch.cat([xy - wh / 2, xy + wh / 2], dim=-1)
    scores = torch.sigmoid(o[..., 4:5]) * torch.sigmoid(o[..., 5:])
    xyxy = xyxy.reshape(N, S * S * A, 4)
    scores = scores.reshape(N, S * S * A, num_classes)
    wh_im = torch.cat([im_size[:, 1:2], im_size[:, 0:1]], 1).unsqueeze(1).repeat(1, xyxy.shape[1], 1)
    p0 = xyxy[:, :, 0:2] / S / stride * wh_im
    p1 = xyxy[:, :, 2:4] / S / stride * wh_im
    if clip_bbox:
        x0, y0 = p0[:, :, 0:1], p0[:, :, 1:2]
        x1, y1 = p1[:, :, 0:1], p1[:, :, 1:2]
        x0 = torch.where(x0 < 0, x0 * 0, x0)
        y0 = torch.where(y0 < 0, y0 * 0, y0)
        x1 = torch.where(x1 > wh_im[:, :, 0:1], wh_im[:, :, 0:1], x1)
        y1 = torch.where(y1 > wh_im[:, :, 1:2], wh_im[:, :, 1:2], y1)
        xyxy = torch.cat([x0, y0, x1, y1], -1)
    else:
        xyxy = torch.cat([p0, p1], -1)
    return xyxy, scores


def decode_all(outs, hcfg, im_size):
    """Box / score assembly of YOLOv3Head.get_prediction -- reference
    model/head.py:439-453 (levels concatenated deepest first)."""
    anchors = np.asarray(hcfg['anchors'], dtype=np.float32)
    boxes, scores = [], []
    for i, o in enumerate(outs):
        mask = hcfg['anchor_masks'][i]
        if hcfg.get('iou_aware', True):
            o = iou_aware_score(o, len(mask), hcfg['num_classes'], hcfg['iou_aware_factor'])
        b, s = yolo_box(o, anchors[mask], hcfg['downsample'][i], hcfg['num_classes'],
                        hcfg['scale_x_y'], im_size, hcfg.get('clip_bbox', True))
        boxes.append(b)
        scores.append(s)
    return torch.cat(boxes, dim=1), torch.cat(scores, dim=1)


# ----------------------------------------------------------------------------
# Matrix-NMS
# ----------------------------------------------------------------------------
def pairwise_iou(a, b):
    """jaccard / intersect -- reference model/matrix_nms.py:15-47 (no +1; inter/union,
    union may be 0 -> NaN/inf propagate)."""
    hi = torch.min(a[:, None, 2:], b[None, :, 2:])
    lo = torch.max(a[:, None, :2], b[None, :, :2])
    d = torch.clamp(hi - lo, min=0)
    inter = d[:, :, 0] * d[:, :, 1]
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None].expand_as(inter)
    area_b = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :].expand_as(inter)
    return inter / (area_a + area_b - inter)


def matrix_nms_decay(boxes, labels, scores, use_gaussian=False, sigma=2.0):
    """_matrix_nms -- reference model/matrix_nms.py:51-97.  Inputs sorted by score desc."""
    n = len(labels)
    iou = pairwise_iou(boxes, boxes).triu(diagonal=1)
    lx = labels.expand(n, n)
    same = (lx == lx.transpose(1, 0)).float().triu(diagonal=1)
    d = iou * same
    comp, _ = d.max(0)
    comp = comp.expand(n, n).transpose(1, 0)
    if use_gaussian:
        coef, _ = (torch.exp(-1 * sigma * (d ** 2)) / torch.exp(-1 * sigma * (comp ** 2))).min(0)
    else:
        coef, _ = ((1 - d) / (1 - comp)).min(0)
    return scores * coef


def _argsort_desc_total(scores, tiebreak):
    """Total order used by this build where the reference calls
    torch.argsort(descending=True) (model/matrix_nms.py:120, :140): score descending,
    ties broken by ascending `tiebreak` (candidate enumeration order).  On tie-free
    inputs this equals the reference's permutation; torch's CPU argsort is not stable,
    so on ties the reference's order is unspecified (hard part 1)."""
    s = scores.detach().cpu().numpy()
    order = np.lexsort((tiebreak, -s.astype(np.float64)))
    return torch.from_numpy(order.astype(np.int64))


def matrix_nms(boxes, scores, score_threshold, post_threshold, nms_top_k, keep_top_k,
               use_gaussian=False, gaussian_sigma=2., return_index=False):
    """matrix_nms -- reference model/matrix_nms.py:102-151.  Returns [K,6] rows
    (label, score, x0, y0, x1, y1) or the [[-1]*6] sentinel.  With return_index also
    returns the flat candidate index (box*C + class) of every kept row."""
    C = scores.shape[1]
    sel = scores > score_threshold
    cand_scores = scores[sel]
    empty = torch.zeros((1, 6)) - 1.0
    if len(cand_scores) == 0:
        return (empty, np.zeros((0,), np.int64)) if return_index else empty
    nz = sel.nonzero()
    flat = (nz[:, 0] * C + nz[:, 1]).numpy()
    order = _argsort_desc_total(cand_scores, flat)
    if nms_top_k > 0 and len(order) > nms_top_k:
        order = order[:nms_top_k]
    b = boxes[nz[:, 0]][order]
    s = cand_scores[order]
    l = nz[:, 1][order]
    f = flat[order.numpy()]
    s = matrix_nms_decay(b, l, s, use_gaussian, gaussian_sigma)
    keep = s >= post_threshold
    if keep.sum() == 0:
        return (empty, np.zeros((0,), np.int64)) if return_index else empty
    b, s, l, f = b[keep], s[keep], l[keep], f[keep.numpy()]
    order = _argsort_desc_total(s, np.arange(len(s)))
    if len(order) > keep_top_k:
        order = order[:keep_top_k]
    b, s, l, f = b[order], s[order], l[order], f[order.numpy()]
    pred = torch.cat([l.unsqueeze(1).float(), s.unsqueeze(1), b], 1)
    return (pred, f) if return_index else pred


# ----------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------
def ppyolo_forward(sd, cfg, x, im_size, return_index=False):
    """PPYOLO.forward(eval=True) -- reference model/ppyolo.py:19-22 +
    model/head.py:424-469.  `cfg` is a config object (backbone_type, backbone, head,
    nms_cfg).  Returns the list of per-image [K,6] tensors."""
    with torch.no_grad():
        if cfg.backbone_type == 'Resnet50Vd':
            feats = resnet50vd(sd, x, cfg.backbone['feature_maps'])
        else:
            feats = resnet18vd(sd, x, cfg.backbone['feature_maps'])
        outs = head_outputs(sd, feats, cfg.head)
        boxes, scores = decode_all(outs, cfg.head, im_size)
        nms = dict(cfg.nms_cfg)
        nms.pop('nms_type')
        return [matrix_nms(boxes[i], scores[i], return_index=return_index, **nms)
                for i in range(boxes.shape[0])]


def backbone_and_head(sd, cfg, x):
    """Feature maps + raw head outputs (for per-stage parity checks)."""
    with torch.no_grad():
        if cfg.backbone_type == 'Resnet50Vd':
            feats = resnet50vd(sd, x, cfg.backbone['feature_maps'])
        else:
            feats = resnet18vd(sd, x, cfg.backbone['feature_maps'])
        return feats, head_outputs(sd, feats, cfg.head)
