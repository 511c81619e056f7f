"""YOLO training targets from ground-truth boxes -- the data format on the near side of the training step (SURVEY.md
section 8f rank 2): the reference's `Gt2YoloTarget` (tools/transform.py:1211-1316), which its reader threads run on the host
for every batch before `train.py:416-426` hands `target0..2` to the model.  Host-side numpy here as well (a few dozen boxes per
image); pinned bit for bit to targets the reference's own operator produced (tests/golden/g12_train_*.npz).

Per ground-truth box (normalised cx, cy, w, h; class; mixup score): the best of ALL anchors by IoU of (w, h) alone; if that
anchor belongs to this level, cell (int(cx * grid), int(cy * grid)) of it gets
    tx, ty = fractional cell offsets; tw, th = log(box side in px / anchor side); tscale = 2 - w * h; tobj = score; one-hot class.
"""
import numpy as np


def _wh_iou_all(gw, gh, an_hw):
    """IoU of every box [B] with every anchor [A], both anchored at the origin (the reference's jaccard_overlap on (0, 0, w, h)
    boxes, tools/transform.py:1192-1208) -> float64 [B, A].  The reference evaluates it on numpy SCALARS -- float32 box sides
    against float64 anchor sides -- so a product of two float32 operands is rounded to float32 before it is widened; the same
    roundings are made here (a best-anchor decision between two near-equal IoUs depends on them)."""
    gw, gh = gw.astype(np.float32)[:, None], gh.astype(np.float32)[:, None]
    aw, ah = an_hw[None, :, 0], an_hw[None, :, 1]                   # float64
    w_box, h_box = gw <= aw, gh <= ah                               # min() keeps its first argument on a tie
    mw, mh = np.where(w_box, gw.astype(np.float64), aw), np.where(h_box, gh.astype(np.float64), ah)
    inter = np.where(w_box & h_box, (gw * gh).astype(np.float64) + 0 * aw, mw * mh)      # both minima float32: a float32 product
    union = ((gw * gh).astype(np.float64) + aw * ah) - inter
    iou = inter / union
    return np.where((aw <= 0.) | (gw <= 0.) | (ah <= 0.) | (gh <= 0.), 0., iou)


def gt2yolo_target(gt_bbox, gt_class, gt_score, anchors, anchor_masks, downsample_ratios, num_classes, image_size, iou_thresh=1.):
    """gt_bbox [N, G, 4], gt_class [N, G], gt_score [N, G] (zero rows = padding) -> list over levels of float32
    [N, len(mask), 6 + num_classes, grid, grid].  Vectorised over all boxes of the batch: one IoU table [N * G, anchors], one
    scatter per level; boxes are written in (image, box) order, so where two boxes claim the same (anchor, cell) the later one
    wins field by field and the class rows of both stay set -- exactly what the reference's nested loops leave behind."""
    gt_bbox = np.asarray(gt_bbox, dtype=np.float32)
    gt_class, gt_score = np.asarray(gt_class), np.asarray(gt_score, dtype=np.float32)
    N, G = gt_bbox.shape[:2]
    size = int(image_size)
    an = np.asarray(anchors)
    an_hw = an / np.array([[size, size]])
    img, box = np.nonzero((gt_bbox[..., 2] > 0.) & (gt_bbox[..., 3] > 0.) & (gt_score > 0.))      # row-major = the loops' order
    gx, gy, gw, gh = (gt_bbox[img, box, k] for k in range(4))
    cls, score = gt_class[img, box].astype(np.int64), gt_score[img, box]
    iou = _wh_iou_all(gw, gh, an_hw)
    best = np.where(iou.max(axis=1) > 0., iou.argmax(axis=1), -1)      # first anchor attaining the maximum, none when all are 0
    f32 = np.float32
    out = []
    for mask, ds in zip(anchor_masks, downsample_ratios):
        grid = int(size / ds)
        target = np.zeros((N, len(mask), 6 + num_classes, grid, grid), dtype=np.float32)
        gi, gj = (gx * f32(grid)).astype(np.int64), (gy * f32(grid)).astype(np.int64)
        picks = []
        for k, a_idx in enumerate(mask):
            sel = best == a_idx
            if iou_thresh < 1:
                sel = sel | ((best != a_idx) & (iou[:, a_idx] > iou_thresh))
            picks.append((k, a_idx, np.nonzero(sel)[0]))
        # one box at a time inside a level would be the reference's order; fields of different (k, cell) never collide, and
        # within one (k, cell) numpy's fancy assignment keeps the LAST of the (ascending) rows -- the same winner
        for k, a_idx, rows in picks:
            if not rows.size:
                continue
            n_, j_, i_ = img[rows], gj[rows], gi[rows]
            target[n_, k, 0, j_, i_] = gx[rows] * f32(grid) - gi[rows].astype(f32)
            target[n_, k, 1, j_, i_] = gy[rows] * f32(grid) - gj[rows].astype(f32)
            target[n_, k, 2, j_, i_] = np.log(gw[rows] * f32(size) / f32(an[a_idx][0]))
            target[n_, k, 3, j_, i_] = np.log(gh[rows] * f32(size) / f32(an[a_idx][1]))
            target[n_, k, 4, j_, i_] = f32(2.0) - gw[rows] * gh[rows]
            target[n_, k, 5, j_, i_] = score[rows]
            target[n_, k, 6 + cls[rows], j_, i_] = 1.
        out.append(target)
    return out


def synth_ground_truth(N, seed, max_boxes=50):
    """Deterministic synthetic ground truth in the reader's format (bench / smoke: there is no dataset here)."""
    rng = np.random.RandomState(seed)
    gt_bbox = np.zeros((N, max_boxes, 4), np.float32)
    gt_class = np.zeros((N, max_boxes), np.int32)
    gt_score = np.zeros((N, max_boxes), np.float32)
    for n in range(N):
        k = 3 + (n % 6)
        gt_bbox[n, :k] = np.concatenate([rng.uniform(0.15, 0.85, size=(k, 2)), rng.uniform(0.04, 0.6, size=(k, 2))], 1)
        gt_class[n, :k] = rng.randint(0, 80, size=k)
        gt_score[n, :k] = np.where(rng.rand(k) < 0.3, rng.uniform(0.3, 0.9, size=k), 1.0)
    return gt_bbox, gt_class, gt_score
