"""YOLO training targets from ground-truth boxes -- the data format on the near side of the training step (SURVEY.md
section 8f rank 2): the reference's `Gt2YoloTarget` (tools/transform.py:1211-1316), which its reader threads run on the host
for every batch before `train.py:416-426` hands `target0..2` to the model.  Host-side numpy here as well (a few dozen boxes per
image); pinned bit for bit to targets the reference's own operator produced (tests/golden/g12_train_*.npz).

Per ground-truth box (normalised cx, cy, w, h; class; mixup score): the best of ALL anchors by IoU of (w, h) alone; if that
anchor belongs to this level, cell (int(cx * grid), int(cy * grid)) of it gets
    tx, ty = fractional cell offsets; tw, th = log(box side in px / anchor side); tscale = 2 - w * h; tobj = score; one-hot class.
"""
import numpy as np


def _wh_iou(w0, h0, w1, h1):
    # jaccard_overlap of two boxes anchored at the origin (reference tools/transform.py:1192-1208)
    if 0. >= w1 or w0 <= 0. or 0. >= h1 or h0 <= 0.:
        return 0
    inter = (min(w0, w1) - 0.) * (min(h0, h1) - 0.)
    return inter / (w0 * h0 + w1 * h1 - inter)


def gt2yolo_target(gt_bbox, gt_class, gt_score, anchors, anchor_masks, downsample_ratios, num_classes, image_size, iou_thresh=1.):
    """gt_bbox [N, G, 4], gt_class [N, G], gt_score [N, G] (zero rows = padding) -> list over levels of float32
    [N, len(mask), 6 + num_classes, grid, grid]."""
    gt_bbox, gt_class, gt_score = np.asarray(gt_bbox), np.asarray(gt_class), np.asarray(gt_score)
    h = w = int(image_size)
    an_hw = np.array(anchors) / np.array([[w, h]])
    out = []
    for mask, ds in zip(anchor_masks, downsample_ratios):
        grid_h, grid_w = int(h / ds), int(w / ds)
        target = np.zeros((gt_bbox.shape[0], len(mask), 6 + num_classes, grid_h, grid_w), dtype=np.float32)
        for n in range(gt_bbox.shape[0]):
            for b in range(gt_bbox.shape[1]):
                gx, gy, gw, gh = gt_bbox[n, b, :]
                cls, score = gt_class[n, b], gt_score[n, b]
                if gw <= 0. or gh <= 0. or score <= 0.:
                    continue
                best_iou, best_idx = 0., -1
                for an_idx in range(an_hw.shape[0]):
                    iou = _wh_iou(gw, gh, an_hw[an_idx, 0], an_hw[an_idx, 1])
                    if iou > best_iou:
                        best_iou, best_idx = iou, an_idx
                gi, gj = int(gx * grid_w), int(gy * grid_h)

                def fill(k, a_idx):
                    target[n, k, 0, gj, gi] = gx * grid_w - gi
                    target[n, k, 1, gj, gi] = gy * grid_h - gj
                    target[n, k, 2, gj, gi] = np.log(gw * w / anchors[a_idx][0])
                    target[n, k, 3, gj, gi] = np.log(gh * h / anchors[a_idx][1])
                    target[n, k, 4, gj, gi] = 2.0 - gw * gh
                    target[n, k, 5, gj, gi] = score
                    target[n, k, 6 + cls, gj, gi] = 1.
                if best_idx in mask:
                    fill(mask.index(best_idx), best_idx)
                if iou_thresh < 1:
                    for k, mask_i in enumerate(mask):
                        if mask_i != best_idx and _wh_iou(gw, gh, an_hw[mask_i, 0], an_hw[mask_i, 1]) > iou_thresh:
                            fill(k, mask_i)
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
