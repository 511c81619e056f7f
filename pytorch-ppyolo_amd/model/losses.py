"""YOLOv3 loss settings (reference model/losses.py:84-241).

A holder: `YOLOv3Head(yolo_loss=YOLOv3Loss(iou_loss=..., iou_aware_loss=..., **cfg.yolo_loss), ...)` as the reference's
train.py:241-249 builds it.  The arithmetic -- the fine-grained loss of `_get_fine_grained_loss`, which is the only branch
the reference's `__call__` takes, whatever `use_fine_grained_loss` / `label_smooth` say (:113-117) -- and its backward are one
HIP kernel per head level (ppyolo_hip/csrc/yolo_loss.hip); oracle/train_oracle.py restates it line by line."""


class YOLOv3Loss(object):
    def __init__(self, ignore_thresh=0.7, label_smooth=True, use_fine_grained_loss=False, iou_loss=None, iou_aware_loss=None,
                 downsample=[32, 16, 8], scale_x_y=1., match_score=False):
        if iou_loss is None:
            raise NotImplementedError('the fused loss kernel always evaluates the IoU loss (both PP-YOLO configurations set one)')
        self._ignore_thresh = ignore_thresh
        self._label_smooth = label_smooth
        self._use_fine_grained_loss = use_fine_grained_loss
        self._iou_loss = iou_loss
        self._iou_aware_loss = iou_aware_loss
        self.downsample = downsample
        self.scale_x_y = scale_x_y
        self.match_score = match_score

    def __call__(self, *a, **k):
        raise NotImplementedError('settings holder: the loss runs inside PPYOLO.forward(..., eval=False, ...) (ppyolo_hip.train)')
