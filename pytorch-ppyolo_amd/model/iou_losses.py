"""IoU loss settings (reference model/iou_losses.py:16-246).

The reference's classes compute the loss with autograd ops; here they only HOLD the settings: the IoU loss, the IoU-aware
loss and their gradients are part of the fused loss kernel (ppyolo_hip/csrc/yolo_loss.hip, ppy_yolov3_loss_f32), which the
training forward of `PPYOLO.forward(..., eval=False, ...)` runs.  Same constructor arguments as the reference, so that its
train.py:241-247 builds them unchanged."""


class IouLoss(object):
    def __init__(self, loss_weight=2.5, max_height=608, max_width=608, ciou_term=False, loss_square=True):
        if ciou_term:
            # (the reference's own branch cannot train either: get_ciou_term sets `alpha.requires_grad = False` on a non-leaf tensor,
            #  model/iou_losses.py:131, which PyTorch refuses with a RuntimeError as soon as the head outputs require gradients)
            raise NotImplementedError('ciou_term=True is not part of the fused loss kernel (no PP-YOLO configuration uses it; reference '
                                      'model/iou_losses.py:95-133)')
        self._loss_weight = loss_weight
        self._MAX_HI, self._MAX_WI = max_height, max_width      # (held, unused: as in the reference)
        self.ciou_term, self.loss_square = ciou_term, loss_square


class IouAwareLoss(IouLoss):
    def __init__(self, loss_weight=1.0, max_height=608, max_width=608):
        super(IouAwareLoss, self).__init__(loss_weight=loss_weight, max_height=max_height, max_width=max_width)
