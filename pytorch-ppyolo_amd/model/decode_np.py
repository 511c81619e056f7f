"""Host harness boundary -- the part of the reference's `model/decode_np.py` that sits on the
model boundary: `Decode.predict` (:142-150), `detect_image` (:41-57), `detect_batch`
(:81-96).  Image pre-processing (`process_image`, cv2 resize/normalise) and drawing are host
CPU work outside the hot path and are not
provided; callers pass already pre-processed `pimage` arrays exactly as the reference's
`predict` receives them."""
import numpy as np
import torch


class Decode(object):
    def __init__(self, _yolo, all_classes, use_gpu, cfg, for_test=True):
        self.all_classes = all_classes
        self.num_classes = len(all_classes)
        self._yolo = _yolo
        self.use_gpu = use_gpu
        self.cfg = cfg

    def predict(self, image, im_size):
        """numpy [N,3,S,S] f32 + numpy [N,2] (h, w) -> list of numpy [K,6] f32."""
        image = torch.as_tensor(np.asarray(image), dtype=torch.float32)
        im_size = torch.as_tensor(np.asarray(im_size), dtype=torch.float32)
        if not self.use_gpu:
            raise RuntimeError('the MI355X path has no CPU mode: construct Decode(use_gpu=True)')
        image, im_size = image.cuda(), im_size.cuda()
        preds = self._yolo(image, im_size)
        return [p.cpu().detach().numpy() for p in preds]

    @staticmethod
    def _split(pred):
        if pred[0][0] < 0.0:                      # the [[-1]*6] "no detection" row
            return np.array([]), np.array([]), np.array([])
        return pred[:, 2:], pred[:, 1], pred[:, 0].astype(np.int32)

    def detect_image(self, image, pimage, im_size, draw_image, draw_thresh=0.0):
        if draw_image:
            raise NotImplementedError('drawing (cv2) is host-side post-processing outside the hot path')
        pred = self.predict(pimage, im_size)
        boxes, scores, classes = self._split(pred[0])
        return image, boxes, scores, classes

    def detect_batch(self, batch_img, batch_pimage, batch_im_size, draw_image, draw_thresh=0.0):
        if draw_image:
            raise NotImplementedError('drawing (cv2) is host-side post-processing outside the hot path')
        pred = self.predict(batch_pimage, batch_im_size)
        res = [self._split(p) for p in pred]
        return (list(batch_img), [r[0] for r in res], [r[1] for r in res], [r[2] for r in res])
