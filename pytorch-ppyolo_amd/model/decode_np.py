"""Host harness boundary -- the part of the reference's `model/decode_np.py` that sits on the
model boundary: `Decode.predict` (:142-150), `detect_image` (:41-57), `detect_batch`
(:81-96), and `process_image` (:125-140) -- the cv2 resize / normalise / permute in front of the
model, here ONE HIP kernel on the device (ppyolo_hip/csrc/preprocess.hip; SURVEY.md section 8f
rank 1).  Drawing (cv2) is host-side post-processing outside the path and is not provided."""
import numpy as np
import torch


class Decode(object):
    def __init__(self, _yolo, all_classes, use_gpu, cfg, for_test=True):
        self.all_classes = all_classes
        self.num_classes = len(all_classes)
        self._yolo = _yolo
        self.use_gpu = use_gpu
        self.cfg = cfg
        # reference :32-37: eval target size unless for_test
        self.target_size = (cfg.test_cfg if for_test else cfg.eval_cfg)['target_size']
        self._pre = None

    def _preprocessor(self):
        if self._pre is None:
            from ppyolo_hip.preprocess import Preprocessor
            self._pre = Preprocessor(self.cfg, self.target_size)
        return self._pre

    def process_image(self, img, to_numpy=True):
        """uint8 BGR [h,w,3] -> (pimage [1,3,S,S] float32, im_size [[h, w]] int32), computed on the device.
        `to_numpy=True` returns numpy arrays like the reference; False keeps pimage on the device (what `predict`
        wants anyway) and saves the round trip."""
        pimage, _ = self._preprocessor()([img])
        im_size = np.array([[img.shape[0], img.shape[1]]]).astype(np.int32)
        return (pimage.cpu().numpy() if to_numpy else pimage), im_size

    def detect_raw(self, images):
        """Throughput form of demo.py's loop: raw uint8 BGR images of any sizes -> per image (boxes, scores, classes),
        pre-processing, forward and Matrix-NMS all on the device."""
        pimage, im_size = self._preprocessor()(images)
        preds = self._yolo(pimage, im_size)
        return [self._split(p.cpu().detach().numpy()) for p in preds]

    def predict(self, image, im_size):
        """numpy [N,3,S,S] f32 + numpy [N,2] (h, w) -> list of numpy [K,6] f32."""
        if not isinstance(image, torch.Tensor):
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
