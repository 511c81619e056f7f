"""Host side of the on-device image pre-processing (reference Decode.process_image, model/decode_np.py:125-140).

The resize runs in the HIP kernel (csrc/preprocess.hip).  The numpy normalisation of the reference
(tools/transform.py:895-917: `im.astype(float32) / 255.0`, `im -= mean`, `im /= std` with float64 mean / std) is a
function of (grey level, channel); it is tabulated here with that very numpy expression, so the kernel's output is the
reference's float32 bit for bit whatever numpy's promotion rules do."""
import numpy as np
import torch

from . import ops
from ._lib import PPYoloHipError


def normalisation_table(mean, std, is_scale=True):
    """[3,256] float32: value of grey level v in output channel c."""
    im = np.tile(np.arange(256, dtype=np.uint8)[:, None, None], (1, 1, 3))        # "image" of 256 x 1 pixels, HWC
    im = im.astype(np.float32, copy=False)
    m = np.array(list(mean))[np.newaxis, np.newaxis, :]
    s = np.array(list(std))[np.newaxis, np.newaxis, :]
    if is_scale:
        im = im / 255.0
    im -= m
    im /= s
    return np.ascontiguousarray(im[:, 0, :].T.astype(np.float32))


class Preprocessor(object):
    """cfg-driven (`decodeImage.to_rgb`, `normalizeImage`, `permute`, target size) pre-processing of raw uint8 HWC
    images into the [n,3,S,S] float32 batch `PPYOLO.forward` takes, without leaving the device."""

    def __init__(self, cfg, target_size, device='cuda'):
        n = cfg.normalizeImage
        if n.get('is_channel_first', False) or not cfg.permute.get('channel_first', True) or cfg.permute.get('to_bgr', False):
            raise PPYoloHipError('only the reference inference layout is implemented: HWC normalise, CHW output, RGB')
        if cfg.resizeImage.get('interp', 2) != 2:
            raise PPYoloHipError('only cv2.INTER_CUBIC (interp=2), the reference configs\' setting, is implemented')
        self.S = int(target_size)
        self.to_rgb = bool(cfg.decodeImage['to_rgb'])
        self.device = torch.device(device)
        self.lut = torch.from_numpy(normalisation_table(n['mean'], n['std'], n.get('is_scale', True))).to(self.device)

    def upload(self, img):
        """numpy / torch uint8 [h,w,3] -> device tensor (pinned staging is the caller's business)."""
        t = torch.as_tensor(np.ascontiguousarray(img)) if not isinstance(img, torch.Tensor) else img
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise PPYoloHipError('expected a uint8 image [h, w, 3], got %s %s' % (t.dtype, tuple(t.shape)))
        return t.to(self.device, non_blocking=True).contiguous()

    def __call__(self, images, out=None):
        """images: list of uint8 [h,w,3] arrays / tensors (BGR as cv2.imread gives them).  Returns
        (pimage [n,3,S,S] float32 on the device, im_size [n,2] float32 (h, w) on the device)."""
        dev = [self.upload(im) for im in images]
        n = len(dev)
        if out is None:
            out = torch.empty((n, 3, self.S, self.S), dtype=torch.float32, device=self.device)
        ops.preprocess_images(dev, self.S, self.lut, out, swap_rb=self.to_rgb)
        im_size = torch.tensor([[d.shape[0], d.shape[1]] for d in dev], dtype=torch.float32).to(self.device, non_blocking=True)
        return out, im_size
