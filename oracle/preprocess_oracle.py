"""CPU oracle for the image pre-processing in front of the hot path  --  TEST INFRASTRUCTURE ONLY.

Restates `Decode.process_image` of miemie2013/Pytorch-PPYOLO (reference model/decode_np.py:125-140):

    cv2.cvtColor(BGR2RGB)                                   decode_np.py:126-127
    ResizeImage(target_size=S, interp=2 = cv2.INTER_CUBIC)  tools/transform.py:955-1003 (fx = S/w, fy = S/h, uint8 in/out)
    NormalizeImage(mean, std, is_scale=True, HWC)           tools/transform.py:895-917
    Permute(to_bgr=False, channel_first=True)               tools/transform.py:1040-1060

Nothing in the product (`pytorch-ppyolo_amd/`) may import this file; allowed importers are `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.

PINNING.  NormalizeImage + Permute are pinned to the reference's own classes (fixture tests/golden/g8_preprocess.npz,
made by tools/make_goldens.py g8 from /root/reference; tests/test_preprocess.py replays it).  The RESIZE IS UNPINNED:
it is third-party arithmetic, OpenCV (`cv2.resize`), which is NOT installed in this image and
which the reference pins to no version (requirements: "opencv-python").  What follows restates OpenCV 4.x's published
algorithm for 8-bit INTER_CUBIC (modules/imgproc/src/resize.cpp: `interpolateCubic` with A = -0.75, the coordinate map
(dx + 0.5) * scale - 0.5, replicated borders, coefficients rounded to 11-bit fixed point, `HResizeCubic<uchar,int,short>`
followed by `VResizeCubic<..., FixedPtCast<int, uchar, 22>>`).  OpenCV's vectorised vertical pass
(`VResizeCubicVec_32s8u`) evaluates the same sum in float32 and can differ from the fixed-point cast by one grey level
on a few pixels in 10^4; without cv2 neither variant can be checked here, hence "unpinned".  Everything after the resize
is numpy arithmetic and is restated operation for operation (float32 division by 255, float64 subtract / divide rounded
back to float32 by the in-place ops).
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def cubic_coeffs(x):
    """OpenCV `interpolateCubic` (resize.cpp), float32 operation by operation (no fused multiply-add)."""
    f = np.float32
    x = np.asarray(x, dtype=np.float32)
    A = f(-0.75)
    t = x + f(1)
    c0 = ((A * t - f(5) * A) * t + f(8) * A) * t - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    u = f(1) - x
    c2 = ((A + f(2)) * u - (A + f(3))) * u * u + f(1)
    c3 = f(1) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def axis_tables(src, dst, inv_scale):
    """Per destination index: first source tap (s - 1 ... s + 2 are used) and the four int16 fixed-point weights."""
    scale = 1.0 / inv_scale                                            # double, as in cv::resize
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)                  # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(fx).astype(np.int32)                                  # cvFloor
    fx = fx - s.astype(np.float32)
    w = np.rint(cubic_coeffs(fx) * np.float32(INTER_RESIZE_COEF_SCALE))         # saturate_cast<short>: cvRound, ties to even
    return s, np.clip(w, -32768, 32767).astype(np.int16)


def resize_cubic_u8(img, fx, fy):
    """cv2.resize(img, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC) for a uint8 HWC image."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    dw, dh = int(np.rint(w * fx)), int(np.rint(h * fy))               # saturate_cast<int>(ssize.width * inv_scale_x)
    sx, ax = axis_tables(w, dw, fx)
    sy, ay = axis_tables(h, dh, fy)
    src = img.astype(np.int32)
    # horizontal pass on every source row: taps sx-1 .. sx+2, indices clamped to the row (replicated border)
    hor = np.zeros((h, dw, img.shape[2]), dtype=np.int64)
    for j in range(4):
        idx = np.clip(sx + (j - 1), 0, w - 1)
        hor += src[:, idx, :] * ax[:, j].astype(np.int64)[None, :, None]
    # vertical pass: rows sy-1 .. sy+2 clamped; FixedPtCast<int, uchar, 22>
    acc = np.zeros((dh, dw, img.shape[2]), dtype=np.int64)
    for k in range(4):
        idy = np.clip(sy + (k - 1), 0, h - 1)
        acc += hor[idy, :, :] * ay[:, k].astype(np.int64)[:, None, None]
    acc = acc.astype(np.int32).astype(np.int64)                        # the C code accumulates in int (no overflow in range)
    out = (acc + (1 << (2 * INTER_RESIZE_COEF_BITS - 1))) >> (2 * INTER_RESIZE_COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def normalize(im_u8_hwc, mean, std, is_scale=True):
    """NormalizeImage.__call__ with is_channel_first=False (transform.py:895-917), numpy semantics kept:
    float32 / python float stays float32; `im -= mean`, `im /= std` run in float64 and round back to float32."""
    im = im_u8_hwc.astype(np.float32, copy=False)
    mean = np.array(mean)[np.newaxis, np.newaxis, :]
    std = np.array(std)[np.newaxis, np.newaxis, :]
    if is_scale:
        im = im / 255.0
    im -= mean
    im /= std
    return im


def process_image(img_bgr_u8, target_size, to_rgb=True, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                  is_scale=True):
    """Decode.process_image (decode_np.py:125-140) -> (pimage [1,3,S,S] float32, im_size [[h, w]] int32)."""
    img = img_bgr_u8[:, :, ::-1] if to_rgb else img_bgr_u8
    h, w = img.shape[0], img.shape[1]
    im = resize_cubic_u8(np.ascontiguousarray(img), float(target_size) / float(w), float(target_size) / float(h))
    im = normalize(im, list(mean), list(std), is_scale)
    im = np.swapaxes(np.swapaxes(im, 1, 2), 1, 0)                      # Permute channel_first (transform.py:1052-1054)
    pimage = np.expand_dims(im, axis=0)
    im_size = np.array([[h, w]]).astype(np.int32)
    return np.ascontiguousarray(pimage), im_size
