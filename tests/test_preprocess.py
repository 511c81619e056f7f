"""Image pre-processing in front of the path: the numpy oracle of Decode.process_image against
independent definitions (CPU), and the HIP kernel against the oracle, bit for bit (GPU).  cv2 is not installed, so
the resize part of the oracle is "parity unpinned" (oracle/preprocess_oracle.py header)."""
import numpy as np
import pytest
import torch

from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from oracle import preprocess_oracle as pre


def rand_image(h, w, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(max(1, h // 7 + 1), max(1, w // 5 + 1), 3)).astype(np.uint8)
    img = np.kron(base, np.ones((7, 5, 1), dtype=np.uint8))[:h, :w]           # blocks: edges and flat areas
    noise = rng.randint(-20, 21, size=img.shape)
    return np.clip(img.astype(np.int32) + noise, 0, 255).astype(np.uint8)


def bicubic_float64(img, S):
    """Independent definition: bicubic convolution, A = -0.75, half-pixel centres, replicated border, float64."""
    h, w, _ = img.shape

    def taps(n_src, n_dst):
        scale = n_src / float(n_dst)
        c = (np.arange(n_dst) + 0.5) * scale - 0.5
        s = np.floor(c).astype(np.int64)
        t = c - s
        A = -0.75
        k0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
        k1 = ((A + 2) * t - (A + 3)) * t * t + 1
        k2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
        return s, np.stack([k0, k1, k2, 1 - k0 - k1 - k2], -1)
    sx, kx = taps(w, S)
    sy, ky = taps(h, S)
    src = img.astype(np.float64)
    hor = sum(src[:, np.clip(sx + j - 1, 0, w - 1), :] * kx[:, j][None, :, None] for j in range(4))
    return sum(hor[np.clip(sy + k - 1, 0, h - 1)] * ky[:, k][:, None, None] for k in range(4))


def test_identity_size_is_exact():
    img = rand_image(64, 64, 0)
    assert np.array_equal(pre.resize_cubic_u8(img, 1.0, 1.0), img)


@pytest.mark.parametrize('h,w,S', [(48, 64, 96), (97, 131, 64), (33, 20, 32), (5, 200, 64), (1, 1, 32), (2, 3, 32)])
def test_fixed_point_resize_tracks_the_float_definition(h, w, S):
    img = rand_image(h, w, h * 1000 + w)
    got = pre.resize_cubic_u8(img, float(S) / w, float(S) / h)
    assert got.shape == (S, S, 3) and got.dtype == np.uint8
    want = np.clip(np.rint(bicubic_float64(img, S)), 0, 255)
    # 11-bit weights: at most one grey level away from the exactly rounded float result, and rarely
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.05


def test_weights_are_opencv_fixed_point():
    s, a = pre.axis_tables(480, 608, 608.0 / 480.0)
    assert a.dtype == np.int16 and a.shape == (608, 4)
    assert np.abs(a.astype(np.int32).sum(1) - 2048).max() <= 2           # rounded, not renormalised (as OpenCV)
    assert s[0] == -1 and s[-1] == 479                                    # (0.5 * 480/608 - 0.5) < 0; last centre 479.1
    # interpolateCubic(0) = (0, 1, 0, 0), interpolateCubic(0.5) symmetric
    assert pre.cubic_coeffs(np.float32(0)).tolist() == [0.0, 1.0, 0.0, 0.0]
    c = pre.cubic_coeffs(np.float32(0.5))
    assert c[0] == c[3] and c[1] == c[2] and abs(float(c.sum()) - 1.0) < 1e-6


def test_normalisation_table_is_the_numpy_expression():
    from ppyolo_hip.preprocess import normalisation_table
    cfg = PPYOLO_2x_Config()
    n = cfg.normalizeImage
    lut = normalisation_table(n['mean'], n['std'], n['is_scale'])
    assert lut.shape == (3, 256) and lut.dtype == np.float32
    levels = np.tile(np.arange(256, dtype=np.uint8)[:, None, None], (1, 4, 3))
    want = pre.normalize(levels, n['mean'], n['std'], n['is_scale'])        # [256,4,3]
    for c in range(3):
        assert np.array_equal(lut[c], want[:, 0, c])


def test_normalise_permute_pinned_to_reference_golden():
    """tests/golden/g8_preprocess.npz = the reference's own NormalizeImage + Permute classes on a uint8 image
    (tools/make_goldens.py g8): the oracle restates them bit for bit, and so does the product's table -- every grey
    level occurs in every channel of the fixture.  (The cv2.resize in front stays unpinned: no cv2 here.)"""
    import os
    from ppyolo_hip.preprocess import normalisation_table
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g8_preprocess.npz'))
    img, want = g['image'], g['normalized_chw']
    mean, std = g['mean'].tolist(), g['std'].tolist()
    got = pre.normalize(img.copy(), mean, std, True)
    got = np.swapaxes(np.swapaxes(got, 1, 2), 1, 0)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    lut = normalisation_table(mean, std, True)
    for c in range(3):
        assert np.array_equal(lut[c][img[:, :, c]], want[c])
    # the reference configuration the fixture was made with is the one the product configs carry
    cfg = PPYOLO_2x_Config()
    assert cfg.normalizeImage['mean'] == mean and cfg.normalizeImage['std'] == std
    assert int(g['to_rgb']) == int(cfg.decodeImage['to_rgb']) and int(g['interp']) == cfg.resizeImage['interp'] == 2
    assert int(g['target_size']) == cfg.test_cfg['target_size']


def test_process_image_layout():
    img = rand_image(60, 80, 3)
    pimage, im_size = pre.process_image(img, 64)
    assert pimage.shape == (1, 3, 64, 64) and pimage.dtype == np.float32
    assert im_size.tolist() == [[60, 80]] and im_size.dtype == np.int32
    # channel 0 of the output is R = channel 2 of the BGR input (decode_np.py:126-127)
    flat = np.zeros((60, 80, 3), dtype=np.uint8)
    flat[:, :, 2] = 255
    p, _ = pre.process_image(flat, 64)
    assert p[0, 0].min() > 2.0 and p[0, 1].max() < -1.9 and p[0, 2].max() < -1.7


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['tile', 'pixel'])
@pytest.mark.parametrize('to_rgb', [True, False])
def test_kernel_matches_oracle_bitwise(to_rgb, kernel, monkeypatch):
    """Both forms of the kernel -- the separable, tiled one of round 5 (default) and the one-thread-per-pixel form (PPY_PRE_PIXEL=1,
    read per call) -- against the numpy oracle, bit for bit; S = 608 / 416 (multiples of 4: 16-byte stores), 64, and 333 / 37 (scalar
    stores, a tile wider than the image)."""
    monkeypatch.setenv('PPY_PRE_PIXEL', '1' if kernel == 'pixel' else '0')
    from ppyolo_hip import ops
    from ppyolo_hip.preprocess import normalisation_table
    cfg = PPYOLO_2x_Config()
    n = cfg.normalizeImage
    lut = torch.from_numpy(normalisation_table(n['mean'], n['std'], n['is_scale'])).cuda()
    sizes = [(480, 640), (1080, 1920), (37, 53), (1, 1), (3, 500), (416, 416), (375, 500), (2, 3), (600, 13)]
    sizes += [(100 + 7 * i, 90 + 11 * i) for i in range(12)]                 # 21 images: two launches of <= 16
    sizes += [(2400, 31), (31, 2400)]                                        # 4x down: the tile's row count shrinks to fit the LDS
    for S in ((416, 64, 333, 37, 608) if kernel == 'tile' else (416, 64)):
        if S in (333, 37, 608):
            sizes = sizes[:9] + sizes[-2:]
        imgs = [rand_image(h, w, 7 * h + w) for h, w in sizes]
        wide = [torch.from_numpy(np.pad(im, ((0, 0), (0, 5), (0, 0)))).cuda() for im in imgs]
        dev = [t[:, :im.shape[1]] for t, im in zip(wide, imgs)]             # padded rows: row stride > 3 w
        out = torch.full((len(imgs), 3, S, S), float('nan'), device='cuda')
        ops.preprocess_images(dev, S, lut, out, swap_rb=to_rgb)
        got = out.cpu().numpy()
        for i, im in enumerate(imgs):
            want, _ = pre.process_image(im, S, to_rgb=to_rgb, mean=n['mean'], std=n['std'], is_scale=n['is_scale'])
            assert np.array_equal(got[i], want[0]), 'image %d %s -> %d' % (i, im.shape, S)


@pytest.mark.gpu
def test_decode_process_image_and_detect_raw():
    """Decode.process_image keeps the reference's return convention; detect_raw = process_image + predict, all on
    the device, equals predict() on the oracle's pre-processed batch."""
    from conftest import build_model
    from model.decode_np import Decode
    cfg = PPYOLO_r18vd_Config()
    cfg.test_cfg['target_size'] = 320
    model, sd = build_model(cfg, 0, 'cuda')
    dec = Decode(model, ['c%d' % i for i in range(80)], True, cfg, for_test=True)
    imgs = [rand_image(240, 320, 1), rand_image(333, 500, 2), rand_image(500, 375, 3)]
    pimage, im_size = dec.process_image(imgs[0])
    want, want_size = pre.process_image(imgs[0], 320)
    assert isinstance(pimage, np.ndarray) and np.array_equal(pimage, want) and np.array_equal(im_size, want_size)
    res = dec.detect_raw(imgs)
    batch = np.concatenate([pre.process_image(im, 320)[0] for im in imgs])
    sizes = np.array([[im.shape[0], im.shape[1]] for im in imgs], dtype=np.int32)
    ref = dec.predict(batch, sizes)
    for (boxes, scores, classes), p in zip(res, ref):
        b, s, c = dec._split(p)
        assert np.array_equal(boxes, b) and np.array_equal(scores, s) and np.array_equal(classes, c)
