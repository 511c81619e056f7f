"""The eval harness's on-disk record format: this repo's `tools.cocotools` writer against
files written by the reference's own `multi_thread_write_json` (tests/golden/g10_coco_records.npz, made from
/root/reference by tools/make_goldens.py g10) -- byte for byte."""
import json
import os

import numpy as np
import pytest

from tools import cocotools


def _cases(golden):
    g = golden('g10_coco_records')
    for j in range(int(g['ncases'])):
        yield (g['boxes%d' % j], g['scores%d' % j], g['classes%d' % j], int(g['im_id%d' % j]), bytes(g['name%d' % j]).decode(),
               bytes(g['json%d' % j]).decode())


def test_category_table(golden):
    g = golden('g10_coco_records')
    assert [cocotools.clsid2catid[i] for i in range(80)] == g['clsid2catid'].tolist()
    assert all(cocotools.catid2clsid[c] == i for i, c in cocotools.clsid2catid.items()) and len(cocotools.catid2clsid) == 80


def test_record_files_match_reference_bytes(golden, tmp_path):
    n = 0
    for boxes, scores, classes, im_id, name, want in _cases(golden):
        path = cocotools.write_bbox_json(str(tmp_path), name, boxes, scores, classes, im_id)
        assert os.path.basename(path) == name.split('.')[0] + '.json'
        assert open(path).read() == want, name
        recs = json.loads(want)
        assert recs == cocotools.bbox_records(boxes, scores, classes, im_id)
        n += len(recs)
    assert n > 100


def test_batch_writer_and_missing_image(golden, tmp_path):
    cs = list(_cases(golden))
    paths = cocotools.write_batch(str(tmp_path), [c[0] for c in cs] + [None], [c[1] for c in cs] + [None],
                                  [c[2] for c in cs] + [None], [c[3] for c in cs] + [1], [c[4] for c in cs] + ['none.jpg'])
    assert paths[-1] is None and not os.path.exists(str(tmp_path / 'bbox' / 'none.json'))
    for p, c in zip(paths, cs):
        assert open(p).read() == c[5]


def test_rounding_is_the_references():
    """w = xmax - xmin + 1 in float32, then round(float(v) * 10) / 10 (half to even on the float64 product)."""
    b = np.array([[0.25, 0.35, 0.45, 0.55]], dtype=np.float32)
    r = cocotools.bbox_records(b, np.array([0.5], np.float32), np.array([3], np.int32), 9)[0]
    assert r == {'image_id': 9, 'category_id': 4, 'bbox': [0.2, 0.3, 1.2, 1.2], 'score': 0.5}
    w32 = float(np.float32(0.45) - np.float32(0.25) + 1)
    assert r['bbox'][2] == round(w32 * 10) / 10


@pytest.mark.gpu
def test_detect_batch_to_records_end_to_end(golden, tmp_path):
    """Decode.detect_batch on the GPU -> record files: same files as the reference's harness wrote for the same inputs
    (g9 inputs, g10 files), up to the score / box tolerance of the path (scores 1e-4 -> compared as parsed numbers)."""
    import torch
    from conftest import build_model
    from config import PPYOLO_r18vd_Config
    from model.decode_np import Decode
    from ppyolo_hip import synth
    g9 = golden('g9_decode_harness')
    S, N, wseed, iseed = [int(v) for v in g9['meta']]
    cfg = PPYOLO_r18vd_Config()
    model, _ = build_model(cfg, wseed, 'cuda')
    dec = Decode(model, ['c%d' % i for i in range(80)], True, cfg, for_test=True)
    x = synth.synth_images(N, S, seed=iseed).numpy()
    _, bb, ss, cc = dec.detect_batch([None] * N, x, g9['im_size'], draw_image=False)
    cs = list(_cases(golden))[:N]
    paths = cocotools.write_batch(str(tmp_path), bb, ss, cc, [c[3] for c in cs], [c[4] for c in cs])
    for p, c in zip(paths, cs):
        got, want = json.load(open(p)), json.loads(c[5])
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a['image_id'] == b['image_id'] and a['category_id'] == b['category_id']
            assert abs(a['score'] - b['score']) <= 1e-4 and max(abs(u - v) for u, v in zip(a['bbox'], b['bbox'])) <= 0.1 + 1e-9
