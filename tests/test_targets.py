"""`Gt2YoloTarget` (reference tools/transform.py:1211-1316): this repo's host-side restatement against the targets the
reference's own operator produced for the g12 training fixtures -- bit for bit."""
import numpy as np
import pytest

from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth


@pytest.mark.parametrize('tag,cfgc', [('r18vd_96', PPYOLO_r18vd_Config), ('r50vd_96', PPYOLO_2x_Config)])
def test_targets_equal_the_references(golden, tag, cfgc):
    g = golden('g12_train_' + tag)
    cfg = cfgc()
    hc = cfg.head
    S = int(g['meta'][0])
    mine = gt2yolo_target(g['gt_bbox'], g['gt_class'], g['gt_score'], hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)
    assert len(mine) == len(hc['anchor_masks'])
    n_pos = 0
    for i, t in enumerate(mine):
        want = g['target%d' % i]
        assert t.dtype == want.dtype and t.shape == want.shape and np.array_equal(t, want), 'level %d' % i
        n_pos += int((t[:, :, 5] > 0).sum())
    assert n_pos >= 5


def test_every_box_lands_in_exactly_one_level():
    cfg = PPYOLO_2x_Config()
    hc = cfg.head
    bb, cc, ss = synth_ground_truth(8, 3)
    tg = gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, 608)
    assert [t.shape for t in tg] == [(8, 3, 86, 19, 19), (8, 3, 86, 38, 38), (8, 3, 86, 76, 76)]
    per_image = sum((t[:, :, 5] > 0).reshape(8, -1).sum(1) for t in tg)
    n_boxes = (ss > 0).sum(1)
    assert (per_image <= n_boxes).all() and (per_image >= 1).all()      # (boxes that share a cell and an anchor overwrite each other)
    for t in tg:
        pos = t[:, :, 5] > 0
        assert np.all(t[:, :, 6:].sum(2)[pos] >= 1) and np.all((t[:, :, 0][pos] >= 0) & (t[:, :, 0][pos] < 1))
