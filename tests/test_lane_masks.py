"""Host logic of the lanes' CU partition (ppyolo_hip/runtime.py::lane_cu_masks) -- no device needed.  The bit order
it assumes (mask bit i = CU i // 8 of XCD i % 8) is what tools/probes/cu_mask_probe.hip measured on MI355X
(profiles/r05_cu_mask_probe.txt)."""
import pytest

from ppyolo_hip._lib import PPYoloHipError
from ppyolo_hip.runtime import lane_cu_masks


def bits(words):
    return {32 * j + b for j, w in enumerate(words) for b in range(32) if w >> b & 1}


def test_off_means_no_masks():
    for spec in ('', '0', 'off', None):
        assert lane_cu_masks(spec, 2, 256) == [None, None]


def test_half_gives_every_lane_an_equal_share_of_every_xcd():
    a, b = lane_cu_masks('half', 2, 256)
    A, B = bits(a), bits(b)
    assert len(A) == len(B) == 128 and not (A & B) and (A | B) == set(range(256))
    for xcd in range(8):
        assert len({i for i in A if i % 8 == xcd}) == 16 and len({i for i in B if i % 8 == xcd}) == 16
    q = lane_cu_masks('half', 4, 256)
    assert [len(bits(m)) for m in q] == [64] * 4 and len(set().union(*[bits(m) for m in q])) == 256


def test_xcd_form_and_terms():
    a, b = lane_cu_masks('xcd', 2, 256)
    assert bits(a) == {i for i in range(256) if i % 8 < 4} and bits(b) == {i for i in range(256) if i % 8 >= 4}
    assert lane_cu_masks('m8:0-3|m8:4-7', 2, 256) == [a, b]
    m = lane_cu_masks('all|m256:0-127', 2, 256)
    assert m[0] is None and bits(m[1]) == set(range(128))
    h = lane_cu_masks('h:ff,0,0,0,0,0,0,f0000000|all', 2, 256)
    assert bits(h[0]) == set(range(8)) | {252, 253, 254, 255}


def test_bad_specs_are_refused():
    for spec, depth in (('m8:0-3', 2), ('m8:4-9|all', 2), ('x|y', 2), ('m300:290-299|all', 2), ('xcd', 3), ('h:0,0|all', 2), ('h:zz|all', 2)):
        with pytest.raises(PPYoloHipError):
            lane_cu_masks(spec, depth, 256)
