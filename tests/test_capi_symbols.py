"""The C-ABI library builds for gfx950, loads on a CPU-only box and exports every entry
point include/ppyolo_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    import __graft_entry__ as ge
    ge.build()
    from ppyolo_hip import _lib
    return _lib.LIB_PATH


def declared():
    text = open(os.path.join(ROOT, 'include', 'ppyolo_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ppy_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), 'missing export %s' % n


def test_binding_covers_header(libpath):
    from ppyolo_hip import _lib
    assert _lib.exported_symbols() == declared()
    assert _lib.lib().ppy_version() >= 100
    assert _lib.lib().ppy_error_string(-3).decode().startswith('workspace')


def test_pick_is_pure_host_logic(libpath):
    from ppyolo_hip import ops
    cfg, split = ops.conv2d_pick(8, 19, 19, 2048, 512, 1, 1, 1, 0)
    assert 0 <= cfg < 67 and split >= 1
    # bad geometry is rejected, not crashed on
    from ppyolo_hip._lib import PPYoloHipError
    with pytest.raises(PPYoloHipError):
        ops.conv2d_pick(8, 19, 19, 30, 512, 1, 1, 1, 0)       # C % 32 != 0


def test_no_packed_fp32_ops_in_device_code(libpath):
    """ppyolo_hip/build.py: v_pk_{add,mul,fma}_f32 of one kernel are corrupted on MI355X while the 16-bit-MFMA
    convolution kernels of another stream share the CU (tools/pk_hazard_probe.py), so no kernel here may contain them."""
    from ppyolo_hip import build
    objdir = os.path.join(os.path.dirname(libpath), 'obj')
    objs = sorted(f for f in os.listdir(objdir) if f.endswith('.o')) if os.path.isdir(objdir) else []
    if len(objs) < len(build.SOURCES):
        pytest.skip('object files of the build are not here (prebuilt library)')
    for f in objs:
        assert build.packed_fp32_ops(os.path.join(objdir, f)) == 0, f
