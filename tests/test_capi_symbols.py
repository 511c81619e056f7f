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
    convolution kernels of another stream share the CU (tools/pk_hazard_probe.py), so no kernel here may contain them.
    Checked on the code objects embedded in the SHIPPED .so (it is the file that travels to the GPU box), so this runs
    wherever the library is -- it needs no build directory."""
    from ppyolo_hip import build
    cos = build.device_code_objects(libpath)
    assert len(cos) >= len(build.SOURCES) - 1, 'expected one gfx950 code object per translation unit with kernels'
    assert sum(len(c) for c in cos) > 1 << 20
    assert build.packed_fp32_ops(libpath) == 0


@pytest.mark.gpu
def test_no_packed_fp32_ops_in_loaded_library():
    """The same check on the GPU box, against the very file the process has mapped."""
    from ppyolo_hip import _lib, build
    _lib.lib()
    mapped = [ln.split()[-1] for ln in open('/proc/self/maps') if ln.rstrip().endswith('libppyolo_hip.so')]
    assert mapped and os.path.samefile(mapped[0], _lib.LIB_PATH)
    assert build.packed_fp32_ops(mapped[0]) == 0
