"""Which torch operations work on a CU-masked external stream (round 5 debugging)?"""
import os, sys, ctypes, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')]
import torch
from ppyolo_hip import runtime, _lib
print('torch', torch.__version__, 'hip', torch.version.hip)
for ln in open('/proc/self/maps'):
    if 'amdhip64' in ln and 'r-xp' in ln:
        print('mapped:', ln.split()[-1])
dev = torch.device('cuda', 0)
a = torch.ones(1 << 20, device=dev)
torch.cuda.synchronize()
masks = runtime.lane_cu_masks('half', 2, 256)
ms = runtime._MaskedStream(dev, masks[0])
print('stream ptr', hex(ms.ptr))
def attempt(name, fn):
    try:
        with torch.cuda.stream(ms.stream):
            fn()
        ms.stream.synchronize()
        print('ok  ', name)
    except Exception as exc:
        print('FAIL', name, type(exc).__name__, str(exc).splitlines()[0])
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
attempt('mul_ (elementwise kernel)', lambda: a.mul_(2.0))
attempt('zero_ (memset)', lambda: a.zero_())
attempt('fill_ (kernel)', lambda: a.fill_(3.0))
attempt('copy_ d2d', lambda: a.copy_(torch.ones_like(a)))
b = torch.ones(1 << 20).pin_memory()
attempt('copy_ h2d', lambda: a.copy_(b, non_blocking=True))
attempt('event record', lambda: torch.cuda.Event().record(ms.stream))
attempt('wait_stream', lambda: ms.stream.wait_stream(torch.cuda.default_stream()))
# the same with a stream made by torch's own runtime handle through ctypes on the mapped libamdhip64
libs = [ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln]
hip = ctypes.CDLL(libs[0])
st = ctypes.c_void_p()
arr = (ctypes.c_uint32 * 8)(*masks[1])
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
print('direct hipExtStreamCreateWithCUMask rc', rc, hex(st.value or 0))
ext = torch.cuda.ExternalStream(st.value, device=dev)
try:
    with torch.cuda.stream(ext):
        a.zero_()
        a.add_(1.0)
    ext.synchronize()
    print('ok   direct stream zero_/add_', float(a[0]))
except Exception as exc:
    print('FAIL direct stream', str(exc).splitlines()[0])
