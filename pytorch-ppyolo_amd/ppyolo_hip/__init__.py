"""ppyolo_hip -- MI355X (gfx950) runtime of the PP-YOLO inference hot path.

csrc/      hand-written HIP kernels + the C ABI (include/ppyolo_hip.h)
_lib.py    ctypes binding of libppyolo_hip.so (no fallback: a missing library raises)
ops.py     tensor-level wrappers (torch tensors are storage only)
engine.py  plan builder + executor (direct launches or one hipGraph)
runtime.py module tree -> plan, cached per input shape
dist.py    one-process-per-GPU batch sharding + RCCL all-gather of detections
synth.py   deterministic synthetic weights / inputs for tests and benchmarks
"""
__all__ = ['engine', 'ops', 'runtime', 'synth']
