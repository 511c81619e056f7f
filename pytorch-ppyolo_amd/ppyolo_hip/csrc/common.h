// Shared helpers for the gfx950 kernels of libppyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/ppyolo_hip.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define PPY_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return PPY_ERR_BAD_ARG; \
    } while (0)

// hipGetLastError() is sticky per thread: an error left behind by the CALLER's own runtime calls (PyTorch probes
// host pointers with hipPointerGetAttributes, which fails benignly) would otherwise be reported as ours.
static inline void ppy_drop_stale_error() { (void)hipGetLastError(); }

static inline int ppy_launch_status() {
    return hipGetLastError() == hipSuccess ? PPY_OK : PPY_ERR_LAUNCH;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float ppy_apply_act(float v, int act) {
    if (act == PPY_ACT_RELU) return v > 0.f ? v : 0.f;      // torch relu: max(v, 0)
    if (act == PPY_ACT_LEAKY) return v > 0.f ? v : v * 0.1f;  // torch leaky_relu(0.1)
    return v;
}
