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

// Producer-side tracking of max|y| of a tensor (the f16x2 kernels scale their input by a power of two derived from it).
// To keep thousands of waves off one address the maximum lives in AMAX_SLOTS slots, AMAX_STRIDE floats apart (one
// 64-byte line each); a consumer takes the maximum over the slots.  Values are non-negative, so the unsigned image
// of the float orders like the float.  The owner zeroes the slots before the producers run.
static constexpr int AMAX_SLOTS = 64, AMAX_STRIDE = 16;
static __device__ __forceinline__ void amax_track(float mx, float *amax, int slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.0f)
        atomicMax(reinterpret_cast<unsigned *>(amax) + (slot & (AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(mx));
}

