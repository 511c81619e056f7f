// Shared helpers for the gfx950 kernels of libppyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../../include/ppyolo_hip.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define PPY_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return PPY_ERR_BAD_ARG; \
    } while (0)

// hipGetLastError() is sticky per thread: an error left behind by the CALLER's own runtime calls (PyTorch probes
// host pointers with hipPointerGetAttributes, which fails benignly) would otherwise be reported as ours.
static inline void ppy_drop_stale_error() { (void)hipGetLastError(); }

extern "C" void ppy_note_hip_error(int hip_error);      // capi.hip: kept per thread for ppy_last_hip_error()
static inline int ppy_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PPY_OK;
    ppy_note_hip_error((int)e);
    return PPY_ERR_LAUNCH;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel PER DEVICE: remember, per kernel, on which device
// ordinals it has been raised (one bit each; set atomically -- launches come from several lane threads).
static constexpr int PPY_LDS_MAX = 160 * 1024;      // gfx950: 160 KB of LDS per CU, all of it available to one workgroup
struct PpyLdsAttr {
    std::atomic<unsigned long long> done{0};
};
static inline int ppy_lds_attr(PpyLdsAttr &st, const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PPY_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (bytes > PPY_LDS_MAX) return PPY_ERR_LAUNCH;
    if (st.done.load(std::memory_order_acquire) & bit) return PPY_OK;
    // once per kernel and device, so the limit must not be the FIRST launch's size (a loss kernel that meets a 255- then a
    // 258-channel head, an SPP backward that meets 10x10 then 19x19 maps): always the CU's whole LDS
    // -- minus the kernel's static __shared__ (the sum is what the runtime checks)
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return PPY_ERR_LAUNCH;
    const int room = PPY_LDS_MAX - (int)fa.sharedSizeBytes;
    if (bytes > room) return PPY_ERR_LAUNCH;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess) return PPY_ERR_LAUNCH;
    st.done.fetch_or(bit, std::memory_order_release);
    return PPY_OK;
}

__device__ __forceinline__ float ppy_apply_act(float v, int act) {
    if (act == PPY_ACT_RELU) return v > 0.f ? v : 0.f;      // torch relu: max(v, 0)
    if (act == PPY_ACT_LEAKY) return v > 0.f ? v : v * 0.1f;  // torch leaky_relu(0.1)
    return v;
}

// Producer-side tracking of max|y| PER IMAGE of a tensor (the f16x2 kernels scale the rows of an image by a power of
// two derived from it, so a result never depends on the other images of the batch).  To keep thousands of waves off
// one address every image owns AMAX_SLOTS slots, AMAX_STRIDE floats apart (one 64-byte line each); a consumer takes
// the maximum over the slots of the image.  Values are non-negative, so the unsigned image of the float orders like
// the float.  The owner zeroes all slots before the producers run.  Layout: amax[(n * AMAX_SLOTS + slot) * AMAX_STRIDE].
static constexpr int AMAX_SLOTS = 8, AMAX_STRIDE = 16;
static __device__ __forceinline__ void amax_store(float mx, float *amax, int n, int slot) {
    if (mx > 0.0f)
        atomicMax(reinterpret_cast<unsigned *>(amax) + ((long long)n * AMAX_SLOTS + (slot & (AMAX_SLOTS - 1))) * AMAX_STRIDE,
                  __float_as_uint(mx));
}
// All 64 lanes must call.  One atomic per wave when the wave's values belong to one image, else one per lane.
static __device__ __forceinline__ void amax_track(float mx, int n, float *amax, int slot) {
    const int n0 = __builtin_amdgcn_readfirstlane(n);
    if (__ballot(n != n0) == 0ull) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) amax_store(mx, amax, n0, slot);
    } else {
        amax_store(mx, amax, n, slot + (int)(threadIdx.x & 63));
    }
}
// A wave whose rows span images n_lo..n_hi (wave-uniform): `lo` = max over the rows of image n_lo, `hi` = max over all
// later rows.  Two images (the case for feature maps of >= 64 pixels): exact per image; more (tiny maps): `hi` is
// merged into every later image, an upper bound.
static __device__ __forceinline__ void amax_track2(float lo, float hi, int n_lo, int n_hi, float *amax, int slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmaxf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        amax_store(lo, amax, n_lo, slot);
        for (int n = n_lo + 1; n <= n_hi; ++n) amax_store(hi, amax, n, slot);
    }
}
static __device__ __forceinline__ float amax_read(const float *amax, int n) {     // max over the slots of image n
    float mx = 0.0f;
#pragma unroll
    for (int s = 0; s < AMAX_SLOTS; ++s) mx = fmaxf(mx, fabsf(amax[((long long)n * AMAX_SLOTS + s) * AMAX_STRIDE]));
    return mx;
}
