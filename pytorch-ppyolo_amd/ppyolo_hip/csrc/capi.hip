// Version / error-string entry points of libppyolo_hip.so.
#include "common.h"

extern "C" int ppy_version(void) { return 100; }

extern "C" const char *ppy_error_string(int code) {
    switch (code) {
        case PPY_OK: return "ok";
        case PPY_ERR_BAD_ARG: return "bad argument (shape, alignment or NULL pointer)";
        case PPY_ERR_UNSUPPORTED: return "unsupported configuration";
        case PPY_ERR_WORKSPACE: return "workspace missing or too small";
        case PPY_ERR_LAUNCH: return "HIP launch failed";
    }
    return "unknown error";
}

// The HIP error behind the last PPY_ERR_LAUNCH of this thread ("hipSuccess" if there was none): diagnostics only.
static thread_local int g_last_hip_error = 0;
extern "C" void ppy_note_hip_error(int hip_error) { g_last_hip_error = hip_error; }
extern "C" const char *ppy_last_hip_error(void) { return hipGetErrorName((hipError_t)g_last_hip_error); }

// Lane streams (ppyolo_hip/runtime.py InFlight).  The two batches kept in flight share the chip; with a CU mask per
// lane each batch owns a fixed part of it (hipExtStreamCreateWithCUMask: bit i of the mask enables one CU; on MI355X consecutive
// bits walk over the eight XCDs first -- tools/probes/cu_mask_probe.hip), so one lane's store phase always runs beside the other
// lane's MFMA phase instead of waiting for a chip-wide launch to drain.  The only entry points of the library that create state.
extern "C" int ppy_lane_stream_create(void **stream, const uint32_t *h_cu_mask, int mask_words) {
    if (stream == nullptr || mask_words < 0 || (mask_words > 0 && h_cu_mask == nullptr)) return PPY_ERR_BAD_ARG;
    hipStream_t st = nullptr;
    hipError_t e;
    if (mask_words == 0) {
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    } else {
        bool any = false;
        for (int i = 0; i < mask_words; ++i) any = any || h_cu_mask[i] != 0;
        if (!any) return PPY_ERR_BAD_ARG;          // a stream without a CU would never finish a kernel
        e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask_words, h_cu_mask);
    }
    if (e != hipSuccess) {
        ppy_note_hip_error((int)e);
        return PPY_ERR_LAUNCH;
    }
    *stream = (void *)st;
    return PPY_OK;
}

extern "C" int ppy_lane_stream_destroy(void *stream) {
    if (stream == nullptr) return PPY_ERR_BAD_ARG;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        ppy_note_hip_error((int)e);
        return PPY_ERR_LAUNCH;
    }
    return PPY_OK;
}

// Debug hook (not part of the ABI in include/ppyolo_hip.h, like ppy_debug_set_trace): an empty kernel whose GRID SIZE carries an
// id -- (id + 1) workgroups of 64 lanes -- so that a `rocprofv3 --kernel-trace` of an eager pass can be cut into the plan's ops by
// the markers launched in front of them (bench.py --trace-child / kernel_trace_leg; tools/trace_layers.py).
namespace {
__global__ void __launch_bounds__(64) ppy_marker_kernel() {}
}  // namespace
extern "C" int ppy_debug_marker(int id, void *stream) {
    if (id < 0 || id >= (1 << 20)) return PPY_ERR_BAD_ARG;
    hipLaunchKernelGGL(ppy_marker_kernel, dim3(id + 1), dim3(64), 0, (hipStream_t)stream);
    return PPY_OK;
}
