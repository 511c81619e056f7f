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
