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
