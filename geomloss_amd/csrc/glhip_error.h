// glhip_error.h — the thread-local error message behind glhip_last_error(), shared by the translation units of the library.
#pragma once

#include <cstdarg>
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/glhip.h"

namespace glhip {

extern thread_local char g_err[512];   // defined in glhip_api.hip

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GLHIP_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return GLHIP_OK;
}

}  // namespace glhip
