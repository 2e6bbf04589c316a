// capi_util.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

namespace rl {

char* error_buffer();              // thread-local, defined in capi.hip
constexpr int ERROR_BUFFER_LEN = 512;

inline int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), ERROR_BUFFER_LEN, fmt, ap);
    va_end(ap);
    return code;
}

// Launch errors (bad configuration, missing code object) surface here without a
// device synchronisation; asynchronous faults surface at the caller's next sync.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(-3, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace rl
