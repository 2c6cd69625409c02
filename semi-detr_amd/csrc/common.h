// Shared host-side helpers for libsemidetr_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "semidetr_hip.h"

namespace semidetr {

char *error_buffer();            // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Turns the status of the launch just issued into the C-ABI return code.
inline int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return SEMIDETR_OK;
}

#define SEMIDETR_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return ::semidetr::fail((code), __VA_ARGS__); } while (0)

}  // namespace semidetr
