// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/glfusion.h"

namespace glf {

char* err_buf();                       // thread-local message buffer (glf_api.hip)
int fail(int code, const char* fmt, ...);
int num_cus();                         // cached CU count of the current device
int ensure_init();

inline hipStream_t S(glf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GLF_OK;
}

#define GLF_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return ::glf::fail((code), __VA_ARGS__); } while (0)

}  // namespace glf
