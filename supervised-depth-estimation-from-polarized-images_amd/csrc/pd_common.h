// Shared host-side helpers for libpolardepth.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/polardepth.h"

namespace pd {
char* err_buf();
int fail(int code, const char* fmt, ...);
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PD_OK;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// PD_CONV_* flags of the convolution entry points: known bits only, and not both arithmetic requests at once
inline bool conv_flags_ok(unsigned f) {
    return (f & ~PD_CONV_FLAGS_ALL) == 0 && (f & (PD_CONV_FP32_MFMA | PD_CONV_BF16X3)) != (PD_CONV_FP32_MFMA | PD_CONV_BF16X3);
}
}  // namespace pd

#define PD_REQUIRE(cond, ...) \
    do { if (!(cond)) return pd::fail(PD_EINVAL, __VA_ARGS__); } while (0)
