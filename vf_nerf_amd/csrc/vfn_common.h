// vfn_common.h — error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vfn.h"

void vfn_set_error(const char* fmt, ...);

#define VFN_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            vfn_set_error(__VA_ARGS__);        \
            return VFN_ERR_INVALID;            \
        }                                      \
    } while (0)

static inline int vfn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vfn_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return VFN_ERR_LAUNCH;
    }
    return VFN_OK;
}
