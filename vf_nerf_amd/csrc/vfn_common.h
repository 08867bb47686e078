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

// launches shared between translation units, not part of the ABI (csrc/vfn_rays.hip, used by csrc/vfn_render.hip)
int vfn_internal_raygen(const vfn_raygen_params* p, const float* uv, const float* pose, const float* intrinsics, const float* k_sign,
                        const float* t_vals, const float* far_per_ray, const float* u_coarse, int gen_u, long long u_base, uint64_t seed,
                        uint64_t offset, float* directions, float* ray_dirs, float* cam_loc, float* z_vals, float* points, void* stream);
int vfn_internal_density_fine(const vfn_density_params* dp, const float* normals_c, const float* ray_dirs, const float* z_c,
                              const float* density_scalars, const vfn_fine_params* fp, const float* directions, const float* cam_loc,
                              const float* far_per_ray, const float* u_fine, const float* u_add, int gen_fine, int gen_add,
                              long long fine_base, long long add_base, uint64_t seed, uint64_t offset, float* z_vals, float* points,
                              int32_t* src, float* new_points, int32_t* dst, int64_t new_row0, void* stream);
int vfn_internal_composite_gather(const vfn_density_params* dp, float* normals, const float* ray_dirs, const float* z_vals,
                                  const float* density_scalars, float* colors, const int32_t* src, const float* normals_c,
                                  const float* colors_c, int64_t n_stored_c, float* weights, float* rgb, float* depth, void* stream);

// csrc/vfn_dwf.hip, used by csrc/vfn_wgrad.hip: n (<= 8) weight-gradient products of one shape and operand form as ONE launch of
// n x groups workgroups (vfn_weight_grad_frag is the n = 1 case)
int vfn_internal_weight_grad_frag_batch(int32_t shape, int32_t dy_form, int32_t x_form, int32_t n, const void* const* dy, const void* const* x,
                                        float* const* dw_part, float* const* db_part, int64_t n_points, int32_t groups, void* stream);
