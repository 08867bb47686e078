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
int vfn_internal_weight_grad_frag_batch_dev(int32_t shape, int32_t dy_form, int32_t x_form, int32_t n, const void* const* dy, const void* const* x,
                                            float* const* dw_part, float* const* db_part, int64_t n_points, const int32_t* n_dev, int32_t groups,
                                            void* stream);

// launches over a number of points that only the device knows (csrc/vfn_train.hip: the samples with non-zero weight): `n_points` is the
// capacity the launch is sized for, `n_dev` (device memory, may be NULL) the live count; points past it are not touched
int vfn_internal_fused16_fwd_train_at(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                      const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                                      const int32_t* n_dev, int32_t samples_per_ray, float* normals, float* colors, float* saved,
                                      float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks, int32_t save_f16, int64_t ws_first,
                                      int64_t ws_points, int32_t colour_products, void* stream);
int vfn_internal_bwd_chain_bf16_ws_at(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                      const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                      const float* feats, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                                      const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                                      int32_t vec_stride, int64_t n_points, const int32_t* n_dev, float* dz_rgb, float* dz_vec,
                                      int64_t ws_first, int64_t ws_points, void* stream);
int vfn_internal_net_weight_grads_frag_part(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                                            const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                                            const float* aux, const float* dz_head, int64_t n_points, const int32_t* n_dev, uint32_t parts,
                                            int32_t accumulate, void* scratch, void* stream, int32_t stages = 3);

// csrc/vfn_rays.hip: the samples with non-zero weight, compacted on the device (the sparse colour branch of vfn_train_step / vfn_render_fwd)
// (sigma / z_vals NULL: w > 0; given: also the samples whose weight is zero by an underflowed alpha alone — what a training step needs)
int vfn_internal_select_positive(const float* weights, const float* sigma, const float* z_vals, int n_rays, int n_samples, const float* points,
                                 const float* ray_dirs, int32_t* cnt, int32_t* off, int32_t* k_dev, int32_t* sel_sorted, float* pts_sel,
                                 float* dirs_sel, void* stream);
int vfn_internal_rows3_by_index(const float* a, const int32_t* index, const int32_t* k_dev, int64_t capacity, float* out, int gather, void* stream);
// csrc/vfn_mlp16.hip: the gradient-free fused launch over min(n_points, *n_dev) points (outputs scattered through out_index when given)
int vfn_internal_fused16_products_dev(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom, const void* rn_packed16,
                                      const float* points, const float* ray_dirs, int64_t n_points, const int32_t* n_dev, int32_t samples_per_ray,
                                      const int32_t* out_index, int32_t colour_products, float* normals, float* colors, void* stream);
// csrc/vfn_mlp16.hip: the calling thread's range-report word (vfn_f16x3_set_status), or NULL
uint32_t* vfn_internal_f16x3_status();
// The report targets of ONE composite call (vfn_render_params.status_word / clock_stamps): in force for the f16x3 launches the call
// issues, the thread's previous targets back in place when the scope ends.  NULL members leave the thread's targets as they are.
struct VfnReportScope {
    VfnReportScope(uint32_t* status_word, uint64_t* clock_stamps, int64_t clock_slots);
    ~VfnReportScope();
    VfnReportScope(const VfnReportScope&) = delete;
    VfnReportScope& operator=(const VfnReportScope&) = delete;
    uint32_t* old_status; unsigned long long* old_clock; long long old_slots;
};
