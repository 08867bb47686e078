/*
 * vfn.h — C ABI of the MI355X-native VF-NeRF volume-rendering hot path.
 *
 * This is the drop-in boundary: every entry point is stateless and stream-ordered, takes plain
 * DEVICE pointers + sizes + a POD parameter struct + the hipStream_t to launch on (passed as
 * void*), allocates nothing, and returns 0 on success or a negative vfn_status (message via
 * vfn_last_error()).  All tensors are dense row-major fp32 unless stated.  The reference is pure
 * Python, so there is no pre-existing FFI; each entry names the reference function it replaces
 * (paths relative to the reference root) and INTEGRATION.md shows the ctypes binding the facade uses.
 *
 * Notation: N rays, S_c coarse samples, N_f fine samples, S_t = S_c + N_f, M points.
 */
#ifndef VFN_H
#define VFN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a POD struct's layout or an entry point's signature changes (2: vfn_render_params.timing_events,
 * vfn_abi_struct_bytes; 3: vfn_f16x3_set_clock_probe, vfn_train_step, vfn_linear_rows_dx_sums; 4: the session form of vfn_train_step —
 * VFN_TRAIN_RENDER / VFN_TRAIN_BACKWARD, vfn_train_step_workspace_layout, vfn_train_step_supervision_points / _forward / _backward; 5: vfn_select_samples,
 * the training selection of vfn_train_step's sparse colour branch, vfn_grid_lattice_points, vfn_linear_rows_fold, vfn_weight_grad_partials_bf16_fold).  The Python binding reads this constant from this file and refuses a library
 * that reports another. */
#define VFN_ABI_VERSION 5

typedef enum vfn_status {
    VFN_OK = 0,
    VFN_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    VFN_ERR_LAUNCH = -2,      /* HIP launch error */
    VFN_ERR_UNSUPPORTED = -3  /* network geometry the kernels are not specialised for */
} vfn_status;

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* vfn_last_error(void);
int vfn_abi_version(void);
/* sizeof() of the POD structs of this header as the library was compiled, by index: 0 vfn_net_geom, 1 vfn_layer_params,
 * 2 vfn_raygen_params, 3 vfn_density_params, 4 vfn_fine_params, 5 vfn_render_params, 6 vfn_unfold_entry, 7 vfn_wgrad_layer,
 * 8 vfn_loss_params, 9 vfn_train_step_params, 10 vfn_train_step_io;
 * -1 for any other index.  A binding checks its own mirrors of the structs against these (tests/test_host_logic.py). */
int32_t vfn_abi_struct_bytes(int32_t which);

/* ---------------------------------------------------------------------------------------------
 * Network geometry + packed weights.
 *
 * The kernels are specialised for the shipped architecture family (confs/vf_nerf.conf:13-37):
 * hidden width 256 everywhere, ReLU between layers, eval-mode BatchNorm folded into each Linear,
 * one optional skip layer that concatenates the positional encoding (vector_field_network.py:192-193),
 * VF head = 3 vector columns + F feature columns with tanh, rendering head = 3 columns with sigmoid.
 * ------------------------------------------------------------------------------------------- */
#define VFN_MAX_LAYERS 16
#define VFN_HIDDEN 256

typedef struct vfn_net_geom {
    int32_t n_layers;        /* number of Linear layers (VF: 9, render: 5 in the shipped conf)          */
    int32_t multires;        /* positional-encoding octaves L (embedder.py:40-52); VF: 6, render: 4     */
    int32_t skip_layer;      /* index of the layer whose input is cat([x, PE])/sqrt(2); -1 = none       */
    int32_t feature_dims;    /* VF: F (256 or 0); render: width of the feature input (256 or 0)         */
    int32_t in_dims[VFN_MAX_LAYERS];   /* reference in_features of each Linear                          */
    int32_t out_dims[VFN_MAX_LAYERS];  /* reference out_features of each Linear                         */
    int32_t has_bn[VFN_MAX_LAYERS];    /* 1 when the Linear is followed by BatchNorm1d                  */
} vfn_net_geom;

/* Raw (reference-layout) parameter pointers of one layer: Linear weight[out][in], bias[out], and the
 * BatchNorm1d weight/bias/running_mean/running_var[out] (NULL when has_bn == 0). */
typedef struct vfn_layer_params {
    const float* weight;
    const float* bias;
    const float* bn_weight;
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
} vfn_layer_params;

#define VFN_NET_VF 0
#define VFN_NET_RENDER 1

/* Number of floats of the packed-weight workspace for a network of this geometry (host-side, no GPU).
 * Returns < 0 (vfn_status) when the geometry is unsupported. */
int64_t vfn_packed_size(int32_t net_kind, const vfn_net_geom* geom);

/* Fold eval-mode BatchNorm (eps 1e-5) and the skip 1/sqrt(2) into each Linear and re-order the result
 * into the MFMA-fragment order the fused kernels stream (see DESIGN.md "packed weights").  Must be
 * re-run after every optimizer step; reads the live parameter storage, writes `packed`.
 * Replaces: the per-call nn.Linear/nn.BatchNorm1d parameter reads of vector_field_network.py:177-208
 * and rendering_network.py:98-103. */
int vfn_pack_weights(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                     float* packed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1 — rays + coarse sampler.
 * Replaces utils/rendering.py:12-60 (+ utils/pinhole_model.py:9-63) and
 * models/samplers/ray_sampler.py:49-80,113-142.
 * ------------------------------------------------------------------------------------------- */
typedef struct vfn_raygen_params {
    int32_t n_rays;
    int32_t n_samples;       /* S_c */
    int32_t pose_is_quat;    /* 0: pose[N,4,4]; 1: pose[N,7] = (qr,qi,qj,qk,tx,ty,tz) */
    float near;
    float far;               /* used when far_per_ray == NULL */
} vfn_raygen_params;

/* uv[N,2] (u = x, v = y), pose, intrinsics[N,4,4], t_vals[S_c] = linspace(0,1,S_c) (host supplied so
 * that it is bit-identical to torch.linspace), far_per_ray[N] or NULL, u_coarse[N,S_c] uniforms in
 * [0,1) or NULL for deterministic sampling.
 * Outputs: directions[N,3] (un-normalised, Q7), ray_dirs[N,3] (unit), cam_loc[N,3], z_vals[N,S_c],
 * points[N,S_c,3]. */
int vfn_raygen_uniform(const vfn_raygen_params* p, const float* uv, const float* pose, const float* intrinsics,
                       const float* t_vals, const float* far_per_ray, const float* u_coarse,
                       float* directions, float* ray_dirs, float* cam_loc, float* z_vals, float* points,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2/K4 — fused MLPs (fp32 MFMA, weights from vfn_pack_weights).
 * ------------------------------------------------------------------------------------------- */
/* points[M,3] -> out.  out_cols == 3: only the vector columns [M,3] (proposal pass, grid queries:
 * vector_field_nerf.py:253-256, evaluation/utils/mc_utils.py:100); out_cols == 3+F: the full
 * [M,3+F] row (vector_field_network.py:140-208, eval mode). */
int vfn_vf_mlp_fwd(const vfn_net_geom* geom, const float* packed, const float* points, int64_t n_points,
                   int32_t out_cols, float* out, void* stream);

/* Rendering MLP alone: rendering_network.py:62-108 (mode "idr").  view_dirs[M,3] per point. */
int vfn_render_mlp_fwd(const vfn_net_geom* geom, const float* packed, const float* points, const float* normals,
                       const float* view_dirs, const float* feats, int64_t n_points, float* colors,
                       void* stream);

/* Fine pass in one launch: VF MLP -> (features stay in LDS) -> rendering MLP.
 * points[M,3] with M = N*S_t, ray_dirs[N,3]; row m uses ray_dirs[m / samples_per_ray].
 * Outputs normals[M,3] (tanh'ed vector columns) and colors[M,3] (sigmoid); feats_out[M,F] optional
 * (NULL to skip).  Replaces vector_field_nerf.py:294-297 + :315-318. */
int vfn_vf_render_fused_fwd(const vfn_net_geom* vf_geom, const float* vf_packed,
                            const vfn_net_geom* rn_geom, const float* rn_packed,
                            const float* points, const float* ray_dirs, int64_t n_points,
                            int32_t samples_per_ray, float* normals, float* colors, float* feats_out,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3 — per-ray density -> VolSDF weights (-> argmax / composite).
 * Replaces vector_field_nerf.py:442-474 (get_density), models/helpers/functions.py:41-72,
 * models/helpers/density_functions.py:129-204, utils/rendering.py:122-148 and the sums at
 * vector_field_nerf.py:322-323.
 * ------------------------------------------------------------------------------------------- */
typedef struct vfn_density_params {
    int32_t n_rays;
    int32_t n_samples;        /* S: samples per ray in this pass                                   */
    int32_t n_window;         /* W = len(cos_sim_weights); uniform ones/W weights are used (Q6)     */
    int32_t normalize;        /* normalize_rendering                                                */
    float dir_to_normal_th;
    float beta_min, beta_max; /* density_config.beta_bounds                                         */
    float mean_min, mean_max; /* density_config.mean_bounds                                         */
    float scale_min;
    float cutoff;             /* -0.5: the reference drops the configured cutoff (Q5)               */
} vfn_density_params;

/* normals[N,S,3], ray_dirs[N,3], z_vals[N,S]; density_scalars = device pointer to {beta, mean, scale}
 * (raw, un-clamped learnable values).  Outputs (any may be NULL): sigma[N,S], weights[N,S],
 * argmax[N] int64 (first maximum of weights), and, when colors[N,S,3] is given, rgb[N,3] and
 * depth[N] (= sum_s w c, sum_s w z). */
int vfn_ray_density_weights(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                            const float* z_vals, const float* density_scalars, const float* colors,
                            float* sigma, float* weights, int64_t* argmax, float* rgb, float* depth,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3b — range fine sampler.  Replaces models/samplers/ray_sampler.py:264-302 + :77-78.
 * ------------------------------------------------------------------------------------------- */
typedef struct vfn_fine_params {
    int32_t n_rays;
    int32_t n_coarse;   /* S_c */
    int32_t n_fine;     /* N_f = min(N_samples, max_samples) */
    float near;
    float far;          /* used when far_per_ray == NULL */
    float half_range;   /* fine_range, as fp32 */
    float window_step;  /* fp32(2*fine_range/(N_f-1)) evaluated in double by the host, as Python does */
    float span;         /* fp32(far - near) evaluated in double by the host; ignored with far_per_ray */
} vfn_fine_params;

/* z_coarse[N,S_c], argmax[N] (from vfn_ray_density_weights), directions[N,3], cam_loc[N,3],
 * u_fine[N,N_f] or NULL (deterministic window), u_add[N,N_f] (always consumed, Q9).
 * Outputs z_vals[N,S_t] ascending, points[N,S_t,3]. */
int vfn_range_fine_sample(const vfn_fine_params* p, const float* z_coarse, const int64_t* argmax,
                          const float* directions, const float* cam_loc, const float* far_per_ray,
                          const float* u_fine, const float* u_add, float* z_vals, float* points, void* stream);

/* The same sampler, additionally reporting the provenance of every sorted sample, so that a caller that already holds
 * per-sample results for the proposal samples (the reference evaluates the vector-field net on them twice:
 * vector_field_nerf.py:252-277 and :294-297) can evaluate only the N_f new ones and gather:
 * src[N,S_t] = ray*S_c + j for proposal sample j of the ray, new_row0 + ray*N_f + k for new sample k (new_row0 >= N*S_c:
 * the row at which the caller stores the new samples' results); new_points[N,N_f,3] are the new samples in generation
 * order; dst[new_row0 + N*N_f] is the inverse map, dst[src[i]] = i (rows no sample comes from are left untouched).
 * Each of the three may be NULL. */
int vfn_range_fine_sample_indexed(const vfn_fine_params* p, const float* z_coarse, const int64_t* argmax,
                                  const float* directions, const float* cam_loc, const float* far_per_ray,
                                  const float* u_fine, const float* u_add, float* z_vals, float* points, int32_t* src,
                                  float* new_points, int32_t* dst, int64_t new_row0, void* stream);

/* The samplers as stand-alone calls, for callers that sample through the sampler OBJECTS rather than through render():
 * UniformSampler.get_z_vals / RaySampler.sample (models/samplers/ray_sampler.py:113-142, :49-80) on given
 * directions[N,3] (un-normalised, Q7) / cam_loc[N,3]: z = near (1 - t) + far t, stratified with u[N,S] when given
 * (NULL = deterministic), points = cam_loc + z * directions (points may be NULL: depths only).  Same arithmetic, in the
 * same order, as vfn_raygen_uniform: bit-identical depths. */
int vfn_uniform_sample(int32_t n_rays, int32_t n_samples, float near, float far, const float* directions,
                       const float* cam_loc, const float* t_vals, const float* far_per_ray, const float* u,
                       float* z_vals, float* points, void* stream);

/* out[r] = index of the FIRST maximum of row r of w[n_rows, n_cols] (torch.argmax(coarse_weights, dim=-1),
 * ray_sampler.py:277; an all-zero row gives 0, Q9), int64. */
int vfn_rows_argmax(const float* w, int32_t n_rows, int32_t n_cols, int64_t* out, void* stream);

/* The sample selection of the sparse colour branch (vfn_render_fwd with sparse_colours, vfn_train_step with sparse_colours) on its own:
 * weights[N,S], points[N,S,3], ray_dirs[N,3] -> count[0] = K (device memory), and for k < K in ray order: index[k] = ray * S + j,
 * points_sel[k], dirs_sel[k] (each sized for N * S rows).  sigma == z_vals == NULL: the samples with w > 0 — all a forward render's
 * rgb = sum w c needs (models/nerf/vector_field_nerf.py:322).  With sigma[N,S] and z_vals[N,S]: additionally the samples whose weight is
 * zero by an underflowed alpha alone (sigma > 0, delta > 0, T > 0: d w / d sigma = T delta exp(-sigma delta) is not zero there, and the
 * backward of utils/rendering.py:122-148 multiplies it with the sample's colour) — the selection a training step uses.
 * scratch: 2 N int32. */
int vfn_select_samples(const float* weights, const float* sigma, const float* z_vals, int32_t n_rays, int32_t n_samples, const float* points,
                       const float* ray_dirs, int32_t* scratch, int32_t* count, int32_t* index, float* points_sel, float* dirs_sel, void* stream);

/* RaySampler.sample with additional_depths (ray_sampler.py:69-73): z_out[N, S + E] = sort(cat(z_vals[N,S], extra[N,E])) per ray
 * (ascending, NaN last, like torch.sort), points[N, S + E, 3] = cam_loc + z_out * directions (NULL: depths only).  S + E <= 2048. */
int vfn_merge_sort_depths(const float* z_vals, const float* extra, int32_t n_rays, int32_t n_samples, int32_t n_extra,
                          const float* directions, const float* cam_loc, float* z_out, float* points, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The whole gradient-free render() in ONE call: VectorFieldNerf.render (models/nerf/vector_field_nerf.py:216-338) on the f16x3
 * kernels with one vector-field evaluation per distinct sample, issued from C on one stream out of one caller-supplied workspace
 * (vfn_render_fwd_workspace_bytes; no allocation, no synchronisation) as FIVE launches: rays + proposal samples (missing draws
 * generated in place) | fused VF + rendering net on the proposal samples | proposal weights -> argmax -> range fine sampler |
 * the fused launch on the new samples, outputs scattered | weights + composite (which also moves the proposal samples' normals
 * and colours to their sorted positions).  The per-ray launches run the kernels of vfn_raygen_uniform, vfn_fill_uniform,
 * vfn_ray_density_weights, vfn_range_fine_sample_indexed and vfn_scatter_rows3, so every value equals what those entry points
 * produce called one by one.
 * Random draws: u_coarse[N,S_c] / u_fine[N,N_f] (read only when the matching perturb flag is set) and u_add[N,N_f] (always
 * consumed, Q9) may each be NULL, in which case they come from the Philox stream (seed, offset) in that order; the call
 * consumes ceil(generated / 4) counter values.  far_*_per_ray: optional [N] (ray_sampler.py:126-127).
 * Outputs: ray_dirs[N,3] (unit), z_vals[N,S_t], points[N,S_t,3], normals[N*S_t,3], colors[N*S_t,3], weights[N,S_t], rgb[N,3],
 * depth[N]; N*S_t < 2^22. */
typedef struct vfn_render_params {
    int32_t n_rays, n_coarse, n_fine;   /* N, S_c, N_f = min(fine_sampler.N_samples, max_samples) */
    int32_t pose_is_quat;
    int32_t perturb_coarse, perturb_fine;
    float near_coarse, near_fine;       /* ray_sampler.near, fine_sampler.near (the trainer sets both from the dataset bounds) */
    float far_coarse, far_fine;         /* used when the per-ray pointer is NULL */
    float fine_range, window_step, span; /* as in vfn_fine_params: window_step, span evaluated in double by the host */
    vfn_density_params density;         /* n_rays / n_samples are filled in per pass */
    uint64_t seed, offset;              /* Philox stream for the draws not supplied */
    int32_t colour_products;            /* 0 or 3: three products everywhere; 2: the colour branch on two (vfn_vf_render_fused16_products) */
    int32_t separate_launches;          /* 0: the five merged launches; 1: the same pipeline through the stand-alone entry points (eight) */
    int32_t streams;                    /* 2: the batch in two halves, the second on an internal side stream forked from and joined
                                         * into `stream` (same values; the halves fill each other's partial rounds); 1: one stream;
                                         * 0: two halves when the fused launches would leave >= 5 % of their workgroup slots empty */
    int32_t sparse_colours;             /* != 0 (opt-in): colours are evaluated only for the samples whose weight is non-zero — the vector-field
                                         * net on every sample with its vector-only launch, the fused VF + rendering launch on the compacted list
                                         * of samples with w > 0 (a count the host never learns).  rgb, depth, weights, normals, z_vals, points are
                                         * bit-identical to the dense plan; `colors` holds zeros where w = 0.  For callers that keep rgb / depth
                                         * only (evaluation/methods.py:528-540) */
    void* timing_events[4];             /* optional hipEvent_t handles (NULL: none), recorded on the launch stream around the two fused
                                         * VF + rendering launches: [0] before / [1] after the one on the proposal samples, [2] / [3] the one
                                         * on the new samples (with `streams` > 1: around the FIRST range's launches).  bench.py times the
                                         * dominant kernel with these without leaving the one-call path. */
    uint32_t* status_word;              /* optional (ABI 4): where the f16x3 launches of THIS call report operands outside their range (bit 0:
                                         * a hidden activation reached the f16 clamp, bit 1: an input coordinate did) — per call, no hidden
                                         * state; NULL: wherever vfn_f16x3_set_status (the ABI-3 setter, kept as a shim) pointed this thread */
    uint64_t* clock_stamps;             /* optional (ABI 4): [clock_slots][2] per-workgroup (shader-clock cycles, 100 MHz ticks) of the fused
                                         * launches of THIS call (vfn_f16x3_set_clock_probe is the ABI-3 shim) */
    int64_t clock_slots;
} vfn_render_params;
int64_t vfn_render_fwd_workspace_bytes(const vfn_render_params* p);
int vfn_render_fwd(const vfn_render_params* p, const vfn_net_geom* vf_geom, const void* vf_packed16,
                   const vfn_net_geom* rn_geom, const void* rn_packed16, const float* uv, const float* pose,
                   const float* intrinsics, const float* t_vals, const float* far_coarse_per_ray,
                   const float* far_fine_per_ray, const float* density_scalars, const float* u_coarse,
                   const float* u_fine, const float* u_add, void* workspace, float* ray_dirs, float* z_vals,
                   float* points, float* normals, float* colors, float* weights, float* rgb, float* depth,
                   void* stream);

/* Counter-based uniforms in [0,1) for production sampling (Philox4x32-10, one 4-tuple per 4 outputs). */
int vfn_fill_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* =============================================================================================
 * Training path (autograd of the fine pass: models/nerf/vector_field_nerf.py:294-323 under
 * train/vector_field_nerf_train.py:251-260, eval-mode BatchNorm as in :140-141).
 *
 * Workspace layout shared by the entry points below ("slots"): slot s is a dense [M,256] fp32 matrix;
 * slots 0..H_vf-1 are the VF net's hidden layers in order (the last one is the 256-wide feature block of
 * the final Linear when feature_dims > 0), slots H_vf..H_vf+H_rn-1 the rendering net's hidden layers.
 * `saved` holds post-activation outputs (forward), `dy` pre-activation gradients (backward).
 * ============================================================================================= */

/* Size (floats) and fill of the TRANSPOSED weight pack used by dX = dY * W'. */
int64_t vfn_packed_bwd_size(int32_t net_kind, const vfn_net_geom* geom);
int vfn_pack_weights_bwd(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                         float* packed_bwd, void* stream);

/* Forward kernels that additionally write `saved` slots and the auxiliary input tiles
 * (save_aux_vf[M,40] = positional encoding of the point, save_aux_rn[M,40] = [p, PE(d), n]). */
int vfn_vf_mlp_fwd_train(const vfn_net_geom* geom, const float* packed, const float* points, int64_t n_points,
                         int32_t out_cols, float* out, float* saved, float* save_aux_vf, void* stream);
int vfn_vf_render_fused_fwd_train(const vfn_net_geom* vf_geom, const float* vf_packed,
                                  const vfn_net_geom* rn_geom, const float* rn_packed,
                                  const float* points, const float* ray_dirs, int64_t n_points,
                                  int32_t samples_per_ray, float* normals, float* colors, float* saved,
                                  float* save_aux_vf, float* save_aux_rn, void* stream);

/* dX chain.  rn_geom != NULL: gradients d_colors[M,3] (wrt the sigmoid outputs `colors`) and d_vec (wrt the
 * tanh'ed vector columns `vec`, row stride vec_stride) are pushed back through the rendering net, the feature
 * hand-off and the VF net.  rn_geom == NULL: VF net only; d_feats (wrt the tanh'ed features, row stride
 * vec_stride) may be NULL.  Writes every `dy` slot plus dz_rgb[M,4] / dz_vec[M,4] (pre-activation gradients of
 * the two 3-channel heads, 4th column zero). */
int vfn_mlp_bwd_chain(const vfn_net_geom* vf_geom, const float* vf_packed, const float* vf_packed_bwd,
                      const vfn_net_geom* rn_geom, const float* rn_packed, const float* rn_packed_bwd,
                      const float* saved, float* dy, const float* d_colors, const float* colors,
                      const float* d_vec, const float* vec, const float* d_feats, int32_t vec_stride,
                      int64_t n_points, float* dz_rgb, float* dz_vec, void* stream);

/* Weight-gradient partials dW'[n][k] = sum_m dY[m][n] X[m][k] as `groups` slabs [groups][n_out][ld_out], plus
 * db'[n] = sum_m dY[m][n] as [groups][n_out] (db_part may be NULL).  shape 0: n_out 256, ld_out 256 (hidden
 * layer, act inputs); 1: n_out 256, ld_out 64 (aux inputs); 2: n_out 32, ld_out 256 (3-channel head).
 * Columns >= n_valid of dY / >= k_valid of X read as zero. */
int vfn_weight_grad_partials(int32_t shape, const float* dy, int32_t ld_dy, int32_t n_valid, const float* x,
                             int32_t ld_x, int32_t k_valid, int64_t n_points, int32_t groups, float* dw_part,
                             float* db_part, int32_t x_f16, void* stream);
/* x_f16 != 0 (shape 2 only): the rows of X hold 256 f16 values in their first 512 bytes (the opt-in storage of the f16x3
 * training forwards, see vfn_vf_mlp16_fwd_train); likewise for vfn_weight_grad_partials_bf16. */

/* Backward of vfn_ray_density_weights: upstream d_rgb[N,3], d_depth[N], d_weights[N,S] (each may be NULL) ->
 * d_colors[N,S,3] (written, may be NULL), d_normals[N,S,3] (ACCUMULATED into), d_scalars[3] (atomically
 * accumulated gradients of the raw beta, mean, scale).  n_samples <= 256. */
int vfn_ray_density_weights_bwd(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                                const float* z_vals, const float* density_scalars, const float* colors,
                                const float* d_rgb, const float* d_depth, const float* d_weights,
                                float* d_normals, float* d_colors, float* d_scalars, void* stream);

/* Backward of the density alone — VectorFieldNerf.get_density under autograd (models/nerf/vector_field_nerf.py:442-474 is
 * part of the reference's graph): upstream d_sigma[N,S] -> d_normals[N,S,3] (ACCUMULATED into), d_scalars[3]
 * (atomically accumulated, may be NULL).  z_vals[N,S] only feeds the composite terms, which carry no gradient here. */
int vfn_ray_density_sigma_bwd(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                              const float* z_vals, const float* density_scalars, const float* d_sigma,
                              float* d_normals, float* d_scalars, void* stream);

/* =============================================================================================
 * "f16x3" inference kernels: the same MLPs on the f16 matrix cores with fp32-equivalent accuracy.
 * Every value is carried as two halves (hi + lo, 22 significant bits) and each product is evaluated as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 f16 MFMAs per K=16
 * block instead of 8 fp32 MFMAs.  Weights come from vfn_pack16_weights (BatchNorm folded, scaled by 2^6, split,
 * fragment order; re-run after every optimizer step).  Specialised for the shipped layer shapes
 * (confs/vf_nerf.conf:13-37); other geometries return VFN_ERR_UNSUPPORTED and callers use the fp32 kernels.
 * ============================================================================================= */
int64_t vfn_pack16_size(int32_t net_kind, const vfn_net_geom* geom);           /* bytes */
int vfn_pack16_weights(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                       void* packed16, void* stream);
/* points[M,3] -> the 3 tanh'ed vector columns [M,3] (proposal pass / grid queries). */
int vfn_vf_mlp16_fwd(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                     float* out_vec, void* stream);
/* Fine pass: VF MLP -> rendering MLP, features stay in registers; normals[M,3], colors[M,3]. */
int vfn_vf_render_fused16_fwd(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                              const void* rn_packed16, const float* points, const float* ray_dirs,
                              int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                              void* stream);

/* Range guard of the f16x3 kernels.  The split representation covers |activation| < ~937 (the ReLU epilogue clamps the 2^6-scaled
 * value at 60 000 so that an out-of-family value degrades instead of becoming inf - inf), |input coordinate| likewise, and
 * folded weights whose largest entry per layer is neither in the f16 denormal range nor beyond 65 504.  The reference has no
 * such restriction (vector_field_network.py:177-208), so the kernels REPORT when they leave it:
 *  - vfn_f16x3_set_status(word): every later f16x3 launch of the calling thread ORs into *word (device memory, 4 bytes) bit 0
 *    when a hidden activation hit the clamp and bit 1 when an input did; NULL switches the reporting off.  Thread-local.
 *  - vfn_pack16_weights writes, behind the pack (the last 256 bytes of the vfn_pack16_size buffer), one word per pack entry
 *    (hidden layers in plan order, then the 3-channel head): the bit pattern of max |folded weight| of that entry.
 * The facade reads both and repeats a flagged call on the exact-fp32 kernels (vf_nerf_amd/nerf.py, f16x3_guard). */
int vfn_f16x3_set_status(uint32_t* status_word);
/* Clock probe of the dominant kernel.  The fused VF + rendering launch is power-limited: the shader clock it runs at is a result
 * of the launch, not a constant of the chip (1.9-2.1 GHz measured against a nominal 2.4).  After vfn_f16x3_set_clock_probe(stamps,
 * slots) every gradient-free fused launch of the calling thread (vfn_vf_render_fused16_fwd / _products / _scatter, and through them
 * vfn_render_fwd) has workgroup b < slots store two uint64 at stamps[2b], stamps[2b+1] (device memory): the shader-clock cycles
 * (s_memtime) and the 100 MHz constant-clock ticks (s_memrealtime) between its first and its last instruction.  cycles / ticks x
 * 0.1 = GHz while that workgroup ran.  NULL / 0 switches it off (the default).  Thread-local.  No reference counterpart. */
int vfn_f16x3_set_clock_probe(uint64_t* stamps, int64_t slots);
/* vfn_weight_grad_partials(shape 0, ld 256, all 256 columns valid) on the bf16 matrix cores: operands split into two
 * bf16 halves (16 significant bits, fp32 exponent range), three products per K-block, fp32 accumulation; same outputs
 * (`groups` slabs [groups][256][256] and [groups][256]).  ~2^-16 relative error per product under the sum over points. */
int vfn_weight_grad_partials_bf16(const float* dy, const float* x, int64_t n_points, int32_t groups, float* dw_part,
                                  float* db_part, int32_t x_f16, void* stream);
/* The same over 256 columns of wider matrices: rows ld_dy / ld_x floats apart (>= 256, multiples of 4, 16-byte aligned bases) — a
 * 256-column block of the rendering net's 289-wide input or of the vector-field net's 259-wide output gradient (batchstat.py). */
int vfn_weight_grad_partials_bf16_ld(const float* dy, int32_t ld_dy, const float* x, int32_t ld_x, int64_t n_points, int32_t groups,
                                     float* dw_part, float* db_part, int32_t x_f16, void* stream);
/* The same with the ACTIVATION folded into the X operand's read (round 6, ABI 5): z_prev[M, ldz] is the previous layer's pre-BatchNorm
 * output and the operand is post_prev * max(z * scale + shift, 0) on the first n_prev of the 256 columns (coef_prev = [4][n_prev]: scale |
 * shift | mean | rstd, as vfn_bstat_finalize writes them) and post_prev * z on the others (the skip layer's re-injected encoding,
 * vector_field_network.py:192-193) — vfn_bstat_relu_rows' expression value for value, so the activated matrix of a training-mode
 * layer (vector_field_network.py:146-173 under model.train()) need not exist in HBM. */
int vfn_weight_grad_partials_bf16_fold(const float* dy, int32_t ld_dy, const float* z_prev, int32_t ldz, const float* coef_prev,
                                       int32_t n_prev, float post_prev, int64_t n_points, int32_t groups, float* dw_part,
                                       float* db_part, void* stream);

/* vfn_mlp_bwd_chain on the bf16 matrix cores (split operands, three products per K-block, fp32 accumulation; shipped layer
 * shapes only, others return VFN_ERR_UNSUPPORTED).  Takes its own TRANSPOSED bf16 packs (vfn_pack_weights_bwd16; re-run
 * after every optimizer step) and the raw rows 0..2 of each net's last Linear ([3][256] fp32) for the 3-channel heads.
 * Same outputs as vfn_mlp_bwd_chain.  n_points < 2^22 per launch.
 * `masks` = the sign bits the f16x3 training forwards write next to `saved` (save_masks[13][M][2][4] u32: per slot, point and
 * lane half g one 16-byte word; tile t of the layer -> half t & 1 of dword t >> 1, bit r <-> output column
 * 32 t + (r & 3) + 8 (r >> 2) + 4 g): a ReLU's backward needs only whether its output was positive, so the chain reads
 * 32 bytes per point and layer instead of 1 KiB; `saved` itself is read only for the tanh'ed feature block (slot 8). */
int64_t vfn_packed_bwd16_size(int32_t net_kind, const vfn_net_geom* geom);                      /* bytes */
int vfn_pack_weights_bwd16(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers, void* packed,
                           void* stream);
/* The same pack with round_hi != 0: the hi planes hold the bf16 ROUNDING of the folded weights (to nearest even) instead of
 * their truncation — the pack of the single-product chain (dy_flags bit 4 below), which reads nothing else. */
int vfn_pack_weights_bwd16_mode(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers, int32_t round_hi,
                                void* packed, void* stream);
int vfn_mlp_bwd_chain_bf16(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                           const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                           const float* saved, const uint32_t* masks, float* dy, const float* d_colors, const float* colors,
                           const float* d_vec, const float* vec, const float* d_feats, int32_t vec_stride,
                           int64_t n_points, float* dz_rgb, float* dz_vec, void* stream);

/* The same chain for the FRAGMENT-ORDERED workspace (below): `feats` = the tanh'ed features [M][256] row-major fp32 (the one
 * slot the chain reads as values), `dy` = the gradient slots it writes, dy_flags bit 1: fragment order, bit 2 (with bit 1):
 * as bf16, bit 3 (with bit 1, not with bit 2): as SCALED f16 (vfn_weight_grad_frag, dy_form 3).  dy_flags = 0 and feats =
 * slot 8 of saved[13][M][256] is vfn_mlp_bwd_chain_bf16.  n_points < 2^21 per LAUNCH (32-bit offsets relative to the launch's first point);
 * a workspace filled and walked by several launches (the _at forms) may hold up to 2^26 points.
 * dy_flags bit 4 (with bits 1 and 3; fused or vector-only chains): SINGLE-PRODUCT arithmetic — one bf16 product per K-block on
 * round-to-nearest operands (8 significant bits each) instead of three on split ones (16): the chain of the opt-in 16-bit-native
 * training mode (BASELINE.json configs[2], "bf16 MFMA MLPs" as written; vf_nerf_amd: model.training_products = 1).  Takes packs
 * made by vfn_pack_weights_bwd16_mode(round_hi = 1).  Outside the 1e-3 gradient bound of the default chain by construction. */
int vfn_mlp_bwd_chain_bf16_ws(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                              const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                              const float* feats, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                              const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                              int32_t vec_stride, int64_t n_points, float* dz_rgb, float* dz_vec, void* stream);
/* The same over PART of a workspace sized for ws_points points: the launch's n_points points are points ws_first .. of feats,
 * masks, dy, dz_rgb and dz_vec (workspace-indexed); d_colors / colors / d_vec / vec / d_feats stay launch-local.  Several
 * forwards can then share one workspace (vfn_vf_render_fused16_fwd_train_at, vfn_vf_mlp16_fwd_train_at) and the weight-gradient
 * kernels walk it once. */
int vfn_mlp_bwd_chain_bf16_ws_at(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                 const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                 const float* feats, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                                 const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                                 int32_t vec_stride, int64_t n_points, float* dz_rgb, float* dz_vec, int64_t ws_first,
                                 int64_t ws_points, void* stream);

/* FRAGMENT-ORDERED training workspace.  A slot (one layer's saved activations, or its pre-activation gradients) is stored as
 * the producing waves hold it: groups of 32 points (one wave), 32 KiB per group, inside a group piece (t, q) = registers
 * 4q..4q+3 of output tile t at byte (4 t + q) * 1024 (512 in 16-bit forms), inside a piece lane L = 32 g + i at L * 16 (8)
 * bytes holding columns 32 t + 8 q + 4 g .. + 3 of point 32 G + i.  Every store of the f16x3 training forward / the bf16
 * chain and every load of the weight-gradient kernel is then 1 KiB (512 B) of consecutive bytes; row-major slots made each
 * store touch 32 lines with 32 (16) bytes.  Slot stride = ceil(M / 32) * 32 KiB whatever the element size.
 * vfn_weight_grad_frag: the weight-gradient partial slabs of vfn_weight_grad_partials (same `groups` slabs, same shapes
 * 0: [256][256], 1: [256][64], 2: [32][256], db_part [groups][256 | 256 | 32]) from such slots, split-bf16 products on
 * v_mfma_f32_32x32x16_bf16:
 *   dy_form 0 fragment fp32 | 1 fragment bf16 (no low half: two products per K-block) | 2 dz[M][4] fp32 rows (shape 2 only)
 *           3 fragment SCALED f16: the 16 values lane L holds of tile t are f16(dY * 2^k), k chosen by the producer so that the
 *             largest magnitude lies in [2^14, 2^15) (11 significant bits at any gradient scale); pieces where the bf16 form has
 *             them, the byte k + 113 (255: all zero) at byte 16384 + 64 t + L of the group.  The kernel brings a slab's pieces to
 *             one common scale and multiplies on v_mfma_f32_32x32x16_f16: one product per K-block with f16 activations (x_form 1),
 *             two with fp32 ones (f16 hi + lo)
 *   x_form  0 fragment fp32 | 1 fragment f16 | 2 [M][256] fp32 rows (the features) | 3 the encoding tile aux[M][40] (shape 1 only)
 * Points past M in the last group count as zero. */
int vfn_weight_grad_frag(int32_t shape, const void* dy, int32_t dy_form, const void* x, int32_t x_form, int64_t n_points,
                         int32_t groups, float* dw_part, float* db_part, void* stream);

/* Split inference launches.  The reference evaluates the vector-field net on every proposal sample twice (no-grad pass
 * vector_field_nerf.py:252-277, then again among the S_c+N_f samples at :294-297); the two entry points below let a caller
 * evaluate it once per distinct sample and run the rendering net on gathered results — same arithmetic per sample as
 * vfn_vf_render_fused16_fwd, so the outputs are identical.
 * vfn_vf_feat16_fwd: points[M,3] -> out_vec[M,3] (tanh'ed vector columns) and out_blocks, the 256 tanh'ed features in the
 *   rendering kernel's operand format (split f16): 32 KiB per group of 32 consecutive rows,
 *   [operand block 0..15][hi | lo][lane 0..63][16 B], i.e. the register image of the wave that owns those rows — every store
 *   and every later load moves 1 KiB of consecutive bytes.  The buffer holds ceil(M / 32) groups (rows past M in the last
 *   group are written with don't-care values); a launch must start on a group boundary of the buffer.  M < 2^22.
 * vfn_render16_from_blocks: the rendering net over the n_rows stored rows, in storage order: row r has its features in
 *   blocks, its vector columns in vecs[r], and dst[r] = its position among the sorted samples of the batch
 *   (vfn_range_fine_sample_indexed produces dst; dst[r] < 0 marks a padding row): position points[dst[r]], view direction
 *   ray_dirs[dst[r] / S]; writes normals[dst[r]] (= vecs[r]) and colors[dst[r]].  The rendering net is pointwise, so the
 *   result equals vfn_vf_render_fused16_fwd on the sorted samples. */
int vfn_vf_feat16_fwd(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                      float* out_vec, void* out_blocks, void* stream);
/* out_a[index[r]] = a[r] (and out_b[index[r]] = b[r] when b != NULL) for [n_rows,3] fp32 rows; negative indices are skipped.
 * Moves per-sample results computed in generation order to their positions among the sorted samples. */
int vfn_scatter_rows3(const float* a, const float* b, const int32_t* index, int64_t n_rows, float* out_a, float* out_b,
                      void* stream);
/* vfn_vf_render_fused16_fwd whose outputs of point m go to row out_index[m] of normals / colors (negative: dropped): the
 * N_f new samples of the split pipeline need no block round trip at all — they run the fused launch in generation order
 * (view direction of point m = ray_dirs[m / samples_per_ray]) and land at their sorted positions. */
int vfn_vf_render_fused16_scatter(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                  const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                                  int32_t samples_per_ray, const int32_t* out_index, float* normals, float* colors,
                                  void* stream);
/* The fused launch with the number of f16 products per fp32-equivalent product of its COLOUR BRANCH (the feature block of the
 * vector-field net and the whole rendering net) chosen by the caller: colour_products = 3 is vfn_vf_render_fused16_fwd /
 * _scatter; 2 evaluates that branch as w_hi x_hi + w_hi x_lo — weights as their f16 roundings, activations still split — which
 * removes 14 % of the launch's matrix instructions and a third of its weight traffic.  The vector head and every layer before
 * it keep three products, so normals (hence density, weights, depth, sample positions) are bit-identical to the 3-product
 * launch; colours differ from the reference by <= 2e-5 on its golden outputs (contract: 1e-4).  out_index: NULL (outputs in
 * place) or the scatter index.  Same packs as every other f16x3 entry point. */
int vfn_vf_render_fused16_products(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                   const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                                   int32_t samples_per_ray, const int32_t* out_index, int32_t colour_products, float* normals,
                                   float* colors, void* stream);
int vfn_render16_from_blocks(const vfn_net_geom* rn_geom, const void* rn_packed16, const void* blocks, const float* vecs,
                             const int32_t* dst, const float* points, const float* ray_dirs, int64_t n_rows,
                             int32_t samples_per_ray, float* normals, float* colors, void* stream);

/* From partial slabs to parameter gradients, all layer entries of a net in one launch: sums the `groups` slabs written
 * by vfn_weight_grad_partials(_bf16) and un-folds eval-mode BatchNorm and the skip scale (W' = s scale W,
 * b' = s (b - mu) + beta, s = gamma / sqrt(var + 1e-5)) onto the parameters the optimizer holds:
 * rows [row_off, row_off + rows) of g_w[out][in_dim] (columns act_c0.. and aux_c0..), g_b, and, with BatchNorm, g_bn_w, g_bn_b.
 * slab_rows = rows per slab group (256, or 32 for the 3-channel head).  At most 12 entries. */
typedef struct {
    const float* dw_act; const float* dw_aux; const float* db;
    const float* w; const float* b_lin; const float* bn_w; const float* bn_var; const float* bn_mean;
    float* g_w; float* g_b; float* g_bn_w; float* g_bn_b;
    int32_t rows, row_off, in_dim, slab_rows;
    int32_t act_c0, act_nc, aux_c0, aux_nc;
    float scale;
    int32_t groups_act, groups_aux, groups_db;   /* slabs of dw_act / dw_aux / db of THIS entry; 0: the call's `groups` (products batched
                                                  * into one launch, csrc/vfn_wgrad.hip, have fewer slabs than single ones) */
} vfn_unfold_entry;
int vfn_unfold_weight_grads(const vfn_unfold_entry* entries, int32_t n_entries, int32_t groups, void* stream);
/* The same with bit i of accumulate_mask saying that entry i ADDS its result to what g_w / g_b / g_bn_w / g_bn_b already hold:
 * the targets are then the parameters' own .grad tensors, and autograd's ~90 per-parameter accumulation kernels per training
 * step are not launched at all. */
int vfn_unfold_weight_grads_acc(const vfn_unfold_entry* entries, int32_t n_entries, int32_t groups, uint32_t accumulate_mask,
                                void* stream);

/* The parameter gradients of ONE network from a fragment-ordered workspace in ONE call (csrc/vfn_wgrad.hip): the
 * vfn_weight_grad_frag launches of every layer entry (activation columns, encoding columns), of the 3-channel head, and the
 * vfn_unfold_weight_grads_acc launch, issued on `stream` out of `scratch` (vfn_net_weight_grads_scratch_bytes; no allocation, no
 * synchronisation) — loss.backward() through vector_field_network.py:177-208 / rendering_network.py:62-108
 * (train/vector_field_nerf_train.py:252) after the dX chain has written the gradient slots.
 *   layers[geom->n_layers]: the reference's parameters per nn.Linear (+ BatchNorm1d) and where their gradients go;
 *   saved / dy: THIS net's first activation / gradient slot (slot stride slot_bytes = ceil(M/32) * 32 KiB);
 *   dy_form / x_form as in vfn_weight_grad_frag; feats: the tanh'ed features [M][256] fp32 rows (rendering net: input of its
 *   first layer; vector-field net: unused, may be NULL); aux: this net's encoding tile [M][40]; dz_head: [M][4];
 *   with_features = 0 (vector-field net only): the forward was vector-only, the feature block's rows get no gradient;
 *   accumulate != 0: results are ADDED to the gradient tensors (they are the parameters' .grad), else written.
 * vfn_weight_grad_groups(M) = the number of partial slabs per product the entry points use for M points. */
typedef struct vfn_wgrad_layer {
    const float* weight; const float* bias; const float* bn_weight; const float* bn_var; const float* bn_mean;
    float* g_weight; float* g_bias; float* g_bn_weight; float* g_bn_bias;
} vfn_wgrad_layer;
/* ... and the same for a SUBSET of the net's products: parts = VFN_WGRAD_LAYERS (the hidden layers) | VFN_WGRAD_FEATURES (the
 * feature block of the vector-field net's last Linear) | VFN_WGRAD_HEAD (its 3 output channels).  One workspace can hold points
 * whose forward included the feature block (the fine pass of render()) followed by points whose forward did not (the trainer's
 * supervision points, vector_field_nerf_train.py:191,203,215): LAYERS | HEAD then run once over all of them, FEATURES over the
 * first block only (pointers of a sub-range of points: slot base + (first / 32) * 32 KiB, aux + first * 40, dz_head + first * 4). */
#define VFN_WGRAD_LAYERS 1u
#define VFN_WGRAD_FEATURES 2u
#define VFN_WGRAD_HEAD 4u
int vfn_net_weight_grads_frag_part(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                                   const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                                   const float* aux, const float* dz_head, int64_t n_points, uint32_t parts, int32_t accumulate,
                                   void* scratch, void* stream);
int32_t vfn_weight_grad_groups(int64_t n_points);
int64_t vfn_net_weight_grads_scratch_bytes(int32_t net_kind, const vfn_net_geom* geom, int64_t n_points);
int vfn_net_weight_grads_frag(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                              const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                              const float* aux, const float* dz_head, int64_t n_points, int32_t with_features,
                              int32_t accumulate, void* scratch, void* stream);

/* VFLoss (models/losses/vf_loss.py:34-87) and the trainer's centre-ball selection (models/helpers/functions.py:137-157,
 * train/vector_field_nerf_train.py:203-214) fused (csrc/vfn_loss.hip): vfn_vf_loss_fwd = one reduction launch + a one-workgroup finish
 * -> out_terms[8] = {rgb, depth, unit_norm, supervision, norm_smaller_than_one, 0 (the directional-derivative term stays with the
 * caller), weighted total of the five, supervision rows}; vfn_vf_loss_bwd = one elementwise launch writing d total / d inputs times
 * the device scalar grad_out (NULL: 1).  rgb[N,3], depth[N] (has_depth), normals[M,3]; up to three supervision segments
 * sup_pred[k][n_sup[k],3] / sup_gt[k] (host arrays of three device pointers); ray_center != 0: the rows of `normals` whose point
 * (`points`[M,3]) lies within `radius` of `centroid` form a fourth segment with ground truth normalize(point - centroid) — selected
 * and counted on the device, no compaction, no host synchronisation.  `workspace`: vfn_vf_loss_workspace_bytes() bytes, written by
 * the forward call and read by the backward call (the supervision mean's data-dependent denominator lives there). */
typedef struct vfn_loss_params {
    int64_t n_rays, n_normals;
    int64_t n_sup[3];
    int32_t has_depth;       /* 0: no depth term (the batch has no depth ground truth) */
    int32_t smaller_on;      /* epoch >= norm_smaller_than_one_start */
    int32_t ray_center;
    int32_t reserved;
    float w_rgb, w_depth, w_unit, w_sup, w_smaller;
    float depth_clamp;
    float radius;
    float centroid[3];
} vfn_loss_params;
int64_t vfn_vf_loss_workspace_bytes(void);
int vfn_vf_loss_fwd(const vfn_loss_params* p, const float* rgb, const float* rgb_gt, const float* depth, const float* depth_gt,
                    const float* normals, const float* points, const float* const* sup_pred, const float* const* sup_gt,
                    void* workspace, float* out_terms, void* stream);
int vfn_vf_loss_bwd(const vfn_loss_params* p, const float* rgb, const float* rgb_gt, const float* depth, const float* depth_gt,
                    const float* normals, const float* points, const float* const* sup_pred, const float* const* sup_gt,
                    const void* workspace, const float* grad_out, float* d_rgb, float* d_depth, float* d_normals,
                    float* const* d_sup, void* stream);

/* Supervision points of the trainer (train/vector_field_nerf_train.py:186-214): n uniform samples in the spherical
 * shell r_min <= |p - centroid| <= r_max (models/samplers/sampler.py:160-193) and their radial unit ground truth
 * (models/helpers/functions.py:100-135): inward = 1 -> normalize(centroid - p) (sample_border_points), 0 -> normalize(p -
 * centroid) (sample_center_points).  u[n,3] supplies the three uniforms per sample (parity runs) or is NULL (Philox). */
int vfn_sample_sphere_shell(int64_t n, float r_min, float r_max, const float* centroid, int32_t inward, const float* u,
                            uint64_t seed, uint64_t offset, float* points, float* gt, void* stream);

/* The same forwards under autograd (train/vector_field_nerf_train.py:177,191,203,215): they additionally fill the
 * workspace the backward entry points read (`saved` slots, save_aux_vf[M,40], save_aux_rn[M,40]; see "slots" above),
 * exactly like vfn_vf_mlp_fwd_train / vfn_vf_render_fused_fwd_train.  with_features = 0 evaluates only the vector head
 * (the feature slot is not written); with_features = 1 also writes the 256 tanh'ed features into their slot, from which
 * the caller assembles [M, 3+F].  n_points < 2^22 per launch.
 * save_masks: the sign bits of every saved ReLU output (see vfn_mlp_bwd_chain_bf16).  save_f16 = flags.  Bit 0: the ReLU slots
 * are stored as f16 — 256 values in the first 512 bytes of every 1 KiB row, the row stride does not change — which halves
 * what the forward writes and the weight-gradient kernels read, at 11 instead of 24 significant bits in the activations that
 * multiply dY (BASELINE.json configs[2] trains on bf16 matrix cores); the tanh'ed feature slot (8) stays fp32 row-major.
 * Bit 1: the ReLU slots are FRAGMENT-ORDERED (see vfn_weight_grad_frag; slot stride ceil(M/32) * 32 KiB, n_points < 2^21 per launch, ws_points < 2^26).
 * Bit 2 (vfn_vf_mlp16_fwd_train[_at] with with_features = 0; the fused launch takes colour_products = 1 instead): SINGLE-PRODUCT
 * arithmetic — every K-block is ONE f16 product of the operands' f16 roundings (11 significant bits each, fp32 accumulation)
 * instead of three on split operands: the forward of the opt-in 16-bit-native training mode (BASELINE.json configs[2], "bf16
 * MFMA MLPs" as written — f16 keeps three more bits at the same matrix rate).  Same pack (its hi planes), same workspace; the
 * outputs are ~1e-3 from the fp32 ones, i.e. OUTSIDE the 1e-4 contract: never a default, never used by gradient-free renders. */
int vfn_vf_mlp16_fwd_train(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                           int32_t with_features, float* out_vec, float* saved, float* save_aux_vf, uint32_t* save_masks,
                           int32_t save_f16, void* stream);
int vfn_vf_mlp16_fwd_train_at(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                              int32_t with_features, float* out_vec, float* saved, float* save_aux_vf, uint32_t* save_masks,
                              int32_t save_f16, int64_t ws_first, int64_t ws_points, void* stream);    /* see ..._fused16_fwd_train_at */
int vfn_vf_render_fused16_fwd_train(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                    const void* rn_packed16, const float* points, const float* ray_dirs,
                                    int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                    float* saved, float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks, int32_t save_f16,
                                    void* stream);
/* The same launch filling PART of a workspace sized for ws_points points: its n_points points are points ws_first ..
 * ws_first + n_points - 1 of every slot, of the sign-bit words and of the encoding tiles (outputs normals / colors stay
 * launch-local, [n_points,3]).  A training render evaluates the vector-field net once per distinct sample this way: the proposal
 * samples (vector_field_nerf.py:252-256) and, after the fine sampler, the new samples (:294-297) fill ONE workspace in storage
 * order, which the chain and the weight-gradient kernels then walk once.  ws_first % 32 == 0 in fragment order.
 * colour_products: 3, or 2 = the colour branch of THIS forward on two products (vfn_vf_render_fused16_products): the loss sees
 * colours 2e-5 off, the saved activations move by less than their f16 storage rounds them, the backward is unchanged;
 * 1 = ONE product everywhere (see save_f16 bit 2 above). */
int vfn_vf_render_fused16_fwd_train_at(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                       const void* rn_packed16, const float* points, const float* ray_dirs,
                                       int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                       float* saved, float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks,
                                       int32_t save_f16, int64_t ws_first, int64_t ws_points, int32_t colour_products, void* stream);

/* =============================================================================================
 * Dense-grid stages between the vector-field queries and the mesh triangulation (evaluation/utils/mc_utils.py,
 * evaluation/utils/guassian_smoothing.py).  Grid cell (i,j,k) -> (i n + j) n + k; field vt[n^3,3]; n <= 1024.
 * ============================================================================================= */
/* evaluation/methods.py:194-208 fills samples[n^3,3] on the host with a SEPARABLE lattice: column 0 of cell (i,j,k) depends on i alone,
 * column 1 on j, column 2 on k (index * voxel_size + origin + translation + centroid, in fp32).  Given the three axis tables
 * axis0[n], axis1[n], axis2[n] (device; the caller reads them off the host tensor's first row / column / plane), writes
 * points[count,3] = rows [row0, row0 + count) of that grid — the caller's values bit for bit — so the 12 B per point of
 * mc_utils.get_set_predictions' upload (mc_utils.py:96-97) never crosses PCIe.  n <= 2048. */
int vfn_grid_lattice_points(const float* axis0, const float* axis1, const float* axis2, int32_t n, int64_t row0, int64_t count,
                            float* points, void* stream);
/* mc_utils.py:34-86: out[n^3] = 1 where the flux of the normalised field through the cell's 8 corners is <= threshold
 * (-0.5 in the reference), else 0; cells of the last planes are 0. */
int vfn_grid_divergence(const float* vt, int32_t n, float threshold, float* out, void* stream);
/* guassian_smoothing.py:81-97: one axis of the separable Gaussian (k odd, <= 15 host weights), replicate padding;
 * three calls (axis 0, 1, 2) = smooth_vf.  Out of place. */
int vfn_grid_smooth_axis(const float* in, float* out, int32_t n, int32_t axis, const float* weights_host, int32_t k,
                         void* stream);
/* mc_utils.py:107-167: for cells with divergence == 1, which of the cell's two most opposed corner vectors each of the 8
 * corners sides with (0/1; corner order (0,0,0) (0,1,0) (1,1,0) (1,0,0) (0,0,1) (0,1,1) (1,1,1) (1,0,1)); other cells 0.
 * vt is the normalised field [n^3,3]. */
int vfn_grid_unify_direction(const float* divergence, const float* vt, int32_t n, int64_t* choice, void* stream);
/* mc_utils.py:170-223: for the 28 corner pairs (a<b) of every cell: different_side[n^3,28] = choice_a != choice_b,
 * pair_norms[n^3,28,2] = (norms at corner a, norms at corner b), corners outside the grid read as 0. */
int vfn_grid_comb_format(const int64_t* choice, const float* norms, int32_t n, float* different_side, float* pair_norms,
                         void* stream);
/* The same two stages with the side bits of a cell as ONE byte (bit q = corner q's side): `sides`[n^3] is written beside (choice
 * != NULL) or instead of (choice == NULL) the int64 table, and vfn_grid_comb_format_sides reads it instead of 64 B per cell.  What
 * evaluation/methods.py:248-253 does with the table — hand it to make_comb_format and delete it — needs nothing else. */
int vfn_grid_unify_direction_sides(const float* divergence, const float* vt, int32_t n, uint8_t* sides, int64_t* choice, void* stream);
int vfn_grid_comb_format_sides(const uint8_t* sides, const float* norms, int32_t n, float* different_side, float* pair_norms,
                               void* stream);

/* =============================================================================================
 * Optimizer side of a training step over ONE flat fp32 buffer (train/vector_field_nerf_train.py:254-260:
 * torch.nn.utils.clip_grad_norm_(model.parameters(), clip); optimizer.step()).  The unique parameters — and their gradients
 * and Adam moments — are laid out contiguously, sorted into up to four REGIONS [start, end) of equal multiplicity `mult` =
 * how often the parameter occurs in the optimizer's list (the reference lists every vector-field parameter twice,
 * models/nerf/vector_field_nerf.py:57-63): a gradient counts mult times in the norm, is scaled mult times, and its parameter
 * receives mult consecutive Adam updates per step, exactly what the sequential per-entry loops do.
 * vfn_flat_clip_grad_norm: out2[0] = total 2-norm, out2[1] = min(max_norm / (norm + 1e-6), 1); gradients scaled in place.
 *   workspace: vfn_flat_clip_workspace_bytes() bytes of device memory, zero-filled once by the caller.
 * vfn_flat_adam_step: step_size[2 r + k] = lr / (1 - beta1^t), bc2_sqrt[2 r + k] = sqrt(1 - beta2^t) for update k (t = the step
 *   number of that update) of region r, evaluated by the caller in double precision as torch.optim.Adam does. */
int64_t vfn_flat_clip_workspace_bytes(void);
int vfn_flat_clip_grad_norm(float* flat_grad, int64_t n, int32_t n_regions, const int64_t* starts, const int64_t* ends,
                            const int32_t* mults, float max_norm, void* workspace, float* out2, void* stream);
int vfn_flat_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t n_regions,
                       const int64_t* starts, const int64_t* ends, const int32_t* mults, const double* step_size,
                       const double* bc2_sqrt, double beta1, double beta2, double eps, double weight_decay, void* stream);

/* =============================================================================================
 * ONE training step in ONE call (csrc/vfn_train.hip): the reference trainer's loop body, train/vector_field_nerf_train.py:172-260 in the
 * regime every shipped scene runs (eval-mode networks, :140-141; border + centre supervision, :186-216; VFLoss; zero_grad; backward;
 * clip_grad_norm_ over the duplicated parameter list; Adam.step) issued from C on one stream out of one caller-supplied workspace —
 * what vfn_render_fwd is to the gradient-free render().  It SEQUENCES entry points of this header (the f16x3 saving forwards with one
 * vector-field evaluation per distinct sample, vfn_vf_loss_fwd / _bwd, the bf16 chains over one fragment-ordered workspace shared by
 * the fine pass and the supervision batch, vfn_net_weight_grads_frag_part, vfn_flat_clip_grad_norm, vfn_flat_adam_step, the re-packs),
 * so every value is what those entry points produce called one by one.
 *   phases: VFN_TRAIN_FORWARD_BACKWARD ends with every parameter's gradient ADDED into flat_grad (zeroed first); a multi-rank caller
 *           all-reduces flat_grad, then calls again with VFN_TRAIN_OPTIMIZER (clip, Adam, re-pack).  Both bits: the whole step.
 *   render: sizes, samplers, density, Philox (seed, offset) of the three draws (laid out coarse | fine | add, as vfn_render_fwd);
 *           colour_products / separate_launches / timing_events are ignored; streams >= 2: the supervision batch's forward and chain
 *           run on an internal side stream forked from / joined into `stream` (same values).  N S_c and N S_t must be multiples of 32.
 *   loss:   weights, clamp and flags of VFLoss; n_rays / n_normals / n_sup are filled in by the call (ONE supervision segment: the
 *           border batch followed by the centre batch; ray_center selects the ray samples inside the centre ball on the device).
 *   n_sup:  points per supervision batch (the trainer uses (N S_t) / 10); border / center: which batches exist; the shells are
 *           [border_r_min, border_r_max] (inward ground truth) and [0, sup_radius] (outward) around sup_centroid; their draws come
 *           from the Philox stream (sup_seed, sup_offset) — border first — unless sup_u_border / sup_u_center supply them ([n_sup,3]).
 *   save_flags / dy_flags / dy_form / x_form: the storage forms of the training workspace (vfn_vf_render_fused16_fwd_train_at,
 *           vfn_mlp_bwd_chain_bf16_ws_at, vfn_weight_grad_frag); fragment order (save_flags bit 1) is required.
 *   forward_products: colour_products of the saving forwards (3; 2; 1 = the single-product mode, which also takes dy_flags bit 4).
 *   regions / step_size / bc2_sqrt / betas / eps / weight_decay / max_norm: as vfn_flat_clip_grad_norm and vfn_flat_adam_step.
 *   repack: re-pack vf_packed16, rn_packed16 and the two transposed bf16 packs from the updated parameters at the end of phase 2.
 *   sparse_colours: a sample's colour enters the step through rgb = sum_s w_s c_s only, so where w_s is exactly zero — a closed density
 *           ReLU, or a transmittance that has underflowed: 93-97 % of the samples of a batch — neither the colour nor its gradient
 *           (d c_s = w_s d rgb) nor the rendering net's share of that sample in any weight gradient is needed, and every derivative of w_s
 *           that would multiply c_s ends at the closed ReLU / the zero transmittance.  With sparse_colours the step evaluates the
 *           vector-field net on all samples with its vector-only saving forward (region 1 of the workspace), selects on the device the
 *           samples whose colour can reach an output or a gradient — w > 0, or w = 0 by an underflowed alpha alone (sigma > 0, delta > 0
 *           and a non-zero transmittance: alpha = 1 - exp(-sigma delta) rounds to 0 for sigma delta < 3e-8 while d w / d sigma = T delta
 *           does not, and the dense step's (d rgb . c_s) T delta term of that sample's d sigma needs c_s) — a count the host never learns
 *           (launches are sized for the capacity and cut inside the kernels), and runs the fused saving forward, the fused chain and the
 *           colour branch's weight gradients on that compacted list only (region 2).
 *           The upstream gradient splits between the regions (d normals on region 1, d colours on region 2): the parameter gradients are
 *           the dense step's up to the order of their sums; rgb, depth, weights, normals are the dense step's bit for bit; the `colors`
 *           output holds zeros on the samples that were not selected (vf_nerf_amd fills them on first access of NerfOutput.coarse_colors).
 * Outputs of phase 1 (caller-allocated, the NerfOutput of the step's render): ray_dirs[N,3], z_vals[N,S_t], points[N,S_t,3],
 * normals[N S_t,3], colors[N S_t,3], weights[N,S_t], rgb[N,3], depth[N]; out_terms[8] as vfn_vf_loss_fwd; out_counts[2] (optional) =
 * the number of samples the colour branch was evaluated for (sparse_colours: the selected ones) and N S_t.  Phase 2: out_norm[2] as
 * vfn_flat_clip_grad_norm.  No allocation, no synchronisation, no host read-back.
 *
 * SESSION FORM (ABI 4): the same step for a caller that makes the reference trainer's calls ONE BY ONE — render(), the two supervision
 * forwards, the loss, backward(), clip, step (train/vector_field_nerf_train.py:177-260 unchanged, through vf_nerf_amd.dropin) — and still wants
 * the step's launches: same workspace, same kernels, the caller's own loss in the middle.
 *   VFN_TRAIN_RENDER    the render() part of phase 1 alone (prep, rays, saving forwards, selection, composite).  sup_rows_reserved (a
 *                       multiple of 32) rows of the workspace are set aside for supervision batches appended later; their points and
 *                       upstream-gradient rows are zeroed.
 *   vfn_train_step_supervision_points / _forward: one supervision batch into rows [row0, row0 + count) of that region (sampled by the call as
 *                       phase 1 samples them) and its vector-only saving forward over rows [row0, pad32(row0 + count)) (row0 a multiple of 32).
 *                       With render.streams >= 2 they run on the calling thread's side stream, ordered after this step's prep launch, beside
 *                       the render's launches, and are joined into `stream` before the call returns.  The caller reads predictions / ground
 *                       truth / points at the offsets vfn_train_step_workspace_layout reports and leaves the loss's gradient with respect to the
 *                       predictions in the D_SUP rows.
 *   VFN_TRAIN_BACKWARD  the backward part of phase 1 alone, from upstream gradients the caller supplies: io->d_rgb_in[N,3], io->d_depth_in[N]
 *                       (or NULL), io->d_normals_in[N S_t,3] over the SORTED samples (copied into the workspace's DN rows unless it already is
 *                       that address) and the D_SUP rows.  Gradients are ADDED into flat_grad / g_beta / g_mean / g_scale (the caller's
 *                       zero_grad() decides what they start from).
 * vfn_train_step_workspace_layout: out[VFN_TWS_*] for the given sizes — byte offsets into the workspace, or counts. */
#define VFN_TRAIN_FORWARD_BACKWARD 1
#define VFN_TRAIN_OPTIMIZER 2
#define VFN_TRAIN_RENDER 4
#define VFN_TRAIN_BACKWARD 8
#define VFN_TRAIN_CLIP 16         /* the two halves of VFN_TRAIN_OPTIMIZER: clip_grad_norm_ (out_norm) ...                 */
#define VFN_TRAIN_ADAM 32         /* ... and optimizer.step (+ the re-pack), for a caller that makes them as two calls */
#define VFN_TWS_SUP_PTS 0        /* float[rows][3]  supervision points                                            */
#define VFN_TWS_SUP_GT 1         /* float[rows][3]  their radial ground truth                                     */
#define VFN_TWS_SUP_PRED 2       /* float[rows][3]  vector head of the supervision rows                           */
#define VFN_TWS_D_SUP 3          /* float[rows][3]  upstream gradient of those predictions                        */
#define VFN_TWS_DN 4             /* float[N S_t][3] upstream gradient of the sorted normals                       */
#define VFN_TWS_SUP_ROWS 5       /* count: rows of the supervision region                                         */
#define VFN_TWS_TOTAL_ROWS 6     /* count: points of one fragment-ordered slot                                    */
#define VFN_TWS_BYTES 7          /* = vfn_train_step_workspace_bytes                                              */
#define VFN_TWS_COUNT 8
typedef struct vfn_train_step_params {
    vfn_render_params render;
    vfn_loss_params loss;
    int64_t n_sup;
    int32_t border, center;
    float sup_centroid[3];
    float sup_radius;                   /* centre ball [0, sup_radius] (outward ground truth); also the radius of the ray_center selection */
    float border_r_min, border_r_max;   /* border shell (inward ground truth): far - 5 radius .. far, evaluated by the host in double as Python does */
    int32_t phases;
    uint64_t sup_seed, sup_offset;
    int32_t save_flags, dy_flags, dy_form, x_form;
    int32_t forward_products;
    int32_t repack;
    int32_t n_regions;
    int32_t mults[4];
    int64_t starts[4], ends[4];
    double step_size[8], bc2_sqrt[8];
    double beta1, beta2, eps, weight_decay;
    float max_norm;
    int32_t sparse_colours;             /* != 0: the colour branch (feature block + rendering net) is evaluated, differentiated and summed into the
                                         * weight gradients only for the samples whose weight is non-zero (see below) */
    int64_t sup_rows_reserved;          /* session form: rows (a multiple of 32) of the supervision region; 0: pad32(n_sup (border + center)) */
    int64_t sup_rows_used;              /* session form, VFN_TRAIN_BACKWARD: the leading rows (a multiple of 32, <= sup_rows_reserved) of that region that
                                         * supervision forwards have filled — the chain and the weight gradients walk these and no others (a row no
                                         * forward has written holds no activations) */
} vfn_train_step_params;
typedef struct vfn_train_step_io {
    const vfn_net_geom* vf_geom; const vfn_net_geom* rn_geom;
    void* vf_packed16; void* rn_packed16; void* vf_packed_bwd16; void* rn_packed_bwd16;       /* read by phase 1, rewritten by the re-pack */
    const vfn_layer_params* vf_layers; const vfn_layer_params* rn_layers;                     /* host arrays [n_layers]: the re-pack's inputs */
    const vfn_wgrad_layer* vf_wgrad; const vfn_wgrad_layer* rn_wgrad;                         /* host arrays [n_layers]: parameters and where their gradients go */
    const float* vf_head_w; const float* rn_head_w;                                           /* rows 0..2 of each net's last Linear */
    const float* beta; const float* mean; const float* scale;                                 /* the density's three raw scalars ... */
    float* g_beta; float* g_mean; float* g_scale;                                             /* ... and their gradient words (inside flat_grad) */
    float* flat_param; float* flat_grad; float* exp_avg; float* exp_avg_sq; int64_t n_flat;   /* the optimizer's flat buffers (vfn_flat_adam_step) */
    void* clip_workspace;
    const float* uv; const float* pose; const float* intrinsics; const float* t_vals;
    const float* far_coarse_per_ray; const float* far_fine_per_ray;
    const float* u_coarse; const float* u_fine; const float* u_add;                           /* optional supplied draws (parity runs) */
    const float* sup_u_border; const float* sup_u_center;
    const float* rgb_gt; const float* depth_gt;
    void* workspace;                                                                          /* vfn_train_step_workspace_bytes() bytes */
    float* ray_dirs; float* z_vals; float* points; float* normals; float* colors; float* weights; float* rgb; float* depth;
    float* out_terms; float* out_norm;
    float* out_counts;                                                                        /* optional [2]: samples the colour branch ran on, all samples */
    const float* d_rgb_in; const float* d_depth_in; const float* d_normals_in;                /* VFN_TRAIN_BACKWARD: the caller's upstream gradients */
} vfn_train_step_io;
int64_t vfn_train_step_workspace_bytes(const vfn_train_step_params* p, const vfn_net_geom* vf_geom, const vfn_net_geom* rn_geom);
int vfn_train_step(const vfn_train_step_params* p, const vfn_train_step_io* io, void* stream);
int vfn_train_step_workspace_layout(const vfn_train_step_params* p, const vfn_net_geom* vf_geom, const vfn_net_geom* rn_geom, int64_t* out, int32_t n_out);
/* vfn_sample_sphere_shell into rows [row0, row0 + count) of the open step's supervision region (points and ground truth).  The centre is
 * given BY VALUE (cx, cy, cz) — then, with Philox draws (u == NULL), the launch depends on nothing the caller's stream produced after the
 * step's prep and runs on the side stream — or as a device pointer centroid_dev / with supplied draws u[count,3], and then on `stream`.
 * Returns 1 (not an error) when it ran on the side stream, 0 when on `stream`: pass that as on_side to the forward of the same rows.
 * vfn_train_step_supervision_forward: the vector-only saving forward over rows [row0, pad32(row0 + count)), predictions into the SUP_PRED rows. */
int vfn_train_step_supervision_points(const vfn_train_step_params* p, const vfn_train_step_io* io, int32_t inward, float r_min, float r_max,
                                      float cx, float cy, float cz, const float* centroid_dev, int64_t row0, int64_t count, const float* u,
                                      uint64_t seed, uint64_t offset, void* stream);
int vfn_train_step_supervision_forward(const vfn_train_step_params* p, const vfn_train_step_io* io, int64_t row0, int64_t count, int32_t on_side,
                                       void* stream);
/* The backward of supervision rows [row0, pad32(row0 + count)) ALONE (their upstream gradient in the D_SUP rows): the vector-only chain and the
 * vector-field net's weight gradients of those rows, added into flat_grad; the D_SUP rows are zeroed afterwards.  For a backward pass that
 * never reaches VFN_TRAIN_BACKWARD (a loss that uses the supervision predictions only). */
int vfn_train_step_supervision_backward(const vfn_train_step_params* p, const vfn_train_step_io* io, int64_t row0, int64_t count, void* stream);

/* =============================================================================================
 * Networks in TRAINING mode: nn.BatchNorm1d with batch statistics (vector_field_network.py:146-208 and
 * rendering_network.py:62-108 after VectorFieldNerf.train(), vector_field_nerf.py:139-150; the trainer enters it when the
 * directional-derivative loss weight is non-zero, train/vector_field_nerf_train.py:140-141).  Batch statistics couple all
 * rows of a layer, so the layers run one launch at a time on row-major fp32 matrices in HBM.  Leading dimensions are
 * multiples of 4 floats, rows 16-byte aligned, pad columns hold zeros.
 * ============================================================================================= */
/* (transpose_w is a bit field: bit 0 = the transposition below; bits 1-2 = the arithmetic — 0: exact fp32 (v_mfma_f32_32x32x2f32); 2: SPLIT
 *  f16, every operand as two f16 halves and a product as a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16 with fp32
 *  accumulation, 22 significant bits, for forward GEMMs on normalised activations (A rides at 64x its value so that its low halves stay
 *  out of the f16 denormals: |A| must stay below 1023, |W| below 65504 — batch-normalised activations, encodings, points and weights
 *  do; gradients do not, hence form 6 for the backward products); 4: SPLIT bf16, the same with bf16 halves, 16 significant
 *  bits and fp32's exponent range; 6: bf16 in THREE parts (hi | mid | lo = 24 bits), six products: fp32-equivalent at fp32's exponent
 *  range, for backward GEMMs on gradients of any magnitude.  5.3x / 5.3x / 2.7x fewer matrix cycles than the exact form; A goes through LDS
 *  in whole cache lines in these forms.)
 * transpose_w = 0: C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]), W = nn.Linear weight [n_out][ldw], k_in columns
 *                  (torch.nn.functional.linear); act 0 none / 1 tanh / 2 sigmoid.
 * transpose_w = 1: C[m][j] = sum_k A[m][k] W[k][j], W [k_in][ldw], j < n_out  (the input gradient dX = dZ W).
 * A is read over k rounded up to a multiple of 8 (lda must cover it, pad columns zero).
 * stats_part (or NULL): [vfn_linear_rows_stat_parts(m)][2][n_out] per-workgroup column sums of the pre-activation z and
 * of z^2 — the batch statistics, finished by vfn_colsum_finish. */
int vfn_linear_rows(int32_t transpose_w, const float* a, int32_t lda, const float* w, int32_t ldw, const float* bias,
                    int64_t m, int32_t n_out, int32_t k_in, int32_t act, float* c, int32_t ldc, float* stats_part,
                    void* stream);
/* vfn_linear_rows with a scratch for W's 16-bit planes (vfn_linear_rows_wplanes_bytes(n_out, k_in) bytes, 16-byte aligned; NULL: as
 * vfn_linear_rows).  The split arithmetics with two planes (transpose_w & 6 = 2, 4) then split W ONCE per call — one small launch in front of
 * the product — instead of once per workgroup and chunk; same values bit for bit.  The scratch is free again when the call's launches have run. */
int vfn_linear_rows_ws(int32_t transpose_w, const float* a, int32_t lda, const float* w, int32_t ldw, const float* bias,
                       int64_t m, int32_t n_out, int32_t k_in, int32_t act, float* c, int32_t ldc, float* stats_part,
                       void* wplanes, void* stream);
int64_t vfn_linear_rows_wplanes_bytes(int32_t n_out, int32_t k_in);
int64_t vfn_linear_rows_stat_parts(int64_t m);
/* A forward layer product with the PREVIOUS layer's BatchNorm + ReLU folded into its operand read (round 6, ABI 5):
 * C[M, n_out] = act(z_prev) W^T + bias with act(z)[k] = post_prev * max(z[k] * scale[k] + shift[k], 0) for k < n_prev (coef_prev =
 * [4][n_prev]) and post_prev * z[k] for n_prev <= k < k_in; arith 2 (three f16 products on split operands); 129 <= n_out <= 256;
 * stats_part as vfn_linear_rows_ws; wplanes (its scratch for W's planes) is required.  Replaces the vfn_bstat_relu_rows pass + vfn_linear_rows_ws pair of a training-mode layer
 * (models/vector_field/vector_field_network.py:177-208 with batch statistics): same operand values, one [M, 256] write and read less. */
int vfn_linear_rows_fold(int32_t arith, const float* z_prev, int32_t ldz, const float* coef_prev, int32_t n_prev, float post_prev,
                         const float* w, int32_t ldw, const float* bias, int64_t m, int32_t n_out, int32_t k_in, float* c, int32_t ldc,
                         float* stats_part, void* wplanes, void* stream);
/* The input-gradient product C[m][n_out] = dZ[m][k_in] W[k_in][n_out] (vfn_linear_rows with transpose_w = 1 | arith; arith = 4: three
 * bf16 products, 16 significant bits at fp32's exponent range — the arithmetic of the eval-mode dX chain, the default since round 5 — or
 * 6: bf16 in three parts, six products, 24 bits) that
 * also leaves, while C is in registers, the per-workgroup partials sums_part[vfn_linear_rows_stat_parts(m)][2][n_prev] of sum g' and
 * sum g' x_hat of the PREVIOUS layer's BatchNorm backward over the first n_prev (<= 256) columns of C — g' = post_prev C [z_prev scale +
 * shift > 0], x_hat = (z_prev - mean) rstd with coef_prev [4][n_prev] = scale, shift, mean, rstd — i.e. what vfn_bstat_relu_bwd_sums
 * computes in a pass of its own; finished by vfn_colsum_finish(sums_part, parts, 2 n_prev).  wplanes: as vfn_linear_rows_ws (or NULL).
 * Reference: the autograd of nn.Linear + nn.BatchNorm1d (training) + ReLU, models/vector_field/vector_field_network.py:176-208. */
int vfn_linear_rows_dx_sums(const float* dz, int32_t lddz, const float* w, int32_t ldw, int64_t m, int32_t n_out, int32_t k_in, float* c,
                            int32_t ldc, const float* z_prev, int32_t ldz_prev, const float* coef_prev, int32_t n_prev, float post_prev,
                            float* sums_part, int32_t arith, void* wplanes, void* stream);
/* Partials per workgroup of the row-wise kernels below (vfn_bstat_relu_bwd_sums). */
int64_t vfn_bstat_row_parts(int64_t m);
/* part[n_parts][width] fp32 -> sums[width] double. */
int vfn_colsum_finish(const float* part, int64_t n_parts, int32_t width, double* sums, void* stream);
/* sums[2][n] (sum z, sum z^2 over m rows) -> coef[4][n] = scale (gamma rstd), shift (beta - mean scale), mean, rstd with
 * rstd = 1 / sqrt(biased variance + eps); running_mean / running_var (may be NULL) <- (1 - momentum) old + momentum batch,
 * the variance unbiased (m / (m - 1)), as torch.nn.BatchNorm1d does in training mode. */
int vfn_bstat_finalize(const double* sums, int64_t m, int32_t n, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* coef, void* stream);
/* h[m][c] = post_scale * relu(z[m][c] * scale[c] + shift[c]), c < n. */
int vfn_bstat_relu_rows(const float* z, int32_t ldz, const float* coef, int64_t m, int32_t n, float post_scale, float* h,
                        int32_t ldh, void* stream);
/* Backward of the above through the batch statistics.  g = gradient wrt h.  With g' = post_scale g [z scale + shift > 0] and
 * x_hat = (z - mean) rstd:  part[vfn_bstat_row_parts(m)][2][n] = per-workgroup sums of g' and g' x_hat (= d beta, d gamma
 * after vfn_colsum_finish);  dz = gamma rstd (g' - sum g' / m - x_hat sum(g' x_hat) / m). */
int vfn_bstat_relu_bwd_sums(const float* g, int32_t ldg, const float* z, int32_t ldz, const float* coef, int64_t m, int32_t n,
                            float post_scale, float* part, void* stream);
int vfn_bstat_relu_bwd_rows(const float* g, int32_t ldg, const float* z, int32_t ldz, const float* coef, const double* sums,
                            int64_t m, int32_t n, float post_scale, float* dz, int32_t lddz, void* stream);
/* dz = dy (1 - y^2) (act 1, tanh) or dy y (1 - y) (act 2, sigmoid) or dy (act 0).  dy == NULL: dy is 1 in column
 * onehot_col and 0 elsewhere — the grad_outputs of the three autograd.grad calls of vector_field_network.py:150-171. */
int vfn_act_bwd_rows(int32_t act, const float* dy, int32_t lddy, const float* y, int32_t ldy, int64_t m, int32_t n,
                     int32_t onehot_col, float* dz, int32_t lddz, void* stream);
/* dst[row][col0 + j] = scale * PE(src3[row / rows_per_src])[j], j < 3 + 6 multires (models/helpers/embedder.py:11-37;
 * multires 0 copies the 3 values). */
int vfn_embed_rows(const float* src3, int32_t ld_src, int32_t rows_per_src, int64_t m, int32_t multires, float scale,
                   float* dst, int32_t ld_dst, int32_t col0, void* stream);
/* d_src3[m][3] (+)= sum over the (up to two) places the encoding was used of scale * dPE/dx applied to the gradient
 * rows d[row][col..col+3+6 multires) (d_b may be NULL). */
int vfn_embed_rows_bwd(const float* src3, int64_t m, int32_t multires, const float* d_a, int32_t ld_a, int32_t col_a,
                       float scale_a, const float* d_b, int32_t ld_b, int32_t col_b, float scale_b, float* d_src3,
                       int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFN_H */
