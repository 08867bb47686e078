// vfn_train.hip — ONE training step of the reference trainer's loop body as one entry point.
//
// train/vector_field_nerf_train.py:172-260 (the branch every shipped scene takes: eval-mode networks, :140-141; border + centre
// supervision; VFLoss; zero_grad; backward; clip_grad_norm_ over the duplicated parameter list; Adam.step) is ~60 Python-side
// operations per step in the facade (vf_nerf_amd/trainer.py::TrainStep through backward.py's autograd functions): at the reference's
// batch size (1 024 rays) that is 2.2 ms of host time for a step whose kernels take 2.4 ms — eight ranks on one host would be
// host-bound.  vfn_train_step issues the same launches, in the same order, from C on one stream out of ONE caller-supplied
// workspace (vfn_train_step_workspace_bytes), the way vfn_render_fwd does for the gradient-free render:
//
//   phase 1 (VFN_TRAIN_FORWARD_BACKWARD)
//     prep                      density scalars gathered, centroid, padding rows and the three scalar gradients zeroed
//     rays + proposal samples   utils/rendering.py:12-60, ray_sampler.py:49-80,113-142        (draws generated in place or supplied)
//     saving fused forward      vector_field_nerf.py:252-256 on the S_c proposal samples -> workspace rows [0, N S_c)
//     weights -> argmax -> fine ray_sampler.py:264-302 with provenance (src / dst)
//     saving fused forward      :294-297,315-318 on the N_f new samples -> workspace rows [N S_c, N S_t)
//     density + composite       :308-323, gathering every sample from storage order to its sorted position
//     supervision points        train.py:186-214 (functions.py:100-135): border shell, centre ball -> one batch, padded to 32
//     saving vector-only fwd    train.py:201,213 on both batches -> workspace rows [N S_t, N S_t + pad32(2 n_sup))
//     VFLoss forward            vf_loss.py:34-87 (+ the centre-ball rows of functions.py:137-157, selected on the device)
//     zero the flat gradient    train.py:251
//     VFLoss backward           train.py:252 ...
//     dX chain, supervision     ... through the vector-field net over the supervision rows
//     per-ray backward          d rgb / d depth (+ the loss's d normals) -> d colours, d normals, the three density scalars
//     gather to storage order   row r of the workspace is sorted sample dst[r]
//     dX chain, fine pass       rendering net -> feature hand-off -> vector-field net over rows [0, N S_t)
//     weight gradients          rendering net (rows of the fine pass), vector-field net (hidden layers + head over ALL rows,
//                               feature block over the fine pass's), un-folded and ADDED into the flat gradient
//     density scalar gradients  added into the flat gradient
//   [the caller all-reduces the flat gradient here when there is more than one rank: distributed.GradientBucket]
//   phase 2 (VFN_TRAIN_OPTIMIZER)
//     clip_grad_norm_           train.py:254-255 over the duplicated list (Q4)       vfn_flat_clip_grad_norm
//     Adam                      train.py:258, sequential semantics                    vfn_flat_adam_step
//     re-pack                   the four weight packs the next step reads
//
// Nothing here computes: it sequences entry points and internal launches of this library on slices of one workspace, so every
// value is what the launch-by-launch path produces (tests/test_hip_trainer.py::test_one_call_training_step_equals_the_launch_by_launch_step).
#include <string.h>
#include "vfn_common.h"
#include "vfn_plan.h"

namespace {

struct Carve {
    unsigned char* base;
    size_t off;
    template <typename T> T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += ((count * sizeof(T) + 255) / 256) * 256;
        return p;
    }
};

constexpr size_t GROUP_BYTES = 32768;      // one group of 32 points of a fragment-ordered slot

struct Ws {
    // render intermediates
    float *directions, *cam_loc, *z_c, *pts_c, *new_pts;
    int32_t *dst, *src;
    // per-sample results in STORAGE order [proposal samples | new samples] and their gradients
    float *normals_s, *colors_s, *dn_s, *dc_s;
    // sorted per-sample gradients
    float *dn, *dc;
    // supervision batch (padded to whole groups of 32 points)
    float *sup_pts, *sup_gt, *sup_pred, *d_sup;
    // loss
    float *d_rgb, *d_depth;
    void* loss_ws;
    // scalars: [beta, mean, scale] gathered, their gradients, the supervision centroid
    float *scal, *dscal, *centroid;
    // training workspace
    unsigned char *saved, *dy;
    uint32_t* masks;
    float *aux_vf, *aux_rn, *dz_vec, *dz_rgb;
    void *scratch_vf, *scratch_rn, *scratch_vf2;      // (vf2: region 2's vector-field products, beside region 1's on the side stream)
    // sparse colour branch (region 2 = the samples with non-zero weight, compacted): selection and per-selected-sample buffers
    int32_t *cnt, *off, *k_dev, *sel_sorted;
    float *pts_sel, *dirs_sel, *normals_sel, *colors_sel, *dc_sel, *zero3;
    long long cap, r2_first;
    size_t slot_bytes;
    long long m_c, m, m_sup, m_sup_pad, total;
    int vf_h, rn_h;
    size_t bytes;
};

long long pad32(long long n) { return (n + 31) / 32 * 32; }

int hidden_entries(int net_kind, const vfn_net_geom* g, const char* what) {
    VfnNetPlan plan;
    char err[256] = {0};
    int rc = vfn_make_plan(net_kind, g, &plan, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("%s: %s", what, err); return rc; }
    return plan.n_hidden;
}

// `out` (the caller's per-sample / per-ray outputs) are separate buffers: the workspace holds only what a step needs internally
int carve(void* workspace, const vfn_train_step_params* p, const vfn_net_geom* vf_geom, const vfn_net_geom* rn_geom, Ws* w) {
    const long long n = p->render.n_rays, sc = p->render.n_coarse, nf = p->render.n_fine, st = sc + nf;
    w->vf_h = hidden_entries(VFN_NET_VF, vf_geom, "vfn_train_step");
    if (w->vf_h < 0) return w->vf_h;
    w->rn_h = hidden_entries(VFN_NET_RENDER, rn_geom, "vfn_train_step");
    if (w->rn_h < 0) return w->rn_h;
    w->m_c = n * sc; w->m = n * st;
    w->m_sup = (long long)p->n_sup * ((p->border ? 1 : 0) + (p->center ? 1 : 0));
    w->m_sup_pad = pad32(w->m_sup);
    if (p->sup_rows_reserved > 0) {            // session form: the caller appends its batches to a region of this many rows
        if (p->sup_rows_reserved % 32) { vfn_set_error("vfn_train_step: sup_rows_reserved must be a multiple of 32"); return VFN_ERR_INVALID; }
        w->m_sup = w->m_sup_pad = p->sup_rows_reserved;
    }
    // sparse colour branch: region 2 can hold every sample (the count of samples with w > 0 is known to the device only)
    w->cap = p->sparse_colours ? pad32(w->m) : 0;
    w->r2_first = w->m + w->m_sup_pad;
    w->total = w->m + w->m_sup_pad + w->cap;
    const int slots = w->vf_h + w->rn_h;
    w->slot_bytes = (size_t)((w->total + 31) / 32) * GROUP_BYTES;
    Carve c{static_cast<unsigned char*>(workspace), 0};
    w->directions = c.take<float>(n * 3);
    w->cam_loc = c.take<float>(n * 3);
    w->z_c = c.take<float>(n * sc);
    w->pts_c = c.take<float>(n * sc * 3);
    w->new_pts = c.take<float>(n * nf * 3);
    w->dst = c.take<int32_t>(w->m);
    w->src = c.take<int32_t>(w->m);
    // the vector outputs of region 1 and of the supervision batch are ONE array (normals_s | sup_pred), their upstream gradients
    // another (dn_s | d_sup): the sparse path runs one vector-only chain over both
    w->normals_s = c.take<float>((w->m + w->m_sup_pad) * 3);
    w->sup_pred = w->normals_s + w->m * 3;
    w->colors_s = c.take<float>(w->m * 3);
    w->dn_s = c.take<float>((w->m + w->m_sup_pad) * 3);
    w->d_sup = w->dn_s + w->m * 3;
    w->dc_s = c.take<float>(w->m * 3);
    w->dn = c.take<float>(w->m * 3);
    w->dc = c.take<float>(w->m * 3);
    w->sup_pts = c.take<float>(w->m_sup_pad * 3);
    w->sup_gt = c.take<float>(w->m_sup_pad * 3);
    w->cnt = c.take<int32_t>(p->sparse_colours ? n : 0);
    w->off = c.take<int32_t>(p->sparse_colours ? n : 0);
    w->k_dev = c.take<int32_t>(p->sparse_colours ? 4 : 0);
    w->sel_sorted = c.take<int32_t>(w->cap);
    w->pts_sel = c.take<float>(w->cap * 3);
    w->dirs_sel = c.take<float>(w->cap * 3);
    w->normals_sel = c.take<float>(w->cap * 3);
    w->colors_sel = c.take<float>(w->cap * 3);
    w->dc_sel = c.take<float>(w->cap * 3);
    w->zero3 = c.take<float>(w->cap * 3);
    w->d_rgb = c.take<float>(n * 3);
    w->d_depth = c.take<float>(n);
    w->loss_ws = c.take<unsigned char>((size_t)vfn_vf_loss_workspace_bytes());
    w->scal = c.take<float>(4);
    w->dscal = c.take<float>(4);
    w->centroid = c.take<float>(4);
    w->saved = c.take<unsigned char>((size_t)slots * w->slot_bytes);
    w->dy = c.take<unsigned char>((size_t)slots * w->slot_bytes);
    w->masks = c.take<uint32_t>((size_t)slots * w->total * 8);
    w->aux_vf = c.take<float>(w->total * 40);
    w->aux_rn = c.take<float>(w->total * 40);
    w->dz_vec = c.take<float>(w->total * 4);
    w->dz_rgb = c.take<float>((p->sparse_colours ? w->total : w->m) * 4);      // (workspace-indexed: region 2 sits behind the other rows)
    const int64_t s_vf = vfn_net_weight_grads_scratch_bytes(VFN_NET_VF, vf_geom, w->total);
    const int64_t s_rn = vfn_net_weight_grads_scratch_bytes(VFN_NET_RENDER, rn_geom, w->m);
    if (s_vf < 0 || s_rn < 0) return VFN_ERR_UNSUPPORTED;
    w->scratch_vf = c.take<unsigned char>((size_t)s_vf);
    w->scratch_rn = c.take<unsigned char>((size_t)s_rn);
    const int64_t s_vf2 = p->sparse_colours ? vfn_net_weight_grads_scratch_bytes(VFN_NET_VF, vf_geom, w->cap) : 0;
    if (s_vf2 < 0) return VFN_ERR_UNSUPPORTED;
    w->scratch_vf2 = c.take<unsigned char>((size_t)s_vf2);
    w->bytes = c.off;
    return VFN_OK;
}

struct PrepArgs {
    const float *beta, *mean, *scale;
    float *scal, *dscal, *centroid;
    float cx, cy, cz;
    float *pad_pts, *pad_gt, *pad_dsup;       // rows [m_sup, m_sup_pad) of the supervision batch and of its upstream gradient
    int pad_floats;
    float *zero_a, *zero_b;                   // session form: the whole supervision-point and upstream-gradient regions (zero_floats each)
    long long zero_floats;
    // buffers a later launch of the step expects cleared (the colours outside the selection, the zero rows region 2's chain reads as its
    // vector gradient, the flat gradient in the whole-step form): cleared HERE instead of by three fill launches on the step's critical path
    float* clear[3];
    long long clear_floats[3];
};

__global__ void vfn_train_prep_kernel(const PrepArgs a) {
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {
        if (t == 0) {
            a.scal[0] = *a.beta; a.scal[1] = *a.mean; a.scal[2] = *a.scale;
            a.dscal[0] = 0.f; a.dscal[1] = 0.f; a.dscal[2] = 0.f;
            a.centroid[0] = a.cx; a.centroid[1] = a.cy; a.centroid[2] = a.cz;
        }
        if (t < a.pad_floats) { a.pad_pts[t] = 0.f; a.pad_gt[t] = 0.f; a.pad_dsup[t] = 0.f; }
    }
    const long long first = (long long)blockIdx.x * blockDim.x + t, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = first; i < a.zero_floats; i += stride) {
        a.zero_a[i] = 0.f;
        a.zero_b[i] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float* base = a.clear[r];
        const long long n = a.clear_floats[r];
        if (n <= 0) continue;
        long long head = (long long)((16u - (unsigned)(reinterpret_cast<uintptr_t>(base) & 15u)) & 15u) / 4;      // floats in front of the first 16-byte line
        head = head < n ? head : n;
        float4* p4 = reinterpret_cast<float4*>(base + head);
        const long long n4 = (n - head) / 4;
        for (long long i = first; i < n4; i += stride) p4[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (first < head) base[first] = 0.f;
        for (long long i = head + 4 * n4 + first; i < n; i += stride) base[i] = 0.f;
    }
}

__global__ void vfn_train_scalar_grads_kernel(const float* dscal, float* g_beta, float* g_mean, float* g_scale, const int32_t* k_dev, float m,
                                              float* out_counts) {
    if (threadIdx.x == 0) {
        *g_beta += dscal[0]; *g_mean += dscal[1]; *g_scale += dscal[2];
        if (out_counts) { out_counts[0] = k_dev ? (float)*k_dev : m; out_counts[1] = m; }
    }
}

// one side stream (+ fork / join events) per host thread and device, made on first use: the supervision batch's forward and
// chain are independent of the fine pass until the loss / the weight gradients, and at the reference's batch size they are
// 0.8-round launches that leave the chip mostly idle when they run alone.  (A lowest-priority side stream was measured in round 5 — the
// supervision forward forked at the start of the step and left to fill whatever the main stream leaves idle: no gain over the explicit
// placement behind the fine pass, 1-2 % worse at 1 024 rays.)
struct Side { hipStream_t s; hipEvent_t fork, join, after_fine; int dev; const void* armed_ws; const void* gated_ws; };
Side* side_stream() {
    static thread_local Side side = {nullptr, nullptr, nullptr, nullptr, -1, nullptr, nullptr};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (side.dev != dev) {
        if (side.dev >= 0) {
            (void)hipStreamDestroy(side.s); (void)hipEventDestroy(side.fork); (void)hipEventDestroy(side.join); (void)hipEventDestroy(side.after_fine);
            side.dev = -1;
        }
        if (hipStreamCreateWithFlags(&side.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&side.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&side.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&side.after_fine, hipEventDisableTiming) != hipSuccess) return nullptr;
        side.armed_ws = side.gated_ws = nullptr;
        side.dev = dev;
    }
    return &side;
}
int fork_to(Side* sd, hipStream_t from) {
    if (hipEventRecord(sd->fork, from) != hipSuccess || hipStreamWaitEvent(sd->s, sd->fork, 0) != hipSuccess) {
        vfn_set_error("vfn_train_step: could not fork the side stream");
        return VFN_ERR_LAUNCH;
    }
    return VFN_OK;
}
int join_into(Side* sd, hipStream_t into) {
    if (hipEventRecord(sd->join, sd->s) != hipSuccess || hipStreamWaitEvent(into, sd->join, 0) != hipSuccess) {
        vfn_set_error("vfn_train_step: could not join the side stream");
        return VFN_ERR_LAUNCH;
    }
    return VFN_OK;
}

#define STEP(call)                       \
    do {                                 \
        rc = (call);                     \
        if (rc != VFN_OK) return rc;     \
    } while (0)


// ---- the parts of a step (shared by vfn_train_step's whole-step phases and the session form) ------------------------------------------------
struct StepCtx {
    const vfn_train_step_params* p;
    const vfn_train_step_io* io;
    Ws w;
    hipStream_t s, ss;            // the caller's stream; the side stream (== s without one)
    Side* sd;
    int n, sc, nf, st;
    bool sparse;
    int sup_mode;                 // what step_render does once the fine pass's forward is launched: 0 nothing, 1 fork the side stream and issue the
                                  // supervision batch there (whole-step form), 2 record Side::after_fine (session form: gates the caller's forwards)
    float* saved_f;
    vfn_density_params dp;        // n_samples = S_t after the render part
};

int step_open(StepCtx& c, const vfn_train_step_params* p, const vfn_train_step_io* io, void* stream, bool want_side) {
    int rc;
    c.p = p; c.io = io; c.s = (hipStream_t)stream;
    c.n = p->render.n_rays; c.sc = p->render.n_coarse; c.nf = p->render.n_fine; c.st = c.sc + c.nf;
    STEP(carve(io->workspace, p, io->vf_geom, io->rn_geom, &c.w));
    // a LAUNCH addresses its part of a slot with 32-bit offsets (< 2^21 points); the workspace itself may hold more (round 5)
    VFN_REQUIRE(c.w.m + c.w.m_sup_pad < (1ll << 21) && c.w.total < (1ll << 26),
                "vfn_train_step: at most 2097151 points per launch (%lld samples + %lld supervision rows) and 67108863 per workspace (%lld)", c.w.m,
                c.w.m_sup_pad, c.w.total);
    c.sd = (want_side && p->render.streams >= 2) ? side_stream() : nullptr;
    c.ss = c.sd ? c.sd->s : c.s;
    c.sparse = p->sparse_colours != 0;
    c.sup_mode = 0;
    c.saved_f = reinterpret_cast<float*>(c.w.saved);
    c.dp = p->render.density;
    c.dp.n_rays = c.n; c.dp.n_samples = c.st;
    return VFN_OK;
}

int step_prep(StepCtx& c, bool session) {
    const vfn_train_step_params* p = c.p; const vfn_train_step_io* io = c.io; Ws& w = c.w;
    int rc;
    const int pad_floats = session ? 0 : (int)((w.m_sup_pad - w.m_sup) * 3);
    // session form: the caller appends batches of any size — every row it does not fill is a point at the origin with a zero upstream gradient
    const long long zero_floats = session ? w.m_sup_pad * 3 : 0;
    PrepArgs pa{io->beta, io->mean, io->scale, w.scal, w.dscal, w.centroid, p->sup_centroid[0], p->sup_centroid[1], p->sup_centroid[2],
                w.sup_pts + w.m_sup * 3, w.sup_gt + w.m_sup * 3, w.d_sup + w.m_sup * 3, pad_floats, w.sup_pts, w.d_sup, zero_floats,
                {nullptr, nullptr, nullptr}, {0, 0, 0}};
    long long most = zero_floats;
    int nc = 0;
    auto clear = [&](float* ptr, long long floats) {
        if (!ptr || floats <= 0) return;
        pa.clear[nc] = ptr; pa.clear_floats[nc] = floats; ++nc;
        most = floats / 4 > most ? floats / 4 : most;
    };
    if (c.sparse) { clear(io->colors, w.m * 3); clear(w.zero3, w.cap * 3); }
    if (!session) clear(io->flat_grad, io->n_flat);
    const unsigned blocks = (unsigned)(most > 0 ? (most + 1023) / 1024 : 1);
    hipLaunchKernelGGL(vfn_train_prep_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, c.s, pa);
    STEP(vfn_check_launch("vfn_train_step (prep)"));
    return VFN_OK;
}

// the supervision batch of the whole-step form: both shells sampled here, one vector-only saving forward (on c.ss)
int step_supervision(StepCtx& c) {
    const vfn_train_step_params* p = c.p; const vfn_train_step_io* io = c.io; Ws& w = c.w;
    int rc;
    long long row = 0;
    if (p->border && p->n_sup > 0) {
        STEP(vfn_sample_sphere_shell(p->n_sup, p->border_r_min, p->border_r_max, w.centroid, 1, io->sup_u_border, p->sup_seed, p->sup_offset,
                                     w.sup_pts, w.sup_gt, c.ss));
        row += p->n_sup;
    }
    if (p->center && p->n_sup > 0) {
        STEP(vfn_sample_sphere_shell(p->n_sup, 0.0f, p->sup_radius, w.centroid, 0, io->sup_u_center, p->sup_seed,
                                     p->sup_offset + (io->sup_u_border || !p->border ? 0 : (uint64_t)p->n_sup), w.sup_pts + row * 3,
                                     w.sup_gt + row * 3, c.ss));
        row += p->n_sup;
    }
    if (w.m_sup_pad > 0)
        STEP(vfn_vf_mlp16_fwd_train_at(io->vf_geom, io->vf_packed16, w.sup_pts, w.m_sup_pad, 0, w.sup_pred, c.saved_f, w.aux_vf, w.masks,
                                       p->save_flags, w.m, w.total, c.ss));
    return VFN_OK;
}

// render() under autograd: one vector-field evaluation per distinct sample (backward.StoredFinePass)
int step_render(StepCtx& c) {
    const vfn_train_step_params* p = c.p; const vfn_train_step_io* io = c.io; Ws& w = c.w;
    const vfn_render_params& r = p->render;
    const int n = c.n, sc = c.sc, nf = c.nf, st = c.st;
    hipStream_t s = c.s;
    float* saved_f = c.saved_f;
    int rc;
    const int gen_c = r.perturb_coarse && !io->u_coarse, gen_f = r.perturb_fine && !io->u_fine, gen_a = !io->u_add;
    const long long base_f = gen_c ? (long long)n * sc : 0, base_a = base_f + (gen_f ? (long long)n * nf : 0);
    vfn_raygen_params rq = {n, sc, r.pose_is_quat, r.near_coarse, r.far_coarse};
    STEP(vfn_internal_raygen(&rq, io->uv, io->pose, io->intrinsics, io->intrinsics, io->t_vals, io->far_coarse_per_ray,
                             r.perturb_coarse ? io->u_coarse : nullptr, gen_c, 0, r.seed, r.offset, w.directions, io->ray_dirs, w.cam_loc, w.z_c,
                             w.pts_c, s));
    const bool sparse = c.sparse;
    if (sparse)      // region 1: the vector-field net alone (vector head, no feature block) on every sample
        STEP(vfn_vf_mlp16_fwd_train_at(io->vf_geom, io->vf_packed16, w.pts_c, w.m_c, 0, w.normals_s, saved_f, w.aux_vf, w.masks, p->save_flags, 0,
                                       w.total, s));
    else
        STEP(vfn_vf_render_fused16_fwd_train_at(io->vf_geom, io->vf_packed16, io->rn_geom, io->rn_packed16, w.pts_c, io->ray_dirs, w.m_c, sc,
                                                w.normals_s, w.colors_s, saved_f, w.aux_vf, w.aux_rn, w.masks, p->save_flags, 0, w.total,
                                                p->forward_products, s));
    vfn_density_params dp = r.density;
    dp.n_rays = n; dp.n_samples = sc;
    vfn_fine_params fp = {n, sc, nf, r.near_fine, r.far_fine, r.fine_range, r.window_step, r.span};
    STEP(vfn_internal_density_fine(&dp, w.normals_s, io->ray_dirs, w.z_c, w.scal, &fp, w.directions, w.cam_loc, io->far_fine_per_ray,
                                   r.perturb_fine ? io->u_fine : nullptr, io->u_add, gen_f, gen_a, base_f, base_a, r.seed, r.offset, io->z_vals,
                                   io->points, w.src, w.new_pts, w.dst, w.m_c, s));
    dp.n_samples = st;
    if (sparse) {
        STEP(vfn_vf_mlp16_fwd_train_at(io->vf_geom, io->vf_packed16, w.new_pts, w.m - w.m_c, 0, w.normals_s + w.m_c * 3, saved_f, w.aux_vf, w.masks,
                                       p->save_flags, w.m_c, w.total, s));
        // The supervision batch's forward goes HERE (round 5): what follows the fine pass's forward on this stream — the weights, the selection,
        // region 2's forward of under one round of workgroups, the composite, the loss — leaves the chip mostly idle for ~0.3 ms at 4096 rays,
        // which is what that forward needs; beside the proposal pass (where rounds 4 put it) it shared a chip that was already full.
        if (c.sd && c.sup_mode == 1) { STEP(fork_to(c.sd, s)); STEP(step_supervision(c)); }
        if (c.sd && c.sup_mode == 2) {
            if (hipEventRecord(c.sd->after_fine, s) != hipSuccess) { vfn_set_error("vfn_train_step: could not record the fine pass's event"); return VFN_ERR_LAUNCH; }
            c.sd->gated_ws = io->workspace;
        }
        // normals to their sorted positions, weights (no colours yet)
        STEP(vfn_scatter_rows3(w.normals_s, nullptr, w.dst, w.m, io->normals, nullptr, s));
        // (sigma for the selection below goes through w.dc: the sorted colour gradients, written by the backward only)
        STEP(vfn_ray_density_weights(&dp, io->normals, io->ray_dirs, io->z_vals, w.scal, nullptr, w.dc, io->weights, nullptr, nullptr, nullptr, s));
        // the samples whose colour can reach an output or a gradient — w > 0, or w = 0 by an underflowed alpha alone (sigma > 0, T > 0,
        // delta > 0: d w / d sigma is not zero there) — compacted in ray order; their count stays on the device
        STEP(vfn_internal_select_positive(io->weights, w.dc, io->z_vals, n, st, io->points, io->ray_dirs, w.cnt, w.off, w.k_dev, w.sel_sorted, w.pts_sel,
                                          w.dirs_sel, s));
        // region 2: the fused saving forward (vector-field net + rendering net) on the selected samples only
        STEP(vfn_internal_fused16_fwd_train_at(io->vf_geom, io->vf_packed16, io->rn_geom, io->rn_packed16, w.pts_sel, w.dirs_sel, w.cap, w.k_dev, 1,
                                               w.normals_sel, w.colors_sel, saved_f, w.aux_vf, w.aux_rn, w.masks, p->save_flags, w.r2_first, w.total,
                                               p->forward_products, s));
        // colours: zero where w = 0 (they multiply a zero weight), the selected ones at their sorted positions; composite
        STEP(vfn_internal_rows3_by_index(w.colors_sel, w.sel_sorted, w.k_dev, w.cap, io->colors, 0, s));
        STEP(vfn_ray_density_weights(&dp, io->normals, io->ray_dirs, io->z_vals, w.scal, io->colors, nullptr, io->weights, nullptr, io->rgb, io->depth, s));
    } else {
        STEP(vfn_vf_render_fused16_fwd_train_at(io->vf_geom, io->vf_packed16, io->rn_geom, io->rn_packed16, w.new_pts, io->ray_dirs, w.m - w.m_c, nf,
                                                w.normals_s + w.m_c * 3, w.colors_s + w.m_c * 3, saved_f, w.aux_vf, w.aux_rn, w.masks, p->save_flags,
                                                w.m_c, w.total, p->forward_products, s));
        // every sample (proposal and new) moves from storage order to its sorted position on the way into the composite launch
        STEP(vfn_internal_composite_gather(&dp, io->normals, io->ray_dirs, io->z_vals, w.scal, io->colors, w.src, w.normals_s, w.colors_s, w.m,
                                           io->weights, io->rgb, io->depth, s));
    }
    return VFN_OK;
}

// backward: supervision chain, per-ray backward, fine chain, weight gradients, the density's scalar gradients; the loss's d normals is in
// w.dn (where the per-ray backward ADDS the density path's share), its d supervision predictions in w.d_sup
int step_backward(StepCtx& c, const float* d_rgb, const float* d_depth) {
    const vfn_train_step_params* p = c.p; const vfn_train_step_io* io = c.io; Ws& w = c.w;
    // supervision rows the chains and weight gradients walk: all of them in the whole-step form; in the session form those that forwards filled
    long long sup_rows = w.m_sup_pad;
    if (p->sup_rows_reserved > 0) {
        VFN_REQUIRE(p->sup_rows_used >= 0 && p->sup_rows_used % 32 == 0 && p->sup_rows_used <= w.m_sup_pad,
                    "vfn_train_step: sup_rows_used = %lld (a multiple of 32, at most %lld)", (long long)p->sup_rows_used, w.m_sup_pad);
        sup_rows = p->sup_rows_used;
    }
    hipStream_t s = c.s, ss = c.ss;
    Side* sd = c.sd;
    const bool sparse = c.sparse;
    float* saved_f = c.saved_f;
    const vfn_density_params& dp = c.dp;
    int rc;
    const float* feats = saved_f + (size_t)(w.vf_h - 1) * (w.slot_bytes / 4);           // the tanh'ed feature slot, row-major fp32
    const size_t rn_off = (size_t)w.vf_h * w.slot_bytes;
    if (sparse) {
        // per-ray backward on the sorted samples: d colours = w d rgb (zero wherever w is), d normals += the density path's share
        STEP(vfn_ray_density_weights_bwd(&dp, io->normals, io->ray_dirs, io->z_vals, w.scal, io->colors, d_rgb, d_depth, nullptr, w.dn, w.dc, w.dscal, s));
        STEP(vfn_scatter_rows3(w.dn, nullptr, w.src, w.m, w.dn_s, nullptr, s));       // row src[i] of region 1 is sorted sample i
        STEP(vfn_internal_rows3_by_index(w.dc, w.sel_sorted, w.k_dev, w.cap, w.dc_sel, 1, s));
        // Region 2's chain and the rendering net's weight gradients are small launches (a few percent of the samples: 0.6 rounds of
        // workgroups at 4096 rays) that touch nothing region 1's chain and weight gradients touch (other rows of the workspace, other
        // parameters' gradients, their own scratch): they run on the side stream beside them.
        const size_t r2_off = (size_t)(w.r2_first / 32) * GROUP_BYTES;
        if (sd) STEP(fork_to(sd, s));
        // the fused chain over region 2: d colours in, no gradient at the vector head (region 1 carries it)
        STEP(vfn_internal_bwd_chain_bf16_ws_at(io->vf_geom, io->vf_packed_bwd16, io->vf_head_w, io->rn_geom, io->rn_packed_bwd16, io->rn_head_w, feats,
                                               w.masks, w.dy, p->dy_flags, w.dc_sel, w.colors_sel, w.zero3, w.normals_sel, nullptr, 3, w.cap, w.k_dev,
                                               w.dz_rgb, w.dz_vec, w.r2_first, w.total, ss));
        // weight gradients: the rendering net over region 2; the vector-field net's hidden layers + head over region 1 and the
        // supervision rows, its hidden layers + feature block over region 2 (the head's gradient there is zero)
        STEP(vfn_internal_net_weight_grads_frag_part(VFN_NET_RENDER, io->rn_geom, io->rn_wgrad, w.saved + rn_off + r2_off, w.dy + rn_off + r2_off,
                                                     (int64_t)w.slot_bytes, p->dy_form, p->x_form, feats + w.r2_first * 256, w.aux_rn + w.r2_first * 40,
                                                     w.dz_rgb + w.r2_first * 4, w.cap, w.k_dev, VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES | VFN_WGRAD_HEAD, 1,
                                                     w.scratch_rn, ss, 1));
        // ... and the vector-field net's products over region 2 (hidden layers + feature block; the head's gradient there is zero) into a scratch
        // of their own; their un-fold ADDS to the tensors region 1's un-fold adds to, so it waits for the join below
        STEP(vfn_internal_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved + r2_off, w.dy + r2_off, (int64_t)w.slot_bytes,
                                                     p->dy_form, p->x_form, nullptr, w.aux_vf + w.r2_first * 40, w.dz_vec + w.r2_first * 4, w.cap,
                                                     w.k_dev, VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES, 1, w.scratch_vf2, ss, 1));
        STEP(vfn_internal_net_weight_grads_frag_part(VFN_NET_RENDER, io->rn_geom, io->rn_wgrad, w.saved + rn_off + r2_off, w.dy + rn_off + r2_off,
                                                     (int64_t)w.slot_bytes, p->dy_form, p->x_form, feats + w.r2_first * 256, w.aux_rn + w.r2_first * 40,
                                                     w.dz_rgb + w.r2_first * 4, w.cap, w.k_dev, VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES | VFN_WGRAD_HEAD, 1,
                                                     w.scratch_rn, ss, 2));
        // ONE vector-only chain over region 1 and the supervision rows (d normals | d supervision predictions)
        STEP(vfn_mlp_bwd_chain_bf16_ws_at(io->vf_geom, io->vf_packed_bwd16, io->vf_head_w, nullptr, nullptr, nullptr, feats, w.masks, w.dy, p->dy_flags,
                                          nullptr, nullptr, w.dn_s, w.normals_s, nullptr, 3, w.m + sup_rows, nullptr, w.dz_vec, 0, w.total, s));
        STEP(vfn_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved, w.dy, (int64_t)w.slot_bytes, p->dy_form, p->x_form, nullptr,
                                            w.aux_vf, w.dz_vec, w.m + sup_rows, VFN_WGRAD_LAYERS | VFN_WGRAD_HEAD, 1, w.scratch_vf, s));
        if (sd) STEP(join_into(sd, s));
        STEP(vfn_internal_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved + r2_off, w.dy + r2_off, (int64_t)w.slot_bytes,
                                                     p->dy_form, p->x_form, nullptr, w.aux_vf + w.r2_first * 40, w.dz_vec + w.r2_first * 4, w.cap,
                                                     w.k_dev, VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES, 1, w.scratch_vf2, s, 2));
    } else {
        if (sup_rows > 0) {
            // (beside the per-ray backward and the fine pass's chain when there is a side stream; joined in front of the weight gradients)
            if (sd) STEP(fork_to(sd, s));
            STEP(vfn_mlp_bwd_chain_bf16_ws_at(io->vf_geom, io->vf_packed_bwd16, io->vf_head_w, nullptr, nullptr, nullptr, feats, w.masks, w.dy,
                                              p->dy_flags, nullptr, nullptr, w.d_sup, w.sup_pred, nullptr, 3, sup_rows, nullptr, w.dz_vec, w.m,
                                              w.total, ss));
        }
        STEP(vfn_ray_density_weights_bwd(&dp, io->normals, io->ray_dirs, io->z_vals, w.scal, io->colors, d_rgb, d_depth, nullptr, w.dn, w.dc, w.dscal, s));
        // row src[i] of the workspace is sorted sample i: gradients to storage order
        STEP(vfn_scatter_rows3(w.dn, w.dc, w.src, w.m, w.dn_s, w.dc_s, s));
        STEP(vfn_mlp_bwd_chain_bf16_ws_at(io->vf_geom, io->vf_packed_bwd16, io->vf_head_w, io->rn_geom, io->rn_packed_bwd16, io->rn_head_w, feats,
                                          w.masks, w.dy, p->dy_flags, w.dc_s, w.colors_s, w.dn_s, w.normals_s, nullptr, 3, w.m, w.dz_rgb, w.dz_vec, 0,
                                          w.total, s));
        STEP(vfn_net_weight_grads_frag_part(VFN_NET_RENDER, io->rn_geom, io->rn_wgrad, w.saved + rn_off, w.dy + rn_off, (int64_t)w.slot_bytes,
                                            p->dy_form, p->x_form, feats, w.aux_rn, w.dz_rgb, w.m,
                                            VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES | VFN_WGRAD_HEAD, 1, w.scratch_rn, s));
        // vector-field net: hidden layers + head over ALL rows (fine pass + supervision), the feature block over the fine pass's rows
        if (sd && sup_rows > 0) STEP(join_into(sd, s));
        if (sup_rows > 0) {
            STEP(vfn_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved, w.dy, (int64_t)w.slot_bytes, p->dy_form, p->x_form,
                                                nullptr, w.aux_vf, w.dz_vec, w.m + sup_rows, VFN_WGRAD_LAYERS | VFN_WGRAD_HEAD, 1, w.scratch_vf, s));
            STEP(vfn_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved, w.dy, (int64_t)w.slot_bytes, p->dy_form, p->x_form,
                                                nullptr, w.aux_vf, w.dz_vec, w.m, VFN_WGRAD_FEATURES, 1, w.scratch_vf, s));
        } else {
            STEP(vfn_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved, w.dy, (int64_t)w.slot_bytes, p->dy_form, p->x_form,
                                                nullptr, w.aux_vf, w.dz_vec, w.m, VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES | VFN_WGRAD_HEAD, 1,
                                                w.scratch_vf, s));
        }
    }
    hipLaunchKernelGGL(vfn_train_scalar_grads_kernel, dim3(1), dim3(64), 0, s, w.dscal, io->g_beta, io->g_mean, io->g_scale,
                       sparse ? w.k_dev : nullptr, (float)w.m, io->out_counts);
    return vfn_check_launch("vfn_train_step (density scalar gradients)");
}

bool forward_pointers_ok(const vfn_train_step_io* io) {
    return io->workspace && io->vf_packed16 && io->rn_packed16 && io->beta && io->mean && io->scale && io->uv && io->pose && io->intrinsics &&
           io->t_vals && io->ray_dirs && io->z_vals && io->points && io->normals && io->colors && io->weights && io->rgb && io->depth;
}
bool backward_pointers_ok(const vfn_train_step_io* io) {
    return io->workspace && io->vf_packed_bwd16 && io->rn_packed_bwd16 && io->vf_wgrad && io->rn_wgrad && io->vf_head_w && io->rn_head_w && io->g_beta &&
           io->g_mean && io->g_scale && io->flat_grad && io->ray_dirs && io->z_vals && io->normals && io->colors;
}
}  // namespace

extern "C" int64_t vfn_train_step_workspace_bytes(const vfn_train_step_params* p, const vfn_net_geom* vf_geom, const vfn_net_geom* rn_geom) {
    if (!p || !vf_geom || !rn_geom || p->render.n_rays < 1 || p->render.n_coarse < 1 || p->render.n_fine < 2 || p->n_sup < 0) return VFN_ERR_INVALID;
    Ws w;
    const int rc = carve(nullptr, p, vf_geom, rn_geom, &w);
    return rc != VFN_OK ? rc : (int64_t)w.bytes;
}

extern "C" int vfn_train_step_workspace_layout(const vfn_train_step_params* p, const vfn_net_geom* vf_geom, const vfn_net_geom* rn_geom, int64_t* out,
                                               int32_t n_out) {
    VFN_REQUIRE(p && vf_geom && rn_geom && out && n_out >= VFN_TWS_COUNT, "vfn_train_step_workspace_layout: NULL argument or fewer than %d outputs",
                VFN_TWS_COUNT);
    VFN_REQUIRE(p->render.n_rays >= 1 && p->render.n_coarse >= 1 && p->render.n_fine >= 2 && p->n_sup >= 0, "vfn_train_step_workspace_layout: bad sizes");
    // carve() hands out NULL pointers for a NULL base: the offsets are taken against a dummy base that is never dereferenced
    unsigned char* base = reinterpret_cast<unsigned char*>(uintptr_t(1) << 40);
    Ws v;
    const int rc = carve(base, p, vf_geom, rn_geom, &v);
    if (rc != VFN_OK) return rc;
    auto off = [&](const void* q) { return (int64_t)(reinterpret_cast<const unsigned char*>(q) - base); };
    out[VFN_TWS_SUP_PTS] = off(v.sup_pts);
    out[VFN_TWS_SUP_GT] = off(v.sup_gt);
    out[VFN_TWS_SUP_PRED] = off(v.sup_pred);
    out[VFN_TWS_D_SUP] = off(v.d_sup);
    out[VFN_TWS_DN] = off(v.dn);
    out[VFN_TWS_SUP_ROWS] = v.m_sup_pad;
    out[VFN_TWS_TOTAL_ROWS] = v.total;
    out[VFN_TWS_BYTES] = (int64_t)v.bytes;
    return VFN_OK;
}

// Which of the calling thread's side-stream state belongs to the step that is open on `workspace` (the session form's later calls)
static Side* armed_side(const vfn_train_step_params* p, const void* workspace) {
    if (p->render.streams < 2) return nullptr;
    Side* sd = side_stream();
    return (sd && sd->armed_ws == workspace) ? sd : nullptr;
}

__global__ void vfn_train_set3_kernel(float* dst, float x, float y, float z) {
    if (threadIdx.x == 0) { dst[0] = x; dst[1] = y; dst[2] = z; }
}

extern "C" int vfn_train_step_supervision_points(const vfn_train_step_params* p, const vfn_train_step_io* io, int32_t inward, float r_min, float r_max,
                                                 float cx, float cy, float cz, const float* centroid_dev, int64_t row0, int64_t count, const float* u,
                                                 uint64_t seed, uint64_t offset, void* stream) {
    VFN_REQUIRE(p && io && io->vf_geom && io->rn_geom && io->workspace, "vfn_train_step_supervision_points: NULL argument");
    Ws w;
    int rc;
    STEP(carve(io->workspace, p, io->vf_geom, io->rn_geom, &w));
    VFN_REQUIRE(row0 >= 0 && count >= 0 && row0 + count <= w.m_sup_pad, "vfn_train_step_supervision_points: rows [%lld, %lld) of %lld", (long long)row0,
                (long long)(row0 + count), w.m_sup_pad);
    if (count == 0) return VFN_OK;
    hipStream_t s = (hipStream_t)stream;
    // operands the caller's stream may have produced after the step's prep (a device centroid, supplied draws): on the caller's stream
    Side* sd = (centroid_dev || u) ? nullptr : armed_side(p, io->workspace);
    hipStream_t ss = sd ? sd->s : s;
    const float* c = centroid_dev;
    if (!c) {
        float* slot = w.centroid + 4;           // (the carve rounds the centroid block up to 256 bytes: slot 1 of it)
        hipLaunchKernelGGL(vfn_train_set3_kernel, dim3(1), dim3(64), 0, ss, slot, cx, cy, cz);
        STEP(vfn_check_launch("vfn_train_step_supervision_points (centre)"));
        c = slot;
    }
    STEP(vfn_sample_sphere_shell(count, r_min, r_max, c, inward ? 1 : 0, u, seed, offset, w.sup_pts + row0 * 3, w.sup_gt + row0 * 3, ss));
    if (sd) { STEP(join_into(sd, s)); return 1; }
    return VFN_OK;
}

extern "C" int vfn_train_step_supervision_forward(const vfn_train_step_params* p, const vfn_train_step_io* io, int64_t row0, int64_t count, int32_t on_side,
                                                  void* stream) {
    VFN_REQUIRE(p && io && io->vf_geom && io->rn_geom && io->workspace && io->vf_packed16, "vfn_train_step_supervision_forward: NULL argument");
    VFN_REQUIRE(p->save_flags & 2, "vfn_train_step_supervision_forward: the fragment-ordered workspace only (save_flags bit 1)");
    VfnReportScope report(p->render.status_word, nullptr, 0);
    Ws w;
    int rc;
    STEP(carve(io->workspace, p, io->vf_geom, io->rn_geom, &w));
    const long long rows = pad32(count);
    VFN_REQUIRE(row0 >= 0 && row0 % 32 == 0 && count >= 0 && row0 + rows <= w.m_sup_pad,
                "vfn_train_step_supervision_forward: rows [%lld, %lld) of %lld (row0 must be a multiple of 32)", (long long)row0, (long long)(row0 + rows),
                w.m_sup_pad);
    if (count == 0) return VFN_OK;
    hipStream_t s = (hipStream_t)stream;
    Side* sd = on_side ? armed_side(p, io->workspace) : nullptr;
    hipStream_t ss = sd ? sd->s : s;
    // (behind the fine pass's forward, where the chip has room for it: see step_render)
    if (sd && sd->gated_ws == io->workspace && hipStreamWaitEvent(sd->s, sd->after_fine, 0) != hipSuccess) {
        vfn_set_error("vfn_train_step_supervision_forward: could not wait for the fine pass");
        return VFN_ERR_LAUNCH;
    }
    STEP(vfn_vf_mlp16_fwd_train_at(io->vf_geom, io->vf_packed16, w.sup_pts + row0 * 3, rows, 0, w.sup_pred + row0 * 3, reinterpret_cast<float*>(w.saved),
                                   w.aux_vf, w.masks, p->save_flags, w.m + row0, w.total, ss));
    if (sd) STEP(join_into(sd, s));
    return VFN_OK;
}

extern "C" int vfn_train_step_supervision_backward(const vfn_train_step_params* p, const vfn_train_step_io* io, int64_t row0, int64_t count, void* stream) {
    VFN_REQUIRE(p && io && io->vf_geom && io->rn_geom && io->workspace && io->vf_packed_bwd16 && io->vf_wgrad && io->vf_head_w && io->flat_grad,
                "vfn_train_step_supervision_backward: NULL argument");
    Ws w;
    int rc;
    STEP(carve(io->workspace, p, io->vf_geom, io->rn_geom, &w));
    const long long rows = pad32(count);
    VFN_REQUIRE(row0 >= 0 && row0 % 32 == 0 && count >= 0 && row0 + rows <= w.m_sup_pad,
                "vfn_train_step_supervision_backward: rows [%lld, %lld) of %lld (row0 must be a multiple of 32)", (long long)row0, (long long)(row0 + rows),
                w.m_sup_pad);
    if (count == 0) return VFN_OK;
    hipStream_t s = (hipStream_t)stream;
    const float* feats = reinterpret_cast<float*>(w.saved) + (size_t)(w.vf_h - 1) * (w.slot_bytes / 4);
    const long long first = w.m + row0;
    const size_t g_off = (size_t)(first / 32) * GROUP_BYTES;
    STEP(vfn_mlp_bwd_chain_bf16_ws_at(io->vf_geom, io->vf_packed_bwd16, io->vf_head_w, nullptr, nullptr, nullptr, feats, w.masks, w.dy, p->dy_flags, nullptr,
                                      nullptr, w.d_sup + row0 * 3, w.sup_pred + row0 * 3, nullptr, 3, rows, nullptr, w.dz_vec, first, w.total, s));
    STEP(vfn_net_weight_grads_frag_part(VFN_NET_VF, io->vf_geom, io->vf_wgrad, w.saved + g_off, w.dy + g_off, (int64_t)w.slot_bytes, p->dy_form, p->x_form,
                                        nullptr, w.aux_vf + first * 40, w.dz_vec + first * 4, rows, VFN_WGRAD_LAYERS | VFN_WGRAD_HEAD, 1, w.scratch_vf, s));
    if (hipMemsetAsync(w.d_sup + row0 * 3, 0, (size_t)rows * 3 * sizeof(float), s) != hipSuccess) {
        vfn_set_error("vfn_train_step_supervision_backward: could not clear the upstream rows");
        return VFN_ERR_LAUNCH;
    }
    return VFN_OK;
}

extern "C" int vfn_train_step(const vfn_train_step_params* p, const vfn_train_step_io* io, void* stream) {
    VFN_REQUIRE(p && io && io->vf_geom && io->rn_geom, "vfn_train_step: NULL argument");
    const vfn_render_params& r = p->render;
    VFN_REQUIRE(r.n_rays > 0 && r.n_coarse >= 1 && r.n_fine >= 2, "vfn_train_step: bad sizes (n_rays=%d, n_coarse=%d, n_fine=%d)", r.n_rays,
                r.n_coarse, r.n_fine);
    const int all = VFN_TRAIN_FORWARD_BACKWARD | VFN_TRAIN_OPTIMIZER | VFN_TRAIN_RENDER | VFN_TRAIN_BACKWARD | VFN_TRAIN_CLIP | VFN_TRAIN_ADAM;
    VFN_REQUIRE((p->phases & all) && !(p->phases & ~all), "vfn_train_step: phases = %d selects nothing", p->phases);
    VFN_REQUIRE(!((p->phases & VFN_TRAIN_FORWARD_BACKWARD) && (p->phases & (VFN_TRAIN_RENDER | VFN_TRAIN_BACKWARD))),
                "vfn_train_step: VFN_TRAIN_FORWARD_BACKWARD already contains the render and the backward part (phases = %d)", p->phases);
    const int n = r.n_rays, sc = r.n_coarse, nf = r.n_fine, st = sc + nf;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    VfnReportScope report(r.status_word, nullptr, 0);

    if (p->phases & (VFN_TRAIN_FORWARD_BACKWARD | VFN_TRAIN_RENDER | VFN_TRAIN_BACKWARD)) {
        VFN_REQUIRE(io->workspace, "vfn_train_step: NULL workspace");
        VFN_REQUIRE(((long long)n * sc) % 32 == 0 && ((long long)n * st) % 32 == 0,
                    "vfn_train_step: the proposal samples and all samples must be whole groups of 32 points (%d rays x %d + %d)", n, sc, nf);
        VFN_REQUIRE(p->save_flags & 2, "vfn_train_step: the fragment-ordered workspace only (save_flags bit 1)");
    }

    if (p->phases & VFN_TRAIN_FORWARD_BACKWARD) {
        VFN_REQUIRE(forward_pointers_ok(io) && backward_pointers_ok(io) && io->rgb_gt && io->out_terms, "vfn_train_step: NULL argument");
        VFN_REQUIRE(!p->loss.has_depth || io->depth_gt, "vfn_train_step: has_depth without depth_gt");
        VFN_REQUIRE(p->sup_rows_reserved == 0, "vfn_train_step: sup_rows_reserved belongs to the session form (VFN_TRAIN_RENDER)");
        StepCtx c;
        STEP(step_open(c, p, io, stream, true));
        Ws& w = c.w;
        if (w.m_sup_pad == 0) { c.sd = nullptr; c.ss = s; }
        STEP(step_prep(c, false));
        // the supervision batch runs on a side stream beside the render (render.streams >= 2; joined in front of the loss): its points
        // depend on nothing but the prep launch
        if (c.sd && c.sparse && p->render.streams != 3) c.sup_mode = 1;      // (streams = 3: beside the proposal pass as in round 4, for A/B)
        else if (c.sd) { STEP(fork_to(c.sd, s)); STEP(step_supervision(c)); }
        STEP(step_render(c));
        // ---- supervision points and their vector-only forward (train.py:186-216) -----------------------------------------------------
        if (c.sd) STEP(join_into(c.sd, s));
        else STEP(step_supervision(c));

        // ---- VFLoss forward / backward (vf_loss.py:34-87; the centre-ball rows of functions.py:137-157 inside the launches) -------------
        vfn_loss_params lp = p->loss;
        lp.n_rays = n; lp.n_normals = w.m;
        lp.n_sup[0] = w.m_sup; lp.n_sup[1] = 0; lp.n_sup[2] = 0;
        const float* sup_pred[3] = {w.m_sup ? w.sup_pred : nullptr, nullptr, nullptr};
        const float* sup_gt[3] = {w.m_sup ? w.sup_gt : nullptr, nullptr, nullptr};
        float* d_sup[3] = {w.m_sup ? w.d_sup : nullptr, nullptr, nullptr};
        const float* loss_points = lp.ray_center ? io->points : nullptr;
        STEP(vfn_vf_loss_fwd(&lp, io->rgb, io->rgb_gt, lp.has_depth ? io->depth : nullptr, lp.has_depth ? io->depth_gt : nullptr, io->normals,
                             loss_points, sup_pred, sup_gt, w.loss_ws, io->out_terms, s));
        // d normals of the loss lands in `dn`, where the per-ray backward ADDS the density path's share
        STEP(vfn_vf_loss_bwd(&lp, io->rgb, io->rgb_gt, lp.has_depth ? io->depth : nullptr, lp.has_depth ? io->depth_gt : nullptr, io->normals,
                             loss_points, sup_pred, sup_gt, w.loss_ws, nullptr, w.d_rgb, lp.has_depth ? w.d_depth : nullptr, w.dn, d_sup, s));
        STEP(step_backward(c, w.d_rgb, lp.has_depth ? w.d_depth : nullptr));
    }

    if (p->phases & VFN_TRAIN_RENDER) {
        VFN_REQUIRE(forward_pointers_ok(io), "vfn_train_step: NULL argument (render part)");
        StepCtx c;
        STEP(step_open(c, p, io, stream, true));
        STEP(step_prep(c, true));
        if (c.sd) {
            // the supervision calls that follow (vfn_train_step_supervision_points / _forward) run on the side stream, after this step's prep
            STEP(fork_to(c.sd, s));
            c.sd->armed_ws = io->workspace;
            c.sd->gated_ws = nullptr;
            if (c.sparse && p->render.streams != 3) c.sup_mode = 2;
        }
        STEP(step_render(c));
    }

    if (p->phases & VFN_TRAIN_BACKWARD) {
        VFN_REQUIRE(backward_pointers_ok(io) && io->d_rgb_in && io->d_normals_in, "vfn_train_step: NULL argument (backward part)");
        StepCtx c;
        STEP(step_open(c, p, io, stream, true));
        Ws& w = c.w;
        if (io->d_normals_in != w.dn &&
            hipMemcpyAsync(w.dn, io->d_normals_in, (size_t)w.m * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            vfn_set_error("vfn_train_step: could not copy the upstream gradient of the normals");
            return VFN_ERR_LAUNCH;
        }
        if (hipMemsetAsync(w.dscal, 0, 4 * sizeof(float), s) != hipSuccess) {
            vfn_set_error("vfn_train_step: could not clear the scalar gradients");
            return VFN_ERR_LAUNCH;
        }
        STEP(step_backward(c, io->d_rgb_in, io->d_depth_in));
        if (c.sd) c.sd->armed_ws = c.sd->gated_ws = nullptr;
    }

    // VFN_TRAIN_OPTIMIZER = VFN_TRAIN_CLIP then VFN_TRAIN_ADAM (the session form's caller makes them as two calls: clip_grad_norm_, optimizer.step)
    const bool clip = (p->phases & (VFN_TRAIN_OPTIMIZER | VFN_TRAIN_CLIP)) != 0, adam = (p->phases & (VFN_TRAIN_OPTIMIZER | VFN_TRAIN_ADAM)) != 0;
    if (clip || adam)
        VFN_REQUIRE(io->flat_grad && io->n_flat > 0 && p->n_regions >= 1 && p->n_regions <= 4, "vfn_train_step: optimizer phase without its buffers");
    if (clip) {
        VFN_REQUIRE(io->clip_workspace && io->out_norm, "vfn_train_step: clip without its workspace / output");
        STEP(vfn_flat_clip_grad_norm(io->flat_grad, io->n_flat, p->n_regions, p->starts, p->ends, p->mults, p->max_norm, io->clip_workspace,
                                     io->out_norm, s));
    }
    if (adam) {
        VFN_REQUIRE(io->flat_param && io->exp_avg && io->exp_avg_sq, "vfn_train_step: Adam without its buffers");
        STEP(vfn_flat_adam_step(io->flat_param, io->flat_grad, io->exp_avg, io->exp_avg_sq, io->n_flat, p->n_regions, p->starts, p->ends, p->mults,
                                p->step_size, p->bc2_sqrt, p->beta1, p->beta2, p->eps, p->weight_decay, s));
        if (p->repack) {
            VFN_REQUIRE(io->vf_layers && io->rn_layers && io->vf_packed16 && io->rn_packed16 && io->vf_packed_bwd16 && io->rn_packed_bwd16,
                        "vfn_train_step: repack without the layer tables / packs");
            // four small launches (+ two fills) that depend on Adam only: the rendering net's on the side stream beside the vector-field net's
            Side* sd = p->render.streams >= 2 ? side_stream() : nullptr;
            hipStream_t ss = sd ? sd->s : s;
            if (sd) STEP(fork_to(sd, s));
            STEP(vfn_pack16_weights(VFN_NET_RENDER, io->rn_geom, io->rn_layers, io->rn_packed16, ss));
            STEP(vfn_pack_weights_bwd16_mode(VFN_NET_RENDER, io->rn_geom, io->rn_layers, p->forward_products == 1 ? 1 : 0, io->rn_packed_bwd16, ss));
            STEP(vfn_pack16_weights(VFN_NET_VF, io->vf_geom, io->vf_layers, io->vf_packed16, s));
            STEP(vfn_pack_weights_bwd16_mode(VFN_NET_VF, io->vf_geom, io->vf_layers, p->forward_products == 1 ? 1 : 0, io->vf_packed_bwd16, s));
            if (sd) STEP(join_into(sd, s));
        }
    }
    return VFN_OK;
}
