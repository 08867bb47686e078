// vfn_render.hip — VectorFieldNerf.render (models/nerf/vector_field_nerf.py:216-338), gradient-free, as ONE entry point.
//
// The facade used to issue the launches of an inference render() one ctypes call at a time, with a torch allocation per
// intermediate: ~0.5 ms of host time per call, which is what a 1 024-ray chunk takes on the device — the evaluator's chunk loop
// (evaluation/methods.py:520-545) was host-bound.  vfn_render_fwd is the same launch sequence issued from C on one stream out of
// one caller-supplied workspace (SURVEY.md section 8b lists this entry point):
//   rays + proposal samples (draws not supplied by the caller: Philox, in place)  utils/rendering.py:12-60, ray_sampler.py:49-80,113-142
//   fused VF + rendering net on the S_c proposal samples, in generation order    vector_field_nerf.py:252-256 (+ :315 for these samples)
//   density -> weights -> argmax on the proposal pass, then, in the same launch,  :263-272, ray_sampler.py:277
//   the range fine sampler with provenance (where every stored sample lands)     ray_sampler.py:264-302
//   fused VF + rendering net on the N_f NEW samples, outputs scattered           vector_field_nerf.py:294-297,315-318
//   density -> weights -> composite; on its way in the launch moves the proposal  :308-323
//   samples' normals / colours to their sorted positions
// i.e. the f16x3 pipeline with one VF evaluation per distinct sample in FIVE launches; the per-ray ones are
// the kernels of the stand-alone entry points (vfn_raygen_uniform, vfn_ray_density_weights, vfn_range_fine_sample_indexed,
// vfn_scatter_rows3, vfn_fill_uniform) compiled into three launches, so every value is what those entry points produce, bit for
// bit (tests/test_hip_f16x3.py::test_one_call_render_equals_the_launch_by_launch_path).
#include <string.h>
#include "vfn_common.h"

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

struct Ws {
    float *directions, *cam_loc, *z_c, *pts_c, *normals_c, *colors_c, *new_pts, *u;
    int32_t *dst, *src;
    int64_t* imax;
    // sparse colours: the samples with w > 0, compacted (csrc/vfn_rays.hip, vfn_internal_select_positive)
    int32_t *cnt, *off, *k_dev, *sel_sorted;
    float *pts_sel, *dirs_sel;
    size_t bytes;
};

Ws carve(void* workspace, const vfn_render_params* p) {
    const size_t n = (size_t)p->n_rays, sc = (size_t)p->n_coarse, nf = (size_t)p->n_fine;
    Carve c{static_cast<unsigned char*>(workspace), 0};
    Ws w;
    w.directions = c.take<float>(n * 3);
    w.cam_loc = c.take<float>(n * 3);
    w.z_c = c.take<float>(n * sc);
    w.pts_c = c.take<float>(n * sc * 3);
    w.normals_c = c.take<float>(n * (p->sparse_colours ? sc + nf : sc) * 3);      // (sparse colours: [proposal | new] in storage order)
    w.colors_c = c.take<float>(n * sc * 3);
    w.new_pts = c.take<float>(n * nf * 3);
    w.dst = c.take<int32_t>(n * (sc + nf));
    w.src = c.take<int32_t>(n * (sc + nf));
    w.u = c.take<float>(p->separate_launches ? n * (sc + 2 * nf) + 4 : 0);      // eight-launch plan only
    w.imax = c.take<int64_t>(p->separate_launches ? n : 0);
    const size_t cap = p->sparse_colours ? n * (sc + nf) : 0;
    w.cnt = c.take<int32_t>(p->sparse_colours ? n : 0);
    w.off = c.take<int32_t>(p->sparse_colours ? n : 0);
    w.k_dev = c.take<int32_t>(p->sparse_colours ? 4 : 0);
    w.sel_sorted = c.take<int32_t>(cap);
    w.pts_sel = c.take<float>(cap * 3);
    w.dirs_sel = c.take<float>(cap * 3);
    w.bytes = c.off;
    return w;
}

// side streams (+ one fork event and a join event each) per host thread and device, made on first use
constexpr int MAX_SIDE = 3;
struct Side { hipStream_t s[MAX_SIDE]; hipEvent_t fork, join[MAX_SIDE]; int dev; };
Side* side_streams() {
    static thread_local Side side = {{nullptr, nullptr, nullptr}, nullptr, {nullptr, nullptr, nullptr}, -1};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (side.dev != dev) {
        if (side.dev >= 0) {
            for (int i = 0; i < MAX_SIDE; ++i) { (void)hipStreamDestroy(side.s[i]); (void)hipEventDestroy(side.join[i]); }
            (void)hipEventDestroy(side.fork);
            side.dev = -1;
        }
        if (hipEventCreateWithFlags(&side.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        for (int i = 0; i < MAX_SIDE; ++i)
            if (hipStreamCreateWithFlags(&side.s[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&side.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        side.dev = dev;
    }
    return &side;
}

}  // namespace

extern "C" int64_t vfn_render_fwd_workspace_bytes(const vfn_render_params* p) {
    if (!p || p->n_rays < 0 || p->n_coarse < 1 || p->n_fine < 2) return VFN_ERR_INVALID;
    return (int64_t)carve(nullptr, p).bytes;
}

extern "C" int vfn_render_fwd(const vfn_render_params* p, const vfn_net_geom* vf_geom, const void* vf_packed16,
                              const vfn_net_geom* rn_geom, const void* rn_packed16, const float* uv, const float* pose,
                              const float* intrinsics, const float* t_vals, const float* far_coarse_per_ray,
                              const float* far_fine_per_ray, const float* density_scalars, const float* u_coarse,
                              const float* u_fine, const float* u_add, void* workspace, float* ray_dirs, float* z_vals,
                              float* points, float* normals, float* colors, float* weights, float* rgb, float* depth,
                              void* stream) {
    VFN_REQUIRE(p && vf_geom && rn_geom, "vfn_render_fwd: NULL argument");
    if (p->n_rays == 0) return VFN_OK;
    VfnReportScope report(p->status_word, p->clock_stamps, p->clock_slots);
    VFN_REQUIRE(p->n_rays > 0 && p->n_coarse >= 1 && p->n_fine >= 2, "vfn_render_fwd: bad sizes (n_rays=%d, n_coarse=%d, n_fine=%d)",
                p->n_rays, p->n_coarse, p->n_fine);
    VFN_REQUIRE(vf_packed16 && rn_packed16 && uv && pose && intrinsics && t_vals && density_scalars && workspace && ray_dirs && z_vals &&
                points && normals && colors && weights && rgb && depth, "vfn_render_fwd: NULL argument");
    const int n = p->n_rays, sc = p->n_coarse, nf = p->n_fine, st = sc + nf;
    VFN_REQUIRE((long long)n * st < (1ll << 22), "vfn_render_fwd: at most 4194303 samples per call");
    VFN_REQUIRE(p->colour_products == 0 || p->colour_products == 2 || p->colour_products == 3, "vfn_render_fwd: colour_products must be 0, 2 or 3");
    const Ws w = carve(workspace, p);
    int rc;
    // the draws the caller did not supply are elements of ONE Philox stream (seed, offset), laid out as vfn_fill_uniform would
    // fill one buffer with the segments coarse, fine, add (the facade's order); the kernels generate the elements they consume
    const int gen_c = p->perturb_coarse && !u_coarse, gen_f = p->perturb_fine && !u_fine, gen_a = !u_add;
    const long long base_f = gen_c ? (long long)n * sc : 0, base_a = base_f + (gen_f ? (long long)n * nf : 0);
    vfn_raygen_params rp = {n, sc, p->pose_is_quat, p->near_coarse, p->far_coarse};
    const int products = p->colour_products == 2 ? 2 : 3;
    if (p->sparse_colours) {
        // Sparse colours (opt-in; what evaluator.render_view asks for: evaluation/methods.py:528-540 keeps rgb and depth only).  A colour
        // enters rgb = sum_s w_s c_s only where w_s != 0 — a few percent of the samples — so the vector-field net runs on every
        // sample with its vector-only launch, the weights follow, and the fused VF + rendering launch covers the compacted list of
        // samples with w > 0 only (device-side count).  rgb, depth, weights, normals, z_vals, points: bit-identical to the dense
        // plan; `colors` holds zeros where w = 0.
        const float* uc = (p->perturb_coarse && u_coarse) ? u_coarse : nullptr;
        rc = vfn_internal_raygen(&rp, uv, pose, intrinsics, intrinsics, t_vals, far_coarse_per_ray, uc, gen_c, 0, p->seed, p->offset, w.directions, ray_dirs,
                                 w.cam_loc, w.z_c, w.pts_c, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_vf_mlp16_fwd(vf_geom, vf_packed16, w.pts_c, (int64_t)n * sc, w.normals_c, stream);
        if (rc != VFN_OK) return rc;
        vfn_density_params dq = p->density;
        dq.n_rays = n; dq.n_samples = sc;
        vfn_fine_params fq = {n, sc, nf, p->near_fine, p->far_fine, p->fine_range, p->window_step, p->span};
        rc = vfn_internal_density_fine(&dq, w.normals_c, ray_dirs, w.z_c, density_scalars, &fq, w.directions, w.cam_loc, far_fine_per_ray,
                                       (p->perturb_fine && u_fine) ? u_fine : nullptr, u_add, gen_f, gen_a, base_f, base_a, p->seed, p->offset, z_vals,
                                       points, w.src, w.new_pts, w.dst, (int64_t)n * sc, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_vf_mlp16_fwd(vf_geom, vf_packed16, w.new_pts, (int64_t)n * nf, w.normals_c + (size_t)n * sc * 3, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_scatter_rows3(w.normals_c, nullptr, w.dst, (int64_t)n * st, normals, nullptr, stream);
        if (rc != VFN_OK) return rc;
        dq.n_samples = st;
        rc = vfn_ray_density_weights(&dq, normals, ray_dirs, z_vals, density_scalars, nullptr, nullptr, weights, nullptr, nullptr, nullptr, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_internal_select_positive(weights, nullptr, nullptr, n, st, points, ray_dirs, w.cnt, w.off, w.k_dev, w.sel_sorted, w.pts_sel, w.dirs_sel, stream);
        if (rc != VFN_OK) return rc;
        if (hipMemsetAsync(colors, 0, (size_t)n * st * 3 * sizeof(float), (hipStream_t)stream) != hipSuccess) {
            vfn_set_error("vfn_render_fwd: could not clear the colours");
            return VFN_ERR_LAUNCH;
        }
        // (the fused launch writes the selected samples' normals again — the same values — and their colours at their sorted rows)
        if (p->timing_events[0]) (void)hipEventRecord((hipEvent_t)p->timing_events[0], (hipStream_t)stream);
        rc = vfn_internal_fused16_products_dev(vf_geom, vf_packed16, rn_geom, rn_packed16, w.pts_sel, w.dirs_sel, (int64_t)n * st, w.k_dev, 1, w.sel_sorted,
                                               products, normals, colors, stream);
        if (rc != VFN_OK) return rc;
        if (p->timing_events[1]) (void)hipEventRecord((hipEvent_t)p->timing_events[1], (hipStream_t)stream);
        return vfn_ray_density_weights(&dq, normals, ray_dirs, z_vals, density_scalars, colors, nullptr, weights, nullptr, rgb, depth, stream);
    }
    if (p->separate_launches) {
        // the same pipeline through the stand-alone entry points, eight launches (A/B timing of the merged plan; same values)
        const float* uc = p->perturb_coarse ? u_coarse : nullptr;
        const float* uf = p->perturb_fine ? u_fine : nullptr;
        const float* ua = u_add;
        if (gen_c) uc = w.u;
        if (gen_f) uf = w.u + base_f;
        if (gen_a) ua = w.u + base_a;
        const long long need = base_a + (gen_a ? (long long)n * nf : 0);
        if (need) { rc = vfn_fill_uniform(w.u, need, p->seed, p->offset, stream); if (rc != VFN_OK) return rc; }
        rc = vfn_raygen_uniform(&rp, uv, pose, intrinsics, t_vals, far_coarse_per_ray, uc, w.directions, ray_dirs, w.cam_loc, w.z_c, w.pts_c, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_vf_render_fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, w.pts_c, ray_dirs, (int64_t)n * sc, sc, nullptr, products,
                                            w.normals_c, w.colors_c, stream);
        if (rc != VFN_OK) return rc;
        vfn_density_params dq = p->density;
        dq.n_rays = n; dq.n_samples = sc;
        rc = vfn_ray_density_weights(&dq, w.normals_c, ray_dirs, w.z_c, density_scalars, nullptr, nullptr, nullptr, w.imax, nullptr, nullptr, stream);
        if (rc != VFN_OK) return rc;
        vfn_fine_params fq = {n, sc, nf, p->near_fine, p->far_fine, p->fine_range, p->window_step, p->span};
        rc = vfn_range_fine_sample_indexed(&fq, w.z_c, w.imax, w.directions, w.cam_loc, far_fine_per_ray, uf, ua, z_vals, points, nullptr, w.new_pts,
                                           w.dst, (int64_t)n * sc, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_vf_render_fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, w.new_pts, ray_dirs, (int64_t)n * nf, nf,
                                            w.dst + (size_t)n * sc, products, normals, colors, stream);
        if (rc != VFN_OK) return rc;
        rc = vfn_scatter_rows3(w.normals_c, w.colors_c, w.dst, (int64_t)n * sc, normals, colors, stream);
        if (rc != VFN_OK) return rc;
        dq.n_samples = st;
        return vfn_ray_density_weights(&dq, normals, ray_dirs, z_vals, density_scalars, colors, nullptr, weights, nullptr, rgb, depth, stream);
    }
    // The five launches of a range of rays [ray0, ray0 + nh) of the batch on one stream.  Every array is per ray (or per sample of a
    // ray), so a range works on slices; the Philox elements keep their batch-wide indices, the camera's z sign stays the batch's.
    auto stage = [&](int which, int ray0, int nh, hipStream_t s) -> int {
        const size_t r = (size_t)ray0;
        float* dirs_h = ray_dirs + r * 3;
        float* n_c = w.normals_c + r * sc * 3; float* c_c = w.colors_c + r * sc * 3;
        int32_t* dst_h = w.dst + r * st; int32_t* src_h = w.src + r * st;
        vfn_density_params dp = p->density;
        dp.n_rays = nh;
        switch (which) {
        case 0: {
            vfn_raygen_params rq = {nh, sc, p->pose_is_quat, p->near_coarse, p->far_coarse};
            return vfn_internal_raygen(&rq, uv + r * 2, pose + r * (p->pose_is_quat ? 7 : 16), intrinsics + r * 16, intrinsics, t_vals,
                                       far_coarse_per_ray ? far_coarse_per_ray + r : nullptr,
                                       (p->perturb_coarse && u_coarse) ? u_coarse + r * sc : nullptr, gen_c, (long long)r * sc, p->seed, p->offset,
                                       w.directions + r * 3, dirs_h, w.cam_loc + r * 3, w.z_c + r * sc, w.pts_c + r * sc * 3, s);
        }
        case 1:
            return vfn_vf_render_fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, w.pts_c + r * sc * 3, dirs_h, (int64_t)nh * sc, sc, nullptr,
                                                  products, n_c, c_c, s);
        case 2: {
            dp.n_samples = sc;
            vfn_fine_params fp = {nh, sc, nf, p->near_fine, p->far_fine, p->fine_range, p->window_step, p->span};
            return vfn_internal_density_fine(&dp, n_c, dirs_h, w.z_c + r * sc, density_scalars, &fp, w.directions + r * 3, w.cam_loc + r * 3,
                                             far_fine_per_ray ? far_fine_per_ray + r : nullptr, (p->perturb_fine && u_fine) ? u_fine + r * nf : nullptr,
                                             u_add ? u_add + r * nf : nullptr, gen_f, gen_a, base_f + (long long)r * nf, base_a + (long long)r * nf,
                                             p->seed, p->offset, z_vals + r * st, points + r * st * 3, src_h, w.new_pts + r * nf * 3, dst_h,
                                             (int64_t)nh * sc, s);
        }
        case 3:
            return vfn_vf_render_fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, w.new_pts + r * nf * 3, dirs_h, (int64_t)nh * nf, nf,
                                                  dst_h + (size_t)nh * sc, products, normals + r * st * 3, colors + r * st * 3, s);
        default:
            dp.n_samples = st;
            return vfn_internal_composite_gather(&dp, normals + r * st * 3, dirs_h, z_vals + r * st, density_scalars, colors + r * st * 3, src_h, n_c, c_c,
                                                 (int64_t)nh * sc, weights + r * st, rgb + r * 3, depth + r, s);
        }
    };
    hipStream_t main_s = (hipStream_t)stream;
    // optional timing events around the two fused launches (stages 1 and 3) on the stream they are issued on
    auto mark = [&](int slot) {
        if (p->timing_events[slot]) (void)hipEventRecord((hipEvent_t)p->timing_events[slot], main_s);
    };
    // streams: 1 = one stream, 2 = two halves, 0 = two halves when that pays: at least 512 rays and the two fused launches leave >= 5 %
    // of their workgroup slots empty (one workgroup of 128 points per CU and round of 256; e.g. 1 024 rays x (100 + 35) samples =
    // 800 + 280 workgroups = 4 + 2 rounds for 4.2 rounds of work: +13 %; whole rounds, or halves too small to fill the chip: nothing
    // to gain, and 256-ray calls lose 10 %)
    int parts = (p->streams >= 2 && p->streams <= MAX_SIDE + 1 && n >= 32 * p->streams) ? p->streams : 1;
    if (p->streams == 0 && n >= 512) {
        const long long wp = ((long long)n * sc + 127) / 128, wq = ((long long)n * nf + 127) / 128;
        const long long waste = (wp + 255) / 256 * 256 - wp + (wq + 255) / 256 * 256 - wq;
        if (20 * waste >= wp + wq) parts = 2;
    }
    Side* side = parts > 1 ? side_streams() : nullptr;
    if (!side) {
        for (int k = 0; k < 5; ++k) {
            if (k == 1 || k == 3) mark(k - 1);
            rc = stage(k, 0, n, main_s);
            if (rc != VFN_OK) return rc;
            if (k == 1 || k == 3) mark(k);
        }
        return VFN_OK;
    }
    // the batch in `parts` ranges of rays, all but the first on side streams forked from (and joined back into) the caller's: a
    // range's per-ray launches and the last, partial round of its fused launches run while the others' workgroups fill the chip
    int r0[MAX_SIDE + 2];
    for (int i = 0; i <= parts; ++i) r0[i] = i == parts ? n : (int)((long long)n * i / parts + 3) / 4 * 4;
    if (hipEventRecord(side->fork, main_s) != hipSuccess) { vfn_set_error("vfn_render_fwd: could not fork the side streams"); return VFN_ERR_LAUNCH; }
    for (int i = 1; i < parts; ++i)
        if (hipStreamWaitEvent(side->s[i - 1], side->fork, 0) != hipSuccess) { vfn_set_error("vfn_render_fwd: could not fork the side streams"); return VFN_ERR_LAUNCH; }
    rc = VFN_OK;
    for (int k = 0; k < 5 && rc == VFN_OK; ++k)
        for (int i = 0; i < parts && rc == VFN_OK; ++i)
            if (r0[i + 1] > r0[i]) {
                if (i == 0 && (k == 1 || k == 3)) mark(k - 1);
                rc = stage(k, r0[i], r0[i + 1] - r0[i], i == 0 ? main_s : side->s[i - 1]);
                if (i == 0 && (k == 1 || k == 3)) mark(k);
            }
    // (joined even after an error, so that the caller's stream never runs ahead of work already issued on a side stream)
    for (int i = 1; i < parts; ++i)
        if (hipEventRecord(side->join[i - 1], side->s[i - 1]) != hipSuccess || hipStreamWaitEvent(main_s, side->join[i - 1], 0) != hipSuccess) {
            vfn_set_error("vfn_render_fwd: could not join the side streams");
            return VFN_ERR_LAUNCH;
        }
    return rc;
}
