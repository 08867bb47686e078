// vfn_loss.hip — VFLoss (models/losses/vf_loss.py:34-87) and the trainer's centre-ball selection (models/helpers/functions.py:137-157,
// train/vector_field_nerf_train.py:203-214) as three launches per training step: one reduction pass over everything the loss reads,
// a one-workgroup finish (terms, total, the data-dependent row count of the supervision mean), and — at backward() — one elementwise
// pass that writes every gradient.  The tensor-op formulation is ~40 small launches forward and backward over [N*S_t, 3] operands
// (cat, sub, abs, pow, mean, where, normalize, ...) and, in the reference, a boolean-mask compaction whose row count the host waits for.
//
//   rgb_loss          mean |rgb - rgb_gt|                                  over 3 N values
//   depth_loss        mean min(|depth - depth_gt|, clamp)                  over N            (0 without depth ground truth)
//   unit_norm_loss    mean (|n| - 1)^2                                     over the M normals
//   supervision_loss  mean (pred - gt)^2                                   over 3 R values, R = rows of the segments (+ the ray samples
//                                                                          inside the centre ball, gt = normalize(p - centroid), when asked for)
//   smaller_loss      mean relu(|n| - 1)^2                                 over M            (from norm_smaller_than_one_start on)
// The directional-derivative term (a plain mean of a vector that only the training-mode path produces) stays with the caller.
#include <string.h>
#include "vfn_common.h"

namespace {

constexpr int LOSS_BLOCKS = 1024;       // partial-sum slots (grid size cap)
enum { S_RGB = 0, S_DEPTH, S_UNIT, S_SUP, S_SMALL, S_ROWS, S_N };

struct LossArgs {
    vfn_loss_params p;
    const float *rgb, *rgb_gt, *depth, *depth_gt, *normals, *points;
    const float* sup_pred[3];
    const float* sup_gt[3];
    float* partials;        // [gridDim.x][S_N]
    float* out;             // [8]: the five terms + 0 (dd) at 0..5, weighted total at 6, supervision rows at 7
    float* scales;          // [4]: 1 / (3 R) for the backward pass (0 when R = 0), spare
    const float* grad_out;  // device scalar (backward)
    float *d_rgb, *d_depth, *d_normals;
    float* d_sup[3];
};

__device__ __forceinline__ float vnorm(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// the item space: rays | normals | segment 0 | segment 1 | segment 2
__device__ __forceinline__ void item_sums(const LossArgs& a, long long i, float (&s)[S_N]) {
    const vfn_loss_params& p = a.p;
    if (i < p.n_rays) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s[S_RGB] += fabsf(a.rgb[i * 3 + c] - a.rgb_gt[i * 3 + c]);
        if (p.has_depth) s[S_DEPTH] += fminf(fabsf(a.depth[i] - a.depth_gt[i]), p.depth_clamp);
        return;
    }
    i -= p.n_rays;
    if (i < p.n_normals) {
        const float x = a.normals[i * 3], y = a.normals[i * 3 + 1], z = a.normals[i * 3 + 2];
        const float n = vnorm(x, y, z);
        s[S_UNIT] += (n - 1.f) * (n - 1.f);
        if (p.smaller_on) { const float r = fmaxf(n - 1.f, 0.f); s[S_SMALL] += r * r; }
        if (p.ray_center) {
            const float dx = a.points[i * 3] - p.centroid[0], dy = a.points[i * 3 + 1] - p.centroid[1], dz = a.points[i * 3 + 2] - p.centroid[2];
            const float d = vnorm(dx, dy, dz);
            if (d < p.radius) {
                const float inv = 1.f / fmaxf(d, 1e-12f);                      // F.normalize
                const float ex = x - dx * inv, ey = y - dy * inv, ez = z - dz * inv;
                s[S_SUP] += ex * ex + ey * ey + ez * ez;
                s[S_ROWS] += 1.f;
            }
        }
        return;
    }
    i -= p.n_normals;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (i < p.n_sup[k]) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float e = a.sup_pred[k][i * 3 + c] - a.sup_gt[k][i * 3 + c]; s[S_SUP] += e * e; }
            return;
        }
        i -= p.n_sup[k];
    }
}

__global__ __launch_bounds__(256) void vfn_loss_reduce_kernel(const LossArgs a) {
    __shared__ float red[4][S_N];
    const long long total = a.p.n_rays + a.p.n_normals + a.p.n_sup[0] + a.p.n_sup[1] + a.p.n_sup[2];
    float s[S_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) item_sums(a, i, s);
#pragma unroll
    for (int q = 0; q < S_N; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < S_N; ++q) red[wave][q] = s[q];
    __syncthreads();
    if (threadIdx.x < S_N) a.partials[blockIdx.x * S_N + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void vfn_loss_finish_kernel(const LossArgs a, int n_blocks) {
    double s[S_N] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += 64)
#pragma unroll
        for (int q = 0; q < S_N; ++q) s[q] += (double)a.partials[b * S_N + q];
#pragma unroll
    for (int q = 0; q < S_N; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
    if (threadIdx.x != 0) return;
    const vfn_loss_params& p = a.p;
    const double rows = (double)(p.n_sup[0] + p.n_sup[1] + p.n_sup[2]) + s[S_ROWS];
    const double rgb = p.n_rays ? s[S_RGB] / (3.0 * p.n_rays) : 0.0;
    const double depth = (p.has_depth && p.n_rays) ? s[S_DEPTH] / (double)p.n_rays : 0.0;
    const double unit = p.n_normals ? s[S_UNIT] / (double)p.n_normals : 0.0;
    const double sup = rows > 0 ? s[S_SUP] / (3.0 * rows) : 0.0;
    const double small_ = (p.smaller_on && p.n_normals) ? s[S_SMALL] / (double)p.n_normals : 0.0;
    a.out[0] = (float)rgb; a.out[1] = (float)depth; a.out[2] = (float)unit; a.out[3] = (float)sup; a.out[4] = (float)small_; a.out[5] = 0.f;
    a.out[6] = (float)(p.w_rgb * rgb + p.w_depth * depth + p.w_unit * unit + p.w_sup * sup + p.w_smaller * small_);
    a.out[7] = (float)rows;
    a.scales[0] = rows > 0 ? (float)(1.0 / (3.0 * rows)) : 0.f;
}

__global__ __launch_bounds__(256) void vfn_loss_bwd_kernel(const LossArgs a) {
    const vfn_loss_params& p = a.p;
    const long long total = p.n_rays + p.n_normals + p.n_sup[0] + p.n_sup[1] + p.n_sup[2];
    const float g = a.grad_out ? a.grad_out[0] : 1.f;
    const float k_rgb = p.n_rays ? g * p.w_rgb / (3.f * (float)p.n_rays) : 0.f, k_depth = p.n_rays ? g * p.w_depth / (float)p.n_rays : 0.f;
    const float k_unit = p.n_normals ? g * p.w_unit * 2.f / (float)p.n_normals : 0.f, k_small = p.n_normals ? g * p.w_smaller * 2.f / (float)p.n_normals : 0.f;
    const float k_sup = g * p.w_sup * 2.f * a.scales[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long j = i;
        if (j < p.n_rays) {
            if (a.d_rgb)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float e = a.rgb[j * 3 + c] - a.rgb_gt[j * 3 + c];
                    a.d_rgb[j * 3 + c] = e > 0.f ? k_rgb : (e < 0.f ? -k_rgb : 0.f);          // d|x| = sign(x), 0 at 0 (torch)
                }
            if (a.d_depth) {
                float v = 0.f;
                if (p.has_depth) {
                    const float e = a.depth[j] - a.depth_gt[j];
                    if (fabsf(e) <= p.depth_clamp) v = e > 0.f ? k_depth : (e < 0.f ? -k_depth : 0.f);   // clamp(max=c): gradient passes where |x| <= c
                }
                a.d_depth[j] = v;
            }
            continue;
        }
        j -= p.n_rays;
        if (j < p.n_normals) {
            if (!a.d_normals) continue;
            const float x = a.normals[j * 3], y = a.normals[j * 3 + 1], z = a.normals[j * 3 + 2];
            const float n = vnorm(x, y, z);
            float f = 0.f;                                                  // d|n| = n / |n| (0 at the origin, as torch.linalg.vector_norm)
            if (n > 0.f) {
                f = k_unit * (n - 1.f);
                if (p.smaller_on) f += k_small * fmaxf(n - 1.f, 0.f);
                f /= n;
            }
            float gx = f * x, gy = f * y, gz = f * z;
            if (p.ray_center) {
                const float dx = a.points[j * 3] - p.centroid[0], dy = a.points[j * 3 + 1] - p.centroid[1], dz = a.points[j * 3 + 2] - p.centroid[2];
                const float d = vnorm(dx, dy, dz);
                if (d < p.radius) {
                    const float inv = 1.f / fmaxf(d, 1e-12f);
                    gx += k_sup * (x - dx * inv); gy += k_sup * (y - dy * inv); gz += k_sup * (z - dz * inv);
                }
            }
            a.d_normals[j * 3] = gx; a.d_normals[j * 3 + 1] = gy; a.d_normals[j * 3 + 2] = gz;
            continue;
        }
        j -= p.n_normals;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (j < p.n_sup[k]) {
                if (a.d_sup[k])
#pragma unroll
                    for (int c = 0; c < 3; ++c) a.d_sup[k][j * 3 + c] = k_sup * (a.sup_pred[k][j * 3 + c] - a.sup_gt[k][j * 3 + c]);
                break;
            }
            j -= p.n_sup[k];
        }
    }
}

int fill(LossArgs& a, const vfn_loss_params* p, const float* rgb, const float* rgb_gt, const float* depth, const float* depth_gt, const float* normals,
         const float* points, const float* const* sup_pred, const float* const* sup_gt, const char* who) {
    VFN_REQUIRE(p, "%s: NULL parameters", who);
    VFN_REQUIRE(p->n_rays >= 0 && p->n_normals >= 0 && p->n_sup[0] >= 0 && p->n_sup[1] >= 0 && p->n_sup[2] >= 0, "%s: negative sizes", who);
    VFN_REQUIRE(p->n_rays == 0 || (rgb && rgb_gt && (!p->has_depth || (depth && depth_gt))), "%s: NULL rgb / depth argument", who);
    VFN_REQUIRE(p->n_normals == 0 || (normals && (!p->ray_center || points)), "%s: NULL normals / points argument", who);
    for (int k = 0; k < 3; ++k)
        VFN_REQUIRE(p->n_sup[k] == 0 || (sup_pred && sup_gt && sup_pred[k] && sup_gt[k]), "%s: supervision segment %d is NULL", who, k);
    memset(&a, 0, sizeof(a));
    a.p = *p;
    a.rgb = rgb; a.rgb_gt = rgb_gt; a.depth = depth; a.depth_gt = depth_gt; a.normals = normals; a.points = points;
    for (int k = 0; k < 3; ++k) { a.sup_pred[k] = (sup_pred && p->n_sup[k]) ? sup_pred[k] : nullptr; a.sup_gt[k] = (sup_gt && p->n_sup[k]) ? sup_gt[k] : nullptr; }
    return VFN_OK;
}

inline unsigned loss_grid(const vfn_loss_params* p) {
    const long long total = p->n_rays + p->n_normals + p->n_sup[0] + p->n_sup[1] + p->n_sup[2];
    const long long blocks = (total + 1023) / 1024;          // >= 4 items per thread
    return (unsigned)(blocks < 1 ? 1 : (blocks > LOSS_BLOCKS ? LOSS_BLOCKS : blocks));
}

}  // namespace

extern "C" int64_t vfn_vf_loss_workspace_bytes(void) { return (int64_t)(LOSS_BLOCKS * S_N + 8) * sizeof(float); }

extern "C" int vfn_vf_loss_fwd(const vfn_loss_params* p, const float* rgb, const float* rgb_gt, const float* depth, const float* depth_gt,
                               const float* normals, const float* points, const float* const* sup_pred, const float* const* sup_gt,
                               void* workspace, float* out_terms, void* stream) {
    LossArgs a;
    int rc = fill(a, p, rgb, rgb_gt, depth, depth_gt, normals, points, sup_pred, sup_gt, "vfn_vf_loss_fwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(workspace && out_terms, "vfn_vf_loss_fwd: NULL workspace / output");
    a.partials = static_cast<float*>(workspace);
    a.scales = a.partials + LOSS_BLOCKS * S_N;
    a.out = out_terms;
    const unsigned grid = loss_grid(p);
    hipLaunchKernelGGL(vfn_loss_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(vfn_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, (int)grid);
    return vfn_check_launch("vfn_vf_loss_fwd");
}

extern "C" int vfn_vf_loss_bwd(const vfn_loss_params* p, const float* rgb, const float* rgb_gt, const float* depth, const float* depth_gt,
                               const float* normals, const float* points, const float* const* sup_pred, const float* const* sup_gt,
                               const void* workspace, const float* grad_out, float* d_rgb, float* d_depth, float* d_normals,
                               float* const* d_sup, void* stream) {
    LossArgs a;
    int rc = fill(a, p, rgb, rgb_gt, depth, depth_gt, normals, points, sup_pred, sup_gt, "vfn_vf_loss_bwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(workspace, "vfn_vf_loss_bwd: NULL workspace (the one the forward call filled)");
    a.partials = const_cast<float*>(static_cast<const float*>(workspace));
    a.scales = a.partials + LOSS_BLOCKS * S_N;
    a.grad_out = grad_out;
    a.d_rgb = d_rgb; a.d_depth = d_depth; a.d_normals = d_normals;
    for (int k = 0; k < 3; ++k) a.d_sup[k] = (d_sup && p->n_sup[k]) ? d_sup[k] : nullptr;
    hipLaunchKernelGGL(vfn_loss_bwd_kernel, dim3(loss_grid(p) * 4 > 4096 ? 4096 : loss_grid(p) * 4), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_loss_bwd");
}
