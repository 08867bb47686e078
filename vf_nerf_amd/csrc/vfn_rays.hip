// vfn_rays.hip — the per-ray (HBM / latency bound) kernels of the VF-NeRF hot path:
//   K1  pinhole ray generation + stratified uniform sampler
//   K3  windowed cosine -> Laplace density -> VolSDF weights (wavefront scan) -> argmax / composite
//   K3b range fine sampler (window / uniform extras, rank sort, points)
//   Philox uniform fill for production sampling.
//
// One wavefront owns one ray end to end (window stencil, scan, argmax, sort are all wave-local:
// LDS + cross-lane shuffles, no inter-workgroup traffic).  Arithmetic follows the reference's
// operation order with FMA contraction disabled so that z values / sample indices reproduce the
// PyTorch CPU path bit for bit on identical inputs (tests/test_rays_gpu.py).
#pragma clang fp contract(off)
#include <string.h>
#include "vfn_common.h"

namespace {

// torch.argmax order, shared by every argmax in this file (ray_sampler.py:277 takes torch.argmax of the proposal weights): the
// first maximum; a NaN beats every number (the first NaN wins).  `oi` / `besti` break ties towards the smaller index.
__device__ __forceinline__ bool argmax_takes(float ob, int oi, float best, int besti) {
    if (ob != ob) return best == best || oi < besti;
    return best == best && (ob > best || (ob == best && oi < besti));
}

constexpr int WAVE = 64;
constexpr int RAYS_PER_BLOCK = 4;   // one wave per ray, 256 threads
constexpr int MAX_SAMPLES = 512;    // per-ray samples supported by the LDS carve-up (<= 40 KiB dynamic LDS)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

// Element e of the uniform stream (seed, offset) as vfn_fill_uniform lays it out: one Philox block per 4 consecutive outputs.
struct PhiloxStream { unsigned long long seed, offset; };
__device__ __forceinline__ float philox_uniform(const PhiloxStream& ps, long long e) {
    const unsigned long long ctr = ps.offset + (unsigned long long)(e >> 2);
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)ps.seed, (uint32_t)(ps.seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    const int i = (int)(e & 3);
    const uint32_t v = i == 0 ? c[0] : (i == 1 ? c[1] : (i == 2 ? c[2] : c[3]));
    return (float)(v >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// K1: rays + coarse z + points
//   utils/rendering.py:12-60, utils/pinhole_model.py:9-63, models/samplers/ray_sampler.py:49-80,113-142
// ------------------------------------------------------------------------------------------------
constexpr int K1_RAYS = 16;

struct RaygenArgs {
    vfn_raygen_params p;
    const float* uv;
    const float* pose;
    const float* K;
    const float* t_vals;
    const float* far_per_ray;
    const float* u;
    float* directions;
    float* ray_dirs;
    float* cam_loc;
    float* z_vals;
    float* points;
    // vfn_render_fwd: the draws come from the Philox stream instead of `u` (element u_base + ray * S + s of it), the values
    // vfn_fill_uniform would have written there
    int gen_u;
    long long u_base;
    PhiloxStream ps;
    const float* k_sign;     // intrinsics whose [1][1] entry gives the sign of the camera's z axis (utils/rendering.py:42 reads ray 0 of
                             // the BATCH): NULL = K; a launch over part of a batch passes the batch's first matrix
};

__device__ __forceinline__ float coarse_z(float near, float far, float t) { return near * (1.0f - t) + far * t; }

__global__ __launch_bounds__(256) void vfn_raygen_kernel(const RaygenArgs a) {
    __shared__ float s_ray[K1_RAYS][8];  // dir(3), cam(3), far, pad
    const int tid = threadIdx.x;
    const int ray0 = blockIdx.x * K1_RAYS;
    const int n = a.p.n_rays;
    if (tid < K1_RAYS && ray0 + tid < n) {
        const int r = ray0 + tid;
        float P[3][4];
        if (a.p.pose_is_quat) {
            const float* q7 = a.pose + (size_t)r * 7;
            float q0 = q7[0], q1 = q7[1], q2 = q7[2], q3 = q7[3];
            const float nq = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
            const float qr = q0 / nq, qi = q1 / nq, qj = q2 / nq, qk = q3 / nq;
            P[0][0] = 1 - 2 * (qj * qj + qk * qk); P[0][1] = 2 * (qj * qi - qk * qr); P[0][2] = 2 * (qi * qk + qr * qj);
            P[1][0] = 2 * (qj * qi + qk * qr); P[1][1] = 1 - 2 * (qi * qi + qk * qk); P[1][2] = 2 * (qj * qk - qi * qr);
            P[2][0] = 2 * (qk * qi - qj * qr); P[2][1] = 2 * (qj * qk + qi * qr); P[2][2] = 1 - 2 * (qi * qi + qj * qj);
            P[0][3] = q7[4]; P[1][3] = q7[5]; P[2][3] = q7[6];
        } else {
            const float* m = a.pose + (size_t)r * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) P[i][j] = m[i * 4 + j];
        }
        const float* Kr = a.K + (size_t)r * 16;
        const float fx = Kr[0], sk = Kr[1], cx = Kr[2], fy = Kr[5], cy = Kr[6];
        const float fy0 = (a.k_sign ? a.k_sign : a.K)[5];  // sign is read from ray 0 only (utils/rendering.py:42)
        const float zs = (fy0 > 0.f) ? 1.f : ((fy0 < 0.f) ? -1.f : 0.f);
        const float u = a.uv[(size_t)r * 2 + 0], v = a.uv[(size_t)r * 2 + 1];
        const float za = fabsf(zs);
        const float x = (u - cx + cy * sk / fy - sk * v / fy) / fx * za;
        const float y = (v - cy) / fy * za;
        float w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = ((P[i][0] * x + P[i][1] * y) + P[i][2] * zs) + P[i][3];
        const float cam[3] = {P[0][3], P[1][3], P[2][3]};
        float d[3] = {w[0] - cam[0], w[1] - cam[1], w[2] - cam[2]};
        const float nd = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a.directions[(size_t)r * 3 + i] = d[i];
            a.ray_dirs[(size_t)r * 3 + i] = d[i] / nd;
            a.cam_loc[(size_t)r * 3 + i] = cam[i];
            s_ray[tid][i] = d[i];
            s_ray[tid][3 + i] = cam[i];
        }
        s_ray[tid][6] = a.far_per_ray ? a.far_per_ray[r] : a.p.far;
    }
    __syncthreads();
    const int S = a.p.n_samples;
    const int total = K1_RAYS * S;
    for (int idx = tid; idx < total; idx += blockDim.x) {
        const int lr = idx / S, s = idx - lr * S;
        const int r = ray0 + lr;
        if (r >= n) break;
        const float near = a.p.near, far = s_ray[lr][6];
        float z = coarse_z(near, far, a.t_vals[s]);
        if (a.u || a.gen_u) {
            const float zl = (s > 0) ? coarse_z(near, far, a.t_vals[s - 1]) : z;
            const float zu = (s < S - 1) ? coarse_z(near, far, a.t_vals[s + 1]) : z;
            const float upper = (s < S - 1) ? 0.5f * (zu + z) : z;
            const float lower = (s > 0) ? 0.5f * (z + zl) : z;
            z = lower + (upper - lower) * (a.gen_u ? philox_uniform(a.ps, a.u_base + (long long)r * S + s) : a.u[(size_t)r * S + s]);
        }
        const size_t o = (size_t)r * S + s;
        a.z_vals[o] = z;
        a.points[o * 3 + 0] = s_ray[lr][3] + z * s_ray[lr][0];
        a.points[o * 3 + 1] = s_ray[lr][4] + z * s_ray[lr][1];
        a.points[o * 3 + 2] = s_ray[lr][5] + z * s_ray[lr][2];
    }
}

// ------------------------------------------------------------------------------------------------
// K3: density -> weights -> argmax / composite, one wave per ray
//   models/nerf/vector_field_nerf.py:442-474, models/helpers/functions.py:41-72,
//   models/helpers/density_functions.py:129-204, utils/rendering.py:122-148,
//   models/nerf/vector_field_nerf.py:322-323
// ------------------------------------------------------------------------------------------------
struct DensityArgs {
    vfn_density_params p;
    const float* normals;
    const float* ray_dirs;
    const float* z_vals;
    const float* scalars;  // beta, mean, scale (raw)
    const float* colors;
    float* sigma;
    float* weights;
    long long* argmax;
    float* rgb;
    float* depth;
    // vfn_render_fwd, composite pass: sorted sample j of the ray is stored row src[ray * S + j]; rows below n_stored_c are
    // proposal samples whose normal / colour still sit in generation order in normals_c / colors_c — they are moved to their
    // sorted position in `normals` / `colors` (both written here) on the way in (what vfn_scatter_rows3 did in its own launch)
    const int* src;
    const float* normals_c;
    const float* colors_c;
    int n_stored_c;
};

__device__ __forceinline__ float laplace_cdf(float x, float beta, float scale, float mean) {
    const float a = x - mean;
    const float sg = (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f);
    const float t = 1.0f - expf(-fabsf(a) / beta);
    return scale * (0.5f + (0.5f * sg) * t);
}

__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// one ray, one wave; `su` = S * 5 floats of LDS owned by the wave.  Returns the index of the first maximum of the weights.
__device__ __forceinline__ int density_ray(const DensityArgs& a, float* su, int ray, int lane) {
    const int S = a.p.n_samples;
    float* sz = su + (size_t)S * 3;        // z [S]
    float* se = sz + S;                    // free energy / weights [S]

    const float* nrm = a.normals + (size_t)ray * S * 3;
    for (int j = lane; j < S; j += WAVE) {
        if (a.src) {
            const int row = a.src[(size_t)ray * S + j];
            if (row < a.n_stored_c) {
                float* no = const_cast<float*>(nrm) + j * 3;
                float* co = const_cast<float*>(a.colors) + ((size_t)ray * S + j) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) { no[c] = a.normals_c[(size_t)row * 3 + c]; co[c] = a.colors_c[(size_t)row * 3 + c]; }
            }
        }
        const float x = nrm[j * 3 + 0], y = nrm[j * 3 + 1], z = nrm[j * 3 + 2];
        const float nn = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-8f);
        su[j * 3 + 0] = x / nn; su[j * 3 + 1] = y / nn; su[j * 3 + 2] = z / nn;
        sz[j] = a.z_vals[(size_t)ray * S + j];
    }
    float d[3] = {a.ray_dirs[(size_t)ray * 3 + 0], a.ray_dirs[(size_t)ray * 3 + 1], a.ray_dirs[(size_t)ray * 3 + 2]};
    {
        const float nd = fmaxf(sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), 1e-8f);
        d[0] /= nd; d[1] /= nd; d[2] /= nd;
    }
    const float beta = fminf(fmaxf(a.scalars[0], a.p.beta_min), a.p.beta_max);
    const float mean = fminf(fmaxf(a.scalars[1], a.p.mean_min), a.p.mean_max);
    const float scale = fmaxf(fabsf(a.scalars[2]), a.p.scale_min);
    const float cdf_cut = laplace_cdf(a.p.cutoff, beta, scale, mean);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    const int W = a.p.n_window;
    const int start = (int)((W + 1) / 2.0 + 1);
    const float wgt = 1.0f / (float)W;  // torch.ones(W) / W
    float wnorm = 0.f;
    for (int i = 0; i < W; ++i) wnorm += fabsf(wgt);
    const int L = S - 1;
    const int lo = start, hi = L - start;  // interior [lo, hi)

    // sigma and free energy
    for (int j = lane; j < S; j += WAVE) {
        float sg = 0.f;
        if (j < L) {
            const float* uj = su + j * 3;
            float c = dot3(uj, su + (j + 1) * 3);
            if (j >= lo && j < hi) {
                c = c * wgt / wnorm;
                for (int i = 1; i < start - 1; ++i) {
                    const float f = dot3(uj, su + (j + 1 + i) * 3);
                    const float b = dot3(uj, su + (j - i) * 3);
                    c = (c + f * wgt / wnorm) + b * wgt / wnorm;
                }
            }
            const float c_ray = dot3(uj, d);
            sg = fmaxf(laplace_cdf(-c, beta, scale, mean) - cdf_cut, 0.f);
            if (c_ray < a.p.dir_to_normal_th && c < 0.f) sg = 0.f;
        }
        if (a.sigma) a.sigma[(size_t)ray * S + j] = sg;
        const float delta = (j < L) ? (sz[j + 1] - sz[j]) : 1e10f;
        se[j] = delta * sg;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // exclusive scan of the free energy along the ray: each lane owns C consecutive samples, lane
    // totals are scanned across the wavefront (fp64 like torch.cumsum's CPU accumulator).
    const int C = (S + WAVE - 1) / WAVE;
    const int j0 = lane * C;
    double tot = 0.0;
    for (int i = 0; i < C; ++i) { const int j = j0 + i; if (j < S) tot += (double)se[j]; }
    double incl = tot;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const double up = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += up;
    }
    double run = incl - tot;  // exclusive prefix of this lane's first sample
    float wsum_l = 0.f;
    for (int i = 0; i < C; ++i) {
        const int j = j0 + i;
        if (j < S) {
            const float e = se[j];
            const float T = expf(-(float)run);
            const float w = (1.0f - expf(-e)) * T;
            run += (double)e;
            se[j] = w;
            wsum_l += w;
        }
    }
    const float wsum = wave_sum(wsum_l);
    const float den = wsum + 1e-5f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    float best = -INFINITY;
    int besti = 0x7fffffff;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* col = a.colors ? a.colors + (size_t)ray * S * 3 : nullptr;
    for (int j = lane; j < S; j += WAVE) {
        float w = se[j];
        if (a.p.normalize) w = w / den;
        if (a.weights) a.weights[(size_t)ray * S + j] = w;
        if (argmax_takes(w, j, best, besti)) { best = w; besti = j; }
        if (col) {
            acc[0] += w * col[j * 3 + 0]; acc[1] += w * col[j * 3 + 1]; acc[2] += w * col[j * 3 + 2];
            acc[3] += w * sz[j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(besti, o, WAVE);
        if (argmax_takes(ob, oi, best, besti)) { best = ob; besti = oi; }
    }
    besti = (besti == 0x7fffffff) ? 0 : besti;
    if (a.argmax && lane == 0) a.argmax[ray] = besti;
    if (col) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
            a.rgb[(size_t)ray * 3 + 0] = acc[0]; a.rgb[(size_t)ray * 3 + 1] = acc[1]; a.rgb[(size_t)ray * 3 + 2] = acc[2];
            a.depth[ray] = acc[3];
        }
    }
    return besti;
}

__global__ __launch_bounds__(256) void vfn_density_kernel(const DensityArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= a.p.n_rays) return;  // whole wave exits together; no block-level barriers below
    (void)density_ray(a, dsm + (size_t)wv * a.p.n_samples * 5, ray, lane);
}

// ------------------------------------------------------------------------------------------------
// K3-bwd: gradient of (rgb, depth, weights) wrt colours, normals and the three density scalars.
// Recomputes the forward quantities of the ray in LDS, then walks the chain backwards:
//   w = what / (sum what + 1e-5),  what_j = (1 - exp(-e_j)) T_j,  T_j = exp(-sum_{i<j} e_i),  e_j = delta_j sigma_j,
//   sigma_j = relu(s F(-c_j) - s F(cutoff)) (0 where masked),  c_j = windowed cosine of unit normals.
// One wave per ray; prefix / suffix sums are wavefront scans; scalar gradients leave by one atomicAdd per ray.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_SAMPLES_BWD = 256;

struct DensityBwdArgs {
    vfn_density_params p;
    const float* normals;
    const float* ray_dirs;
    const float* z_vals;
    const float* scalars;
    const float* colors;    // may be NULL
    const float* d_rgb;     // [N,3] may be NULL
    const float* d_depth;   // [N]   may be NULL
    const float* d_weights; // [N,S] may be NULL
    const float* d_sigma;   // [N,S] may be NULL: an upstream gradient on sigma itself (get_density under autograd)
    float* d_normals;       // [N,S,3] accumulated into (+=)
    float* d_colors;        // [N,S,3] written (may be NULL)
    float* d_scalars;       // [3] atomically accumulated: raw beta, mean, scale
};

__device__ __forceinline__ double wave_excl_scan_d(double v, int lane) {
    double incl = v;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const double up = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += up;
    }
    return incl - v;
}

// one ray by one wave; the ray's contribution to the gradients of the three density scalars comes back in (gb, gm, gs)
__device__ __forceinline__ void density_bwd_ray(const DensityBwdArgs& a, float* bsm, int wv, int lane, int ray, float& gb_out, float& gm_out,
                                                float& gs_out) {
    const int S = a.p.n_samples;
    float* su = bsm + (size_t)wv * S * 12;  // unit normals [S][3]
    float* sinv = su + (size_t)S * 3;       // 1 / max(|n|, eps)
    float* sz = sinv + S;
    float* sc = sz + S;                     // windowed cosine
    float* se = sc + S;                     // free energy
    float* sT = se + S;                     // transmittance
    float* sw = sT + S;                     // un-normalised weight
    float* sg = sw + S;                     // dL/d what, later dL/d c (gc)
    float* sact = sg + S;                   // 1 where sigma is on its differentiable branch
    float* sdl = sact + S;                  // delta

    const float* nrm = a.normals + (size_t)ray * S * 3;
    for (int j = lane; j < S; j += WAVE) {
        const float x = nrm[j * 3 + 0], y = nrm[j * 3 + 1], z = nrm[j * 3 + 2];
        const float nn = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-8f);
        su[j * 3 + 0] = x / nn; su[j * 3 + 1] = y / nn; su[j * 3 + 2] = z / nn;
        sinv[j] = 1.0f / nn;
        sz[j] = a.z_vals[(size_t)ray * S + j];
    }
    float d[3] = {a.ray_dirs[(size_t)ray * 3 + 0], a.ray_dirs[(size_t)ray * 3 + 1], a.ray_dirs[(size_t)ray * 3 + 2]};
    {
        const float nd = fmaxf(sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), 1e-8f);
        d[0] /= nd; d[1] /= nd; d[2] /= nd;
    }
    const float braw = a.scalars[0], mraw = a.scalars[1], sraw = a.scalars[2];
    const float beta = fminf(fmaxf(braw, a.p.beta_min), a.p.beta_max);
    const float mean = fminf(fmaxf(mraw, a.p.mean_min), a.p.mean_max);
    const float scale = fmaxf(fabsf(sraw), a.p.scale_min);
    const float cdf_cut = laplace_cdf(a.p.cutoff, beta, scale, mean);
    __builtin_amdgcn_wave_barrier();

    const int W = a.p.n_window;
    const int start = (int)((W + 1) / 2.0 + 1);
    const float wgt = 1.0f / (float)W;
    float wnorm = 0.f;
    for (int i = 0; i < W; ++i) wnorm += fabsf(wgt);
    const float coef = wgt / wnorm;
    const int L = S - 1;
    const int lo = start, hi = L - start;

    for (int j = lane; j < S; j += WAVE) {
        float sgm = 0.f, c = 1.f, act = 0.f;
        if (j < L) {
            const float* uj = su + j * 3;
            c = dot3(uj, su + (j + 1) * 3);
            if (j >= lo && j < hi) {
                c = c * wgt / wnorm;
                for (int i = 1; i < start - 1; ++i) {
                    const float f = dot3(uj, su + (j + 1 + i) * 3);
                    const float b = dot3(uj, su + (j - i) * 3);
                    c = (c + f * wgt / wnorm) + b * wgt / wnorm;
                }
            }
            const float c_ray = dot3(uj, d);
            const float raw = laplace_cdf(-c, beta, scale, mean) - cdf_cut;
            const bool masked = (c_ray < a.p.dir_to_normal_th && c < 0.f);
            if (raw > 0.f && !masked) { sgm = raw; act = 1.f; }
        }
        sc[j] = c;
        sact[j] = act;
        const float delta = (j < L) ? (sz[j + 1] - sz[j]) : 1e10f;
        sdl[j] = delta;
        se[j] = delta * sgm;
    }
    __builtin_amdgcn_wave_barrier();

    // forward scan: transmittance, weights
    const int C = (S + WAVE - 1) / WAVE;
    const int j0 = lane * C;
    double tot = 0.0;
    for (int i = 0; i < C; ++i) { const int j = j0 + i; if (j < S) tot += (double)se[j]; }
    double run = wave_excl_scan_d(tot, lane);
    float wsum_l = 0.f;
    for (int i = 0; i < C; ++i) {
        const int j = j0 + i;
        if (j < S) {
            const float e = se[j];
            const float T = expf(-(float)run);
            const float w = (1.0f - expf(-e)) * T;
            run += (double)e;
            sT[j] = T; sw[j] = w;
            wsum_l += w;
        }
    }
    const float wsum = wave_sum(wsum_l);
    const float den = a.p.normalize ? (wsum + 1e-5f) : 1.0f;
    __builtin_amdgcn_wave_barrier();

    // upstream gradient of the weights
    float drgb[3] = {0.f, 0.f, 0.f}, ddep = 0.f;
    if (a.d_rgb) { drgb[0] = a.d_rgb[(size_t)ray * 3 + 0]; drgb[1] = a.d_rgb[(size_t)ray * 3 + 1]; drgb[2] = a.d_rgb[(size_t)ray * 3 + 2]; }
    if (a.d_depth) ddep = a.d_depth[ray];
    const float* col = a.colors ? a.colors + (size_t)ray * S * 3 : nullptr;
    float dot_l = 0.f;
    for (int j = lane; j < S; j += WAVE) {
        const float w = sw[j] / den;
        float g = ddep * sz[j];
        if (col) g += (drgb[0] * col[j * 3 + 0] + drgb[1] * col[j * 3 + 1]) + drgb[2] * col[j * 3 + 2];
        if (a.d_weights) g += a.d_weights[(size_t)ray * S + j];
        if (a.d_colors) {
            float* dc = a.d_colors + ((size_t)ray * S + j) * 3;
            dc[0] = w * drgb[0]; dc[1] = w * drgb[1]; dc[2] = w * drgb[2];
        }
        sg[j] = g;
        dot_l += g * w;
    }
    const float gdot = wave_sum(dot_l);
    __builtin_amdgcn_wave_barrier();
    // dL/d what_j ; q_j = ghat_j * what_j ; suffix (exclusive) sums of q
    double qtot = 0.0;
    for (int i = 0; i < C; ++i) {
        const int j = j0 + i;
        if (j < S) {
            const float gh = a.p.normalize ? (sg[j] - gdot) / den : sg[j];
            sg[j] = gh;
            qtot += (double)(gh * sw[j]);
        }
    }
    const double qpre = wave_excl_scan_d(qtot, lane);
    double qall = qtot;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qall += __shfl_xor(qall, o, WAVE);
    // suffix_excl(j) = qall - prefix_incl(j)
    double qrun = qpre;
    float gb = 0.f, gm = 0.f, gs = 0.f;  // partial scalar gradients (effective beta, mean, scale)
    const float Ec = expf(-fabsf(a.p.cutoff - mean) / beta);
    const float ac = a.p.cutoff - mean;
    const float Fcut = cdf_cut / scale;
    for (int i = 0; i < C; ++i) {
        const int j = j0 + i;
        if (j < S) {
            const float gh = sg[j];
            qrun += (double)(gh * sw[j]);
            const float suffix = (float)(qall - qrun);
            float gc = 0.f;
            if (j < L && sact[j] != 0.f) {
                const float e = se[j];
                const float de = gh * sT[j] * expf(-e) - suffix;
                const float dsig = sdl[j] * de + (a.d_sigma ? a.d_sigma[(size_t)ray * S + j] : 0.f);
                const float x = -sc[j];
                const float ax = x - mean;
                const float E = expf(-fabsf(ax) / beta);
                const float sgx = (ax > 0.f) ? 1.f : ((ax < 0.f) ? -1.f : 0.f);
                const float Fx = 0.5f + 0.5f * sgx * (1.0f - E);
                gc = -dsig * scale * 0.5f * E / beta;                                     // d sigma / d c = -d sigma / d x
                gs += dsig * (Fx - Fcut);
                gm += dsig * scale * (-0.5f * E / beta + 0.5f * Ec / beta);
                gb += dsig * scale * (-0.5f * ax * E + 0.5f * ac * Ec) / (beta * beta);
            }
            sg[j] = gc;
        }
    }
    __builtin_amdgcn_wave_barrier();
    gb_out = wave_sum(gb); gm_out = wave_sum(gm); gs_out = wave_sum(gs);
    // gather dL/d u_j from every cosine it takes part in, then project through the normalisation
    for (int j = lane; j < S; j += WAVE) {
        float du[3] = {0.f, 0.f, 0.f};
        // as centre
        if (j < L) {
            const float gcj = sg[j];
            if (gcj != 0.f) {
                if (j >= lo && j < hi) {
                    for (int t = 1; t <= start - 1; ++t) { const float* un = su + (j + t) * 3; du[0] += coef * gcj * un[0]; du[1] += coef * gcj * un[1]; du[2] += coef * gcj * un[2]; }
                    for (int t = 1; t <= start - 2; ++t) { const float* un = su + (j - t) * 3; du[0] += coef * gcj * un[0]; du[1] += coef * gcj * un[1]; du[2] += coef * gcj * un[2]; }
                } else {
                    const float* un = su + (j + 1) * 3;
                    du[0] += gcj * un[0]; du[1] += gcj * un[1]; du[2] += gcj * un[2];
                }
            }
        }
        // as neighbour of centre i
        for (int i = max(0, j - (start - 1)); i <= min(L - 1, j + (start - 2)); ++i) {
            if (i == j) continue;
            const float gci = sg[i];
            if (gci == 0.f) continue;
            const bool interior = (i >= lo && i < hi);
            const int off = j - i;
            float wgt_ij = 0.f;
            if (interior) { if ((off >= 1 && off <= start - 1) || (off <= -1 && off >= -(start - 2))) wgt_ij = coef; }
            else if (off == 1) wgt_ij = 1.f;
            if (wgt_ij != 0.f) { const float* ui = su + i * 3; du[0] += wgt_ij * gci * ui[0]; du[1] += wgt_ij * gci * ui[1]; du[2] += wgt_ij * gci * ui[2]; }
        }
        const float* uj = su + j * 3;
        const float inv = sinv[j];
        float dn[3];
        if (inv >= 1e8f * 0.999f) {  // |n| clamped to eps: u = n / eps
            dn[0] = du[0] * inv; dn[1] = du[1] * inv; dn[2] = du[2] * inv;
        } else {
            const float pr = (du[0] * uj[0] + du[1] * uj[1]) + du[2] * uj[2];
            dn[0] = (du[0] - pr * uj[0]) * inv; dn[1] = (du[1] - pr * uj[1]) * inv; dn[2] = (du[2] - pr * uj[2]) * inv;
        }
        float* o = a.d_normals + ((size_t)ray * S + j) * 3;
        o[0] += dn[0]; o[1] += dn[1]; o[2] += dn[2];
    }
}

// blockDim.x / 64 rays per workgroup (one wave each).  The gradients of the raw density scalars (clamped parameters: no gradient
// outside their bounds, density_functions.py:129-204) are summed over the workgroup's rays in LDS and leave as ONE atomicAdd per
// scalar and workgroup: one per ray was 12 288 atomics on three words of one line at 4 096 rays, and operations on one address
// serialise at the memory side (~10 ns each) — most of this kernel's 170 us.
__global__ __launch_bounds__(1024) void vfn_density_bwd_kernel(const DensityBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float bsm[];
    __shared__ float s_red[16][3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const int ray = blockIdx.x * waves + wv;
    float gb = 0.f, gm = 0.f, gs = 0.f;
    if (ray < a.p.n_rays) density_bwd_ray(a, bsm, wv, lane, ray, gb, gm, gs);
    if (!a.d_scalars) return;
    if (lane == 0) { s_red[wv][0] = gb; s_red[wv][1] = gm; s_red[wv][2] = gs; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float v = 0.f;
        for (int w = 0; w < waves; ++w) v += s_red[w][threadIdx.x];
        const float braw = a.scalars[0], mraw = a.scalars[1], sraw = a.scalars[2];
        if (threadIdx.x == 0 && braw >= a.p.beta_min && braw <= a.p.beta_max) atomicAdd(a.d_scalars + 0, v);
        if (threadIdx.x == 1 && mraw >= a.p.mean_min && mraw <= a.p.mean_max) atomicAdd(a.d_scalars + 1, v);
        if (threadIdx.x == 2 && fabsf(sraw) >= a.p.scale_min) atomicAdd(a.d_scalars + 2, (sraw >= 0.f) ? v : -v);
    }
}

// ------------------------------------------------------------------------------------------------
// K3b: range fine sampler, one wave per ray   (models/samplers/ray_sampler.py:264-302, :77-78)
// ------------------------------------------------------------------------------------------------
struct FineArgs {
    vfn_fine_params p;
    const float* z_coarse;
    const long long* argmax;
    const float* directions;
    const float* cam_loc;
    const float* far_per_ray;
    const float* u_fine;
    const float* u_add;
    float* z_vals;
    float* points;
    // optional (vfn_range_fine_sample_indexed): where each sorted sample comes from, and the new samples on their own
    int* src;            // [N,S_t]: coarse sample j of the ray -> ray*S_c + j; new sample k -> new_row0 + ray*N_f + k
    float* new_points;   // [N,N_f,3] in generation order
    int* dst;            // the inverse: dst[src[i]] = i (sorted position of every stored sample), or NULL
    int new_row0;        // first row of the new samples in the caller's row numbering (N*S_c, or rounded up)
    // vfn_render_fwd: draws from the Philox stream instead of u_fine / u_add (elements fine_base / add_base + ray * N_f + k)
    int gen_fine, gen_add;
    long long fine_base, add_base;
    PhiloxStream ps;
};

// one ray, one wave; `sv` = St * 2 floats of LDS owned by the wave; imax = first maximum of the proposal weights
__device__ __forceinline__ void fine_ray(const FineArgs& a, float* sv, int ray, int lane, long long imax) {
    const int Sc = a.p.n_coarse, Nf = a.p.n_fine, St = Sc + Nf;
    float* so = sv + St;                    // sorted values [St]
    const float* zc = a.z_coarse + (size_t)ray * Sc;
    for (int j = lane; j < Sc; j += WAVE) sv[j] = zc[j];
    const float near = a.p.near;
    const float far = a.far_per_ray ? a.far_per_ray[ray] : a.p.far;
    if (imax > 0) {
        const float zstar = zc[imax];
        const float base = zstar - a.p.half_range;
        const float step = a.p.window_step;
        for (int k = lane; k < Nf; k += WAVE) {
            float z = base + step * (float)k;
            if (a.u_fine || a.gen_fine) {
                const float zl = base + step * (float)(k - 1);
                const float zu = base + step * (float)(k + 1);
                const float upper = (k < Nf - 1) ? 0.5f * (zu + z) : z;
                const float lower = (k > 0) ? 0.5f * (z + zl) : z;
                z = lower + (upper - lower) * (a.gen_fine ? philox_uniform(a.ps, a.fine_base + (long long)ray * Nf + k) : a.u_fine[(size_t)ray * Nf + k]);
            }
            sv[Sc + k] = z;
        }
    } else {
        const float span = a.far_per_ray ? (far - near) : a.p.span;
        for (int k = lane; k < Nf; k += WAVE)
            sv[Sc + k] = (a.gen_add ? philox_uniform(a.ps, a.add_base + (long long)ray * Nf + k) : a.u_add[(size_t)ray * Nf + k]) * span + near;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // rank sort: position = #smaller + #equal-with-lower-index (stable)
    for (int i = lane; i < St; i += WAVE) {
        const float v = sv[i];
        int rank = 0;
        for (int j = 0; j < St; ++j) {
            const float o = sv[j];
            rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
        so[rank] = v;
        const int row = i < Sc ? ray * Sc + i : a.new_row0 + ray * Nf + (i - Sc);
        if (a.src) a.src[(size_t)ray * St + rank] = row;
        if (a.dst) a.dst[row] = ray * St + rank;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float dx = a.directions[(size_t)ray * 3 + 0], dy = a.directions[(size_t)ray * 3 + 1], dz = a.directions[(size_t)ray * 3 + 2];
    const float ox = a.cam_loc[(size_t)ray * 3 + 0], oy = a.cam_loc[(size_t)ray * 3 + 1], oz = a.cam_loc[(size_t)ray * 3 + 2];
    for (int j = lane; j < St; j += WAVE) {
        const float z = so[j];
        const size_t o = (size_t)ray * St + j;
        a.z_vals[o] = z;
        a.points[o * 3 + 0] = ox + z * dx;
        a.points[o * 3 + 1] = oy + z * dy;
        a.points[o * 3 + 2] = oz + z * dz;
    }
    if (a.new_points) {
        for (int k = lane; k < Nf; k += WAVE) {
            const float z = sv[Sc + k];
            const size_t o = (size_t)ray * Nf + k;
            a.new_points[o * 3 + 0] = ox + z * dx;
            a.new_points[o * 3 + 1] = oy + z * dy;
            a.new_points[o * 3 + 2] = oz + z * dz;
        }
    }
}

__global__ __launch_bounds__(256) void vfn_fine_kernel(const FineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= a.p.n_rays) return;
    fine_ray(a, fsm + (size_t)wv * (a.p.n_coarse + a.p.n_fine) * 2, ray, lane, a.argmax[ray]);
}

// proposal weights -> argmax -> fine samples of the same ray in ONE launch (vfn_render_fwd): the wave that finds the first
// maximum of its ray's weights goes on to place the ray's fine samples; same device functions as the two stand-alone kernels
__global__ __launch_bounds__(256) void vfn_density_fine_kernel(const DensityArgs d, const FineArgs f) {
    extern __shared__ __attribute__((aligned(16))) float dfm[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= d.p.n_rays) return;
    const int per_wave = max(d.p.n_samples * 5, (f.p.n_coarse + f.p.n_fine) * 2);
    float* lds = dfm + (size_t)wv * per_wave;
    const int imax = density_ray(d, lds, ray, lane);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    fine_ray(f, lds, ray, lane, (long long)imax);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 uniforms in [0,1)
// ------------------------------------------------------------------------------------------------
__global__ void vfn_uniform_kernel(float* out, long long n, unsigned long long seed, unsigned long long offset) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one Philox block per 4 outputs
    const long long base = g * 4;
    if (base >= n) return;
    const unsigned long long ctr = offset + (unsigned long long)g;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < n) out[base + i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// Supervision points: uniform samples in a spherical shell around the scene centroid + their radial ground-truth
// directions (models/samplers/sampler.py:160-193 SphereSampler.sample; models/helpers/functions.py:100-135
// sample_border_points / sample_center_points, which the trainer calls with numpy on the host every step).
//   phi = 2 pi u0, cos(theta) = 2 u1 - 1, r = cbrt(u2) (r_max - r_min) + r_min,
//   p = c + r (sin(theta) cos(phi), sin(theta) sin(phi), cos(theta)),   gt = normalize(+-(p - c))  (eps 1e-12)
// ------------------------------------------------------------------------------------------------
struct ShellArgs {
    const float* u;          // [n,3] explicit uniforms, or NULL -> Philox(seed, offset + sample)
    const float* centroid;   // [3]
    float* points;           // [n,3]
    float* gt;               // [n,3]
    long long n;
    float r_min, r_max;
    int inward;              // 1: gt points at the centroid (border supervision); 0: away from it (centre supervision)
    unsigned long long seed, offset;
};

__global__ void vfn_sphere_shell_kernel(const ShellArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float u0, u1, u2;
    if (a.u) { u0 = a.u[i * 3 + 0]; u1 = a.u[i * 3 + 1]; u2 = a.u[i * 3 + 2]; }
    else {
        const unsigned long long ctr = a.offset + (unsigned long long)i;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k[2] = {(uint32_t)a.seed, (uint32_t)(a.seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; ++r) philox_round(c, k);
        u0 = (float)(c[0] >> 8) * (1.0f / 16777216.0f);
        u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
        u2 = (float)(c[2] >> 8) * (1.0f / 16777216.0f);
    }
    const float phi = 6.283185307179586f * u0;
    const float ct = 2.0f * u1 - 1.0f;
    const float st = sqrtf(fmaxf(0.f, 1.0f - ct * ct));
    const float r = cbrtf(u2) * (a.r_max - a.r_min) + a.r_min;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    const float cx = a.centroid[0], cy = a.centroid[1], cz = a.centroid[2];
    const float px = r * st * cp + cx, py = r * st * sp + cy, pz = r * ct + cz;
    a.points[i * 3 + 0] = px; a.points[i * 3 + 1] = py; a.points[i * 3 + 2] = pz;
    float dx = px - cx, dy = py - cy, dz = pz - cz;
    if (a.inward) { dx = -dx; dy = -dy; dz = -dz; }
    const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    a.gt[i * 3 + 0] = dx * inv; a.gt[i * 3 + 1] = dy * inv; a.gt[i * 3 + 2] = dz * inv;
}


// out_a[index[r]] = a[r], out_b[index[r]] = b[r] for 3-float rows (negative index: skipped): per-sample results computed in
// storage order move to their positions among the sorted samples
__global__ __launch_bounds__(256) void vfn_scatter_rows3_kernel(const float* a, const float* b, const int* index, long long n,
                                                                 float* out_a, float* out_b) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int d = index[r];
    if (d < 0) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out_a[(long long)d * 3 + c] = a[r * 3 + c];
        if (b) out_b[(long long)d * 3 + c] = b[r * 3 + c];
    }
}

// ------------------------------------------------------------------------------------------------
// The samplers as stand-alone calls (RaySampler.sample / get_z_vals through the sampler objects,
// models/samplers/ray_sampler.py:49-80,113-142): proposal depths + points for GIVEN directions / origins, and the
// first-maximum index of every row of a weight matrix (torch.argmax's tie rule, :277).
// ------------------------------------------------------------------------------------------------
struct UniformSampleArgs {
    const float* directions;   // [N,3] un-normalised (Q7)
    const float* cam_loc;      // [N,3]
    const float* t_vals;       // [S]
    const float* far_per_ray;  // [N] or NULL
    const float* u;            // [N,S] or NULL
    float* z_vals;             // [N,S]
    float* points;             // [N,S,3] or NULL
    int n_rays, n_samples;
    float near, far;
};

__global__ __launch_bounds__(256) void vfn_uniform_sample_kernel(const UniformSampleArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = a.n_samples;
    if (idx >= (long long)a.n_rays * S) return;
    const int r = (int)(idx / S), s = (int)(idx - (long long)r * S);
    const float near = a.near, far = a.far_per_ray ? a.far_per_ray[r] : a.far;
    float z = coarse_z(near, far, a.t_vals[s]);
    if (a.u) {
        const float zl = (s > 0) ? coarse_z(near, far, a.t_vals[s - 1]) : z;
        const float zu = (s < S - 1) ? coarse_z(near, far, a.t_vals[s + 1]) : z;
        const float upper = (s < S - 1) ? 0.5f * (zu + z) : z;
        const float lower = (s > 0) ? 0.5f * (z + zl) : z;
        z = lower + (upper - lower) * a.u[idx];
    }
    a.z_vals[idx] = z;
    if (a.points) {
        a.points[idx * 3 + 0] = a.cam_loc[(size_t)r * 3 + 0] + z * a.directions[(size_t)r * 3 + 0];
        a.points[idx * 3 + 1] = a.cam_loc[(size_t)r * 3 + 1] + z * a.directions[(size_t)r * 3 + 1];
        a.points[idx * 3 + 2] = a.cam_loc[(size_t)r * 3 + 2] + z * a.directions[(size_t)r * 3 + 2];
    }
}

__global__ __launch_bounds__(256) void vfn_rows_argmax_kernel(const float* w, int n_rows, int n_cols, long long* out) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * RAYS_PER_BLOCK + wv;
    if (row >= n_rows) return;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int j = lane; j < n_cols; j += WAVE) {
        const float v = w[(size_t)row * n_cols + j];
        if (argmax_takes(v, j, best, besti)) { best = v; besti = j; }   // first maximum; a NaN wins, like torch.argmax
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(besti, o, WAVE);
        if (argmax_takes(ob, oi, best, besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) out[row] = (besti == 0x7fffffff) ? 0 : besti;
}

// RaySampler.sample with additional_depths (ray_sampler.py:69-73): the sampler's depths and the extra ones concatenated, sorted
// ascending per ray (torch.sort: a NaN sorts last), the points recomputed.  One wave per ray; a bitonic network over the ray's
// depths in LDS, padded to a power of two with NaNs (they sort behind everything, a real +inf included).
__global__ __launch_bounds__(256) void vfn_merge_sort_depths_kernel(const float* z, const float* extra, int n_rays, int s, int e, int p2,
                                                                    const float* directions, const float* cam_loc, float* z_out, float* points) {
    extern __shared__ float sbuf[];                          // RAYS_PER_BLOCK x p2
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool live = ray < n_rays;
    float* buf = sbuf + wv * p2;
    const int tot = s + e;
    for (int i = lane; i < p2; i += WAVE)
        buf[i] = !live || i >= tot ? __builtin_nanf("") : (i < s ? z[(size_t)ray * s + i] : extra[(size_t)ray * e + (i - s)]);
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < p2; i += WAVE) {
                const int q = i ^ j;
                if (q > i) {
                    const float a = buf[i], b = buf[q];
                    const bool a_gt_b = (a != a && b == b) || a > b;          // NaN above every number, like torch.sort
                    const bool up = (i & k) == 0;
                    if (a_gt_b == up && (a_gt_b || (b != b && a == a) || b > a)) { buf[i] = b; buf[q] = a; }
                }
            }
            __syncthreads();
        }
    if (!live) return;
    for (int i = lane; i < tot; i += WAVE) {
        const float zz = buf[i];
        const size_t idx = (size_t)ray * tot + i;
        z_out[idx] = zz;
        if (points) {
            points[idx * 3 + 0] = cam_loc[(size_t)ray * 3 + 0] + zz * directions[(size_t)ray * 3 + 0];
            points[idx * 3 + 1] = cam_loc[(size_t)ray * 3 + 1] + zz * directions[(size_t)ray * 3 + 1];
            points[idx * 3 + 2] = cam_loc[(size_t)ray * 3 + 2] + zz * directions[(size_t)ray * 3 + 2];
        }
    }
}

}  // namespace

extern "C" int vfn_uniform_sample(int32_t n_rays, int32_t n_samples, float near, float far, const float* directions,
                                  const float* cam_loc, const float* t_vals, const float* far_per_ray, const float* u,
                                  float* z_vals, float* points, void* stream) {
    if (n_rays == 0) return VFN_OK;
    VFN_REQUIRE(n_rays > 0 && n_samples >= 1, "vfn_uniform_sample: bad sizes (n_rays=%d, n_samples=%d)", n_rays, n_samples);
    VFN_REQUIRE(t_vals && z_vals && (!points || (directions && cam_loc)), "vfn_uniform_sample: NULL argument");
    UniformSampleArgs a{directions, cam_loc, t_vals, far_per_ray, u, z_vals, points, n_rays, n_samples, near, far};
    const long long total = (long long)n_rays * n_samples;
    hipLaunchKernelGGL(vfn_uniform_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_uniform_sample");
}

extern "C" int vfn_merge_sort_depths(const float* z_vals, const float* extra, int32_t n_rays, int32_t n_samples, int32_t n_extra,
                                     const float* directions, const float* cam_loc, float* z_out, float* points, void* stream) {
    if (n_rays == 0) return VFN_OK;
    VFN_REQUIRE(n_rays > 0 && n_samples >= 0 && n_extra >= 0 && n_samples + n_extra >= 1 && n_samples + n_extra <= 2048,
                "vfn_merge_sort_depths: bad sizes (n_rays=%d, n_samples=%d, n_extra=%d; at most 2048 depths per ray)", n_rays, n_samples, n_extra);
    VFN_REQUIRE((z_vals || n_samples == 0) && (extra || n_extra == 0) && z_out && (!points || (directions && cam_loc)),
                "vfn_merge_sort_depths: NULL argument");
    int p2 = 2;
    while (p2 < n_samples + n_extra) p2 <<= 1;
    hipLaunchKernelGGL(vfn_merge_sort_depths_kernel, dim3((unsigned)((n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK)), dim3(256),
                       (size_t)RAYS_PER_BLOCK * p2 * sizeof(float), (hipStream_t)stream, z_vals, extra, (int)n_rays, (int)n_samples, (int)n_extra, p2,
                       directions, cam_loc, z_out, points);
    return vfn_check_launch("vfn_merge_sort_depths");
}

extern "C" int vfn_rows_argmax(const float* w, int32_t n_rows, int32_t n_cols, int64_t* out, void* stream) {
    if (n_rows == 0) return VFN_OK;
    VFN_REQUIRE(w && out && n_rows > 0 && n_cols >= 1, "vfn_rows_argmax: bad argument");
    hipLaunchKernelGGL(vfn_rows_argmax_kernel, dim3((unsigned)((n_rows + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK)), dim3(256), 0,
                       (hipStream_t)stream, w, (int)n_rows, (int)n_cols, (long long*)out);
    return vfn_check_launch("vfn_rows_argmax");
}

extern "C" int vfn_raygen_uniform(const vfn_raygen_params* p, const float* uv, const float* pose, const float* intrinsics,
                                  const float* t_vals, const float* far_per_ray, const float* u_coarse, float* directions,
                                  float* ray_dirs, float* cam_loc, float* z_vals, float* points, void* stream) {
    if (p && p->n_rays == 0) return VFN_OK;   // empty batch: nothing to launch (device pointers may be NULL)
    VFN_REQUIRE(p && uv && pose && intrinsics && t_vals && directions && ray_dirs && cam_loc && z_vals && points,
                "vfn_raygen_uniform: NULL argument");
    VFN_REQUIRE(p->n_rays >= 0 && p->n_samples >= 1, "vfn_raygen_uniform: bad sizes (n_rays=%d, n_samples=%d)", p->n_rays,
                p->n_samples);
    if (p->n_rays == 0) return VFN_OK;
    RaygenArgs a{*p, uv, pose, intrinsics, t_vals, far_per_ray, u_coarse, directions, ray_dirs, cam_loc, z_vals, points};
    const unsigned blocks = (unsigned)((p->n_rays + K1_RAYS - 1) / K1_RAYS);
    hipLaunchKernelGGL(vfn_raygen_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_raygen_uniform");
}

extern "C" int vfn_ray_density_weights(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                                       const float* z_vals, const float* density_scalars, const float* colors, float* sigma,
                                       float* weights, int64_t* argmax, float* rgb, float* depth, void* stream) {
    if (p && p->n_rays <= 0) return VFN_OK;
    VFN_REQUIRE(p && normals && ray_dirs && z_vals && density_scalars, "vfn_ray_density_weights: NULL argument");
    VFN_REQUIRE(p->n_samples >= 2 && p->n_samples <= MAX_SAMPLES, "vfn_ray_density_weights: n_samples=%d outside [2,%d]",
                p->n_samples, MAX_SAMPLES);
    VFN_REQUIRE(p->n_window >= 1, "vfn_ray_density_weights: n_window must be >= 1");
    VFN_REQUIRE(!colors || (rgb && depth), "vfn_ray_density_weights: colors given but rgb/depth NULL");
    if (p->n_rays <= 0) return VFN_OK;
    DensityArgs a{*p, normals, ray_dirs, z_vals, density_scalars, colors, sigma, weights, (long long*)argmax, rgb, depth};
    const unsigned blocks = (unsigned)((p->n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    const size_t shmem = (size_t)RAYS_PER_BLOCK * p->n_samples * 5 * sizeof(float);
    hipLaunchKernelGGL(vfn_density_kernel, dim3(blocks), dim3(256), shmem, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_ray_density_weights");
}

extern "C" int vfn_range_fine_sample(const vfn_fine_params* p, const float* z_coarse, const int64_t* argmax,
                                     const float* directions, const float* cam_loc, const float* far_per_ray,
                                     const float* u_fine, const float* u_add, float* z_vals, float* points, void* stream) {
    return vfn_range_fine_sample_indexed(p, z_coarse, argmax, directions, cam_loc, far_per_ray, u_fine, u_add, z_vals, points,
                                         nullptr, nullptr, nullptr, p ? (int64_t)p->n_rays * p->n_coarse : 0, stream);
}

extern "C" int vfn_range_fine_sample_indexed(const vfn_fine_params* p, const float* z_coarse, const int64_t* argmax,
                                             const float* directions, const float* cam_loc, const float* far_per_ray,
                                             const float* u_fine, const float* u_add, float* z_vals, float* points,
                                             int32_t* src, float* new_points, int32_t* dst, int64_t new_row0, void* stream) {
    if (p && p->n_rays <= 0) return VFN_OK;
    VFN_REQUIRE(p && z_coarse && argmax && directions && cam_loc && u_add && z_vals && points,
                "vfn_range_fine_sample: NULL argument (u_add is always required, ray_sampler.py:292)");
    VFN_REQUIRE(p->n_coarse >= 1 && p->n_fine >= 2 && p->n_coarse + p->n_fine <= MAX_SAMPLES,
                "vfn_range_fine_sample: bad sizes (n_coarse=%d, n_fine=%d)", p->n_coarse, p->n_fine);
    VFN_REQUIRE((long long)p->n_rays * (p->n_coarse + p->n_fine) < (1ll << 31), "vfn_range_fine_sample: more than 2^31 samples");
    VFN_REQUIRE(new_row0 >= (long long)p->n_rays * p->n_coarse && new_row0 + (long long)p->n_rays * p->n_fine < (1ll << 31),
                "vfn_range_fine_sample_indexed: new_row0=%lld must not overlap the proposal rows", (long long)new_row0);
    FineArgs a{*p, z_coarse, (const long long*)argmax, directions, cam_loc, far_per_ray, u_fine, u_add, z_vals, points, src, new_points,
               dst, (int)new_row0};
    const unsigned blocks = (unsigned)((p->n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    const size_t shmem = (size_t)RAYS_PER_BLOCK * (p->n_coarse + p->n_fine) * 2 * sizeof(float);
    hipLaunchKernelGGL(vfn_fine_kernel, dim3(blocks), dim3(256), shmem, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_range_fine_sample");
}

// ------------------------------------------------------------------------------------------------
// Launches used by vfn_render_fwd only (csrc/vfn_render.hip; declared in vfn_common.h): the same kernels with the Philox draws
// generated in place, the proposal argmax and the fine sampler in one launch, and the proposal results moved to their sorted
// positions by the composite launch.  Values are those of the stand-alone entry points, bit for bit.
// ------------------------------------------------------------------------------------------------
int vfn_internal_raygen(const vfn_raygen_params* p, const float* uv, const float* pose, const float* intrinsics, const float* k_sign,
                        const float* t_vals, const float* far_per_ray, const float* u_coarse, int gen_u, long long u_base, uint64_t seed,
                        uint64_t offset, float* directions, float* ray_dirs, float* cam_loc, float* z_vals, float* points, void* stream) {
    RaygenArgs a{*p, uv, pose, intrinsics, t_vals, far_per_ray, u_coarse, directions, ray_dirs, cam_loc, z_vals, points};
    a.gen_u = gen_u; a.u_base = u_base; a.ps = PhiloxStream{seed, offset}; a.k_sign = k_sign;
    const unsigned blocks = (unsigned)((p->n_rays + K1_RAYS - 1) / K1_RAYS);
    hipLaunchKernelGGL(vfn_raygen_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_render_fwd (rays)");
}

int vfn_internal_density_fine(const vfn_density_params* dp, const float* normals_c, const float* ray_dirs, const float* z_c,
                              const float* density_scalars, const vfn_fine_params* fp, const float* directions, const float* cam_loc,
                              const float* far_per_ray, const float* u_fine, const float* u_add, int gen_fine, int gen_add,
                              long long fine_base, long long add_base, uint64_t seed, uint64_t offset, float* z_vals, float* points,
                              int32_t* src, float* new_points, int32_t* dst, int64_t new_row0, void* stream) {
    VFN_REQUIRE(dp->n_samples >= 2 && dp->n_samples <= MAX_SAMPLES && fp->n_coarse + fp->n_fine <= MAX_SAMPLES && fp->n_fine >= 2 &&
                dp->n_window >= 1, "vfn_render_fwd: sample counts outside what the per-ray kernels support (n_coarse=%d, n_fine=%d)",
                fp->n_coarse, fp->n_fine);
    VFN_REQUIRE((long long)fp->n_rays * (fp->n_coarse + fp->n_fine) < (1ll << 31), "vfn_render_fwd: more than 2^31 samples");
    DensityArgs d{*dp, normals_c, ray_dirs, z_c, density_scalars, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    FineArgs f{*fp, z_c, nullptr, directions, cam_loc, far_per_ray, u_fine, u_add, z_vals, points, src, new_points, dst, (int)new_row0};
    f.gen_fine = gen_fine; f.gen_add = gen_add; f.fine_base = fine_base; f.add_base = add_base; f.ps = PhiloxStream{seed, offset};
    const unsigned blocks = (unsigned)((dp->n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    const int per_wave = dp->n_samples * 5 > (fp->n_coarse + fp->n_fine) * 2 ? dp->n_samples * 5 : (fp->n_coarse + fp->n_fine) * 2;
    hipLaunchKernelGGL(vfn_density_fine_kernel, dim3(blocks), dim3(256), (size_t)RAYS_PER_BLOCK * per_wave * sizeof(float),
                       (hipStream_t)stream, d, f);
    return vfn_check_launch("vfn_render_fwd (proposal weights + fine sampler)");
}

int vfn_internal_composite_gather(const vfn_density_params* dp, float* normals, const float* ray_dirs, const float* z_vals,
                                  const float* density_scalars, float* colors, const int32_t* src, const float* normals_c,
                                  const float* colors_c, int64_t n_stored_c, float* weights, float* rgb, float* depth, void* stream) {
    VFN_REQUIRE(dp->n_samples >= 2 && dp->n_samples <= MAX_SAMPLES, "vfn_render_fwd: n_samples=%d outside [2,%d]", dp->n_samples, MAX_SAMPLES);
    DensityArgs a{*dp, normals, ray_dirs, z_vals, density_scalars, colors, nullptr, weights, nullptr, rgb, depth};
    a.src = src; a.normals_c = normals_c; a.colors_c = colors_c; a.n_stored_c = (int)n_stored_c;
    const unsigned blocks = (unsigned)((dp->n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(vfn_density_kernel, dim3(blocks), dim3(256), (size_t)RAYS_PER_BLOCK * dp->n_samples * 5 * sizeof(float),
                       (hipStream_t)stream, a);
    return vfn_check_launch("vfn_render_fwd (composite)");
}

// ------------------------------------------------------------------------------------------------
// Sparse colour branch (csrc/vfn_train.hip, csrc/vfn_render.hip): the samples of a batch whose weight is non-zero, compacted.
// rgb = sum_s w_s c_s needs a colour only where w_s != 0 — a few percent of the samples (a closed density ReLU, or a transmittance
// that has underflowed, make the weight EXACTLY zero).
// ------------------------------------------------------------------------------------------------
namespace {
// One wave per ray (4 rays per workgroup).  Pass 1 counts, a one-workgroup scan turns the counts into offsets (ray order: the
// compacted list is deterministic), pass 2 writes, per selected sample, its sorted index, its point and its ray's direction.
//
// The predicate.  Forward-only callers (sigma == NULL) select w > 0: a colour enters a render through w c only.  A TRAINING step also
// needs a sample's colour where its weight is zero but the weight's DERIVATIVE is not: d w_j / d sigma_j = T_j delta_j exp(-e_j) survives
// an alpha that has underflowed (1 - exp(-e) == 0 for e < 6e-8 while sigma > 0), and the dense step's (d rgb . c_j) term of d sigma_j
// with it.  With sigma and z given the predicate is therefore  w > 0  OR  (sigma > 0 AND delta > 0 AND T > 0)  — every sample whose
// colour can reach any gradient or output.  T_j = exp(-sum_{i<j} e_i) is tested through its exponent (float exp underflows to zero past
// 103.98; the sum is re-accumulated here in fp64, the bound 110 errs on the side of selecting).
__device__ __forceinline__ bool sel_predicate(const float* w, const float* sigma, const float* z, int ray, int S, int j, int lane, double& carry) {
    const size_t i = (size_t)ray * S + j;
    const bool in = j < S;
    const bool pos = in && w[i] > 0.f;
    if (!sigma) return pos;
    const float sg = in ? sigma[i] : 0.f;
    const float delta = !in ? 0.f : (j < S - 1 ? z[i + 1] - z[i] : 1e10f);
    const double e = (double)(sg * delta);
    double incl = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const double before = carry + (incl - e);
    carry += __shfl(incl, 63, 64);
    return pos || (in && sg > 0.f && delta > 0.f && before < 110.0);
}

__global__ void vfn_sel_count_kernel(const float* w, const float* sigma, const float* z, int n_rays, int S, int32_t* cnt) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    int c = 0;
    double carry = 0.0;
    for (int j0 = 0; j0 < S; j0 += 64) c += sel_predicate(w, sigma, z, ray, S, j0 + lane, lane, carry) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) cnt[ray] = c;
}

__global__ __launch_bounds__(256) void vfn_sel_scan_kernel(const int32_t* cnt, int n, int32_t* off, int32_t* k_dev) {
    __shared__ int s_sum[256];
    const int t = threadIdx.x;
    const int per = (n + 255) / 256, lo = t * per, hi = min(n, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    s_sum[t] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {          // inclusive scan of the 256 chunk sums
        const int v = t >= o ? s_sum[t - o] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int run = s_sum[t] - sum;                    // exclusive
    for (int i = lo; i < hi; ++i) { off[i] = run; run += cnt[i]; }
    if (t == 255) k_dev[0] = s_sum[255];
}

__global__ void vfn_sel_compact_kernel(const float* w, const float* sigma, const float* z, int n_rays, int S, const int32_t* off, const float* points,
                                       const float* ray_dirs, int32_t* sel_sorted, float* pts_sel, float* dirs_sel) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    int base = off[ray];
    const float dx = ray_dirs[(size_t)ray * 3 + 0], dy = ray_dirs[(size_t)ray * 3 + 1], dz = ray_dirs[(size_t)ray * 3 + 2];
    double carry = 0.0;
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        const bool sel = sel_predicate(w, sigma, z, ray, S, j, lane, carry);
        const unsigned long long mask = __ballot(sel);
        if (sel) {
            const int k = base + __popcll(mask & ((1ull << lane) - 1ull));
            const size_t i = (size_t)ray * S + j;
            sel_sorted[k] = (int32_t)i;
            pts_sel[(size_t)k * 3 + 0] = points[i * 3 + 0]; pts_sel[(size_t)k * 3 + 1] = points[i * 3 + 1]; pts_sel[(size_t)k * 3 + 2] = points[i * 3 + 2];
            dirs_sel[(size_t)k * 3 + 0] = dx; dirs_sel[(size_t)k * 3 + 1] = dy; dirs_sel[(size_t)k * 3 + 2] = dz;
        }
        base += __popcll(mask);
    }
}

// out[index[k]] = a[k] (scatter) or out[k] = a[index[k]] (gather) for the k < *k_dev selected rows of [.,3] arrays
__global__ void vfn_sel_rows3_kernel(const float* a, const int32_t* index, const int32_t* k_dev, float* out, int gather) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)*k_dev) return;
    const size_t i = (size_t)index[k];
    const size_t from = gather ? i : (size_t)k, to = gather ? (size_t)k : i;
    out[to * 3 + 0] = a[from * 3 + 0]; out[to * 3 + 1] = a[from * 3 + 1]; out[to * 3 + 2] = a[from * 3 + 2];
}

}  // namespace

// weights[N,S] (sorted order), points[N,S,3], ray_dirs[N,3] -> k_dev[0] = number K of selected samples (device memory; the host never
// learns it), and for k < K, in ray order: sel_sorted[k] = the sample's index ray * S + j, pts_sel[k], dirs_sel[k].  cnt / off: [N] scratch.
// sigma == z_vals == NULL: the samples with w > 0 (forward-only renders); with sigma[N,S] and z_vals[N,S]: also those whose weight is
// zero by an underflowed alpha only (sel_predicate above: what a training step's gradients need).
int vfn_internal_select_positive(const float* weights, const float* sigma, const float* z_vals, int n_rays, int n_samples, const float* points,
                                 const float* ray_dirs, int32_t* cnt, int32_t* off, int32_t* k_dev, int32_t* sel_sorted, float* pts_sel,
                                 float* dirs_sel, void* stream) {
    VFN_REQUIRE((sigma == nullptr) == (z_vals == nullptr), "vfn_internal_select_positive: sigma and z_vals come together");
    VFN_REQUIRE(weights && points && ray_dirs && cnt && off && k_dev && sel_sorted && pts_sel && dirs_sel && n_rays > 0 && n_samples > 0,
                "vfn_internal_select_positive: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n_rays + 3) / 4);
    hipLaunchKernelGGL(vfn_sel_count_kernel, dim3(blocks), dim3(256), 0, s, weights, sigma, z_vals, n_rays, n_samples, cnt);
    hipLaunchKernelGGL(vfn_sel_scan_kernel, dim3(1), dim3(256), 0, s, cnt, n_rays, off, k_dev);
    hipLaunchKernelGGL(vfn_sel_compact_kernel, dim3(blocks), dim3(256), 0, s, weights, sigma, z_vals, n_rays, n_samples, off, points, ray_dirs,
                       sel_sorted, pts_sel, dirs_sel);
    return vfn_check_launch("sample selection (w > 0)");
}

extern "C" int vfn_select_samples(const float* weights, const float* sigma, const float* z_vals, int32_t n_rays, int32_t n_samples, const float* points,
                                  const float* ray_dirs, int32_t* scratch, int32_t* count, int32_t* index, float* points_sel, float* dirs_sel,
                                  void* stream) {
    VFN_REQUIRE(scratch && count, "vfn_select_samples: NULL scratch / count");
    return vfn_internal_select_positive(weights, sigma, z_vals, n_rays, n_samples, points, ray_dirs, scratch, scratch + n_rays, count, index, points_sel,
                                        dirs_sel, stream);
}

// gather != 0: out[k] = a[index[k]]; else out[index[k]] = a[k]; for the k < *k_dev (<= capacity) selected rows of [., 3] arrays
int vfn_internal_rows3_by_index(const float* a, const int32_t* index, const int32_t* k_dev, int64_t capacity, float* out, int gather, void* stream) {
    VFN_REQUIRE(a && index && k_dev && out && capacity >= 0, "vfn_internal_rows3_by_index: bad argument");
    if (capacity == 0) return VFN_OK;
    hipLaunchKernelGGL(vfn_sel_rows3_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, index, k_dev, out, gather);
    return vfn_check_launch("selected rows scatter / gather");
}

extern "C" int vfn_scatter_rows3(const float* a, const float* b, const int32_t* index, int64_t n_rows, float* out_a, float* out_b,
                                 void* stream) {
    VFN_REQUIRE(n_rows >= 0 && (n_rows == 0 || (a && index && out_a && (!b || out_b))), "vfn_scatter_rows3: NULL argument");
    if (n_rows == 0) return VFN_OK;
    hipLaunchKernelGGL(vfn_scatter_rows3_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, index,
                       (long long)n_rows, out_a, out_b);
    return vfn_check_launch("vfn_scatter_rows3");
}

extern "C" int vfn_fill_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    VFN_REQUIRE(out || n == 0, "vfn_fill_uniform: NULL output");
    if (n <= 0) return VFN_OK;
    const long long groups = (n + 3) / 4;
    const unsigned blocks = (unsigned)((groups + 255) / 256);
    hipLaunchKernelGGL(vfn_uniform_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, (long long)n,
                       (unsigned long long)seed, (unsigned long long)offset);
    return vfn_check_launch("vfn_fill_uniform");
}

// rays (waves) per workgroup of the backward kernel: up to sixteen while their LDS (48 bytes per sample and ray) stays inside the
// 64 KiB a launch may ask for dynamically without further ado (eight rays at 128 samples)
static int bwd_rays_per_block(int n_samples) {
    int rpb = 16;
    while (rpb > 1 && (size_t)rpb * n_samples * 12 * sizeof(float) > 60 * 1024) rpb >>= 1;
    return rpb;
}

extern "C" int vfn_ray_density_weights_bwd(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                                           const float* z_vals, const float* density_scalars, const float* colors,
                                           const float* d_rgb, const float* d_depth, const float* d_weights,
                                           float* d_normals, float* d_colors, float* d_scalars, void* stream) {
    if (p && p->n_rays <= 0) return VFN_OK;
    VFN_REQUIRE(p && normals && ray_dirs && z_vals && density_scalars && d_normals, "vfn_ray_density_weights_bwd: NULL argument");
    VFN_REQUIRE(p->n_samples >= 2 && p->n_samples <= MAX_SAMPLES_BWD,
                "vfn_ray_density_weights_bwd: n_samples=%d outside [2,%d]", p->n_samples, MAX_SAMPLES_BWD);
    VFN_REQUIRE(!(d_rgb && !colors), "vfn_ray_density_weights_bwd: d_rgb given without colors");
    if (p->n_rays <= 0) return VFN_OK;
    DensityBwdArgs a{*p, normals, ray_dirs, z_vals, density_scalars, colors, d_rgb, d_depth, d_weights, nullptr, d_normals, d_colors, d_scalars};
    const int rpb = bwd_rays_per_block(p->n_samples);
    const unsigned blocks = (unsigned)((p->n_rays + rpb - 1) / rpb);
    const size_t shmem = (size_t)rpb * p->n_samples * 12 * sizeof(float);
    hipLaunchKernelGGL(vfn_density_bwd_kernel, dim3(blocks), dim3(64 * rpb), shmem, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_ray_density_weights_bwd");
}

extern "C" int vfn_ray_density_sigma_bwd(const vfn_density_params* p, const float* normals, const float* ray_dirs,
                                         const float* z_vals, const float* density_scalars, const float* d_sigma,
                                         float* d_normals, float* d_scalars, void* stream) {
    if (p && p->n_rays <= 0) return VFN_OK;
    VFN_REQUIRE(p && normals && ray_dirs && z_vals && density_scalars && d_sigma && d_normals, "vfn_ray_density_sigma_bwd: NULL argument");
    VFN_REQUIRE(p->n_samples >= 2 && p->n_samples <= MAX_SAMPLES_BWD,
                "vfn_ray_density_sigma_bwd: n_samples=%d outside [2,%d]", p->n_samples, MAX_SAMPLES_BWD);
    DensityBwdArgs a{*p, normals, ray_dirs, z_vals, density_scalars, nullptr, nullptr, nullptr, nullptr, d_sigma, d_normals, nullptr, d_scalars};
    const int rpb = bwd_rays_per_block(p->n_samples);
    const unsigned blocks = (unsigned)((p->n_rays + rpb - 1) / rpb);
    const size_t shmem = (size_t)rpb * p->n_samples * 12 * sizeof(float);
    hipLaunchKernelGGL(vfn_density_bwd_kernel, dim3(blocks), dim3(64 * rpb), shmem, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_ray_density_sigma_bwd");
}

extern "C" int vfn_sample_sphere_shell(int64_t n, float r_min, float r_max, const float* centroid, int32_t inward, const float* u,
                                       uint64_t seed, uint64_t offset, float* points, float* gt, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(centroid && points && gt, "vfn_sample_sphere_shell: NULL argument");
    VFN_REQUIRE(r_max >= r_min && r_min >= 0.f, "vfn_sample_sphere_shell: need 0 <= r_min <= r_max (got %g, %g)", (double)r_min, (double)r_max);
    ShellArgs a{u, centroid, points, gt, (long long)n, r_min, r_max, inward ? 1 : 0, (unsigned long long)seed, (unsigned long long)offset};
    hipLaunchKernelGGL(vfn_sphere_shell_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_sample_sphere_shell");
}
