// vfn_grid.hip — the dense-grid stages between the vector-field queries and the mesh triangulation:
// evaluation/utils/mc_utils.py:34-86 (extract_divergence), :107-167 (unify_direction), :170-223 (make_comb_format) and
// evaluation/utils/guassian_smoothing.py:81-97 (smooth_vf).  The reference runs them as conv3d / gather chains on CPU
// tensors of res^3 x 3 floats; here each is one HBM-bound kernel over the grid, one thread per cell, the innermost grid
// index on consecutive lanes (the 2x2x2 corner gathers of neighbouring cells overlap in L2).
//
// Grid layout: cell (i, j, k) -> flat index (i N + j) N + k; vector field [N^3, 3] row-major.  The 8 cell corners in the
// order of the reference's selection filters: (0,0,0) (0,1,0) (1,1,0) (1,0,0) (0,0,1) (0,1,1) (1,1,1) (1,0,1) as (di,dj,dk);
// corners outside the grid read as zero (the reference's zero padding).
#include <string.h>
#include "vfn_common.h"

namespace {

__device__ __constant__ int CORNER[8][3] = {{0, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}, {1, 0, 1}};

__device__ __forceinline__ void load_vec(const float* vt, long long N, int i, int j, int k, float (&v)[3]) {
    if (i < N && j < N && k < N) {
        const long long o = (((long long)i * N + j) * N + k) * 3;
        v[0] = vt[o]; v[1] = vt[o + 1]; v[2] = vt[o + 2];
    } else { v[0] = v[1] = v[2] = 0.f; }
}

// ---- divergence mask: 1 where the normalised field converges onto the cell (mc_utils.py:34-86) ----
__global__ void vfn_grid_divergence_kernel(const float* vt, float* out, int N, float threshold) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * N * N;
    if (idx >= total) return;
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((long long)N * N));
    float res = 0.f;
    if (i < N - 1 && j < N - 1 && k < N - 1) {
        const float inv3 = 1.0f / sqrtf(3.0f);
        const float face_area = (float)(1.7320508075688772 / 4.0), shape_volume = (float)(1.4142135623730951 / 3.0);
        float s = 0.f;
        // corner c of the 2x2x2 box = (a, b, cc) = (c >> 2, (c >> 1) & 1, c & 1), outward direction (2a-1, 2b-1, 2cc-1) / sqrt(3)
        for (int c = 0; c < 8; ++c) {
            const int a = c >> 2, b = (c >> 1) & 1, cc = c & 1;
            float v[3];
            load_vec(vt, N, i + a, j + b, k + cc, v);
            const float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
            const float x = (v[0] / nrm) * (a ? inv3 : -inv3) + (v[1] / nrm) * (b ? inv3 : -inv3) + (v[2] / nrm) * (cc ? inv3 : -inv3);
            s += x * fabsf(x) * face_area;
        }
        res = s / shape_volume;
    }
    out[idx] = res > threshold ? 0.f : 1.f;
}

// ---- one pass of the separable Gaussian along one axis, replicate padding (guassian_smoothing.py:81-97) ----
struct SmoothArgs { const float* in; float* out; int N; int axis; int k; float w[16]; };
__global__ void vfn_grid_smooth_kernel(const SmoothArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // voxel index
    const long long N = a.N, total = N * N * N;
    if (idx >= total) return;
    int pos[3] = {(int)(idx / (N * N)), (int)((idx / N) % N), (int)(idx % N)};
    const long long stride = a.axis == 0 ? N * N : (a.axis == 1 ? N : 1);
    const int p0 = pos[a.axis], half = a.k / 2;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int t = 0; t < a.k; ++t) {
        int q = p0 + t - half;
        q = q < 0 ? 0 : (q >= a.N ? a.N - 1 : q);
        const long long o = (idx + (long long)(q - p0) * stride) * 3;
        acc[0] += a.w[t] * a.in[o]; acc[1] += a.w[t] * a.in[o + 1]; acc[2] += a.w[t] * a.in[o + 2];
    }
    a.out[idx * 3] = acc[0]; a.out[idx * 3 + 1] = acc[1]; a.out[idx * 3 + 2] = acc[2];
}

// ---- per surface cell: the two most opposed corner vectors, and for every corner which of the two it sides with
//      (mc_utils.py:107-167) ----
__global__ void vfn_grid_unify_kernel(const float* div, const float* vt, long long* choice, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * N * N;
    if (idx >= total) return;
    long long* out = choice + idx * 8;
    if (div[idx] != 1.0f) {
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q] = 0;
        return;
    }
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((long long)N * N));
    float v[8][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) load_vec(vt, N, i + CORNER[q][0], j + CORNER[q][1], k + CORNER[q][2], v[q]);
    float best = -3.4e38f;
    int bi = 0;
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) {
            const float d = 1.0f - ((v[a][0] * v[b][0] + v[a][1] * v[b][1]) + v[a][2] * v[b][2]);
            if (d > best) { best = d; bi = a * 8 + b; }       // first maximum, as torch.argmax
        }
    const int f = bi >> 3, s = bi & 7;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float d1x = v[f][0] - v[q][0], d1y = v[f][1] - v[q][1], d1z = v[f][2] - v[q][2];
        const float d2x = v[s][0] - v[q][0], d2y = v[s][1] - v[q][1], d2z = v[s][2] - v[q][2];
        const float n1 = sqrtf(d1x * d1x + d1y * d1y + d1z * d1z), n2 = sqrtf(d2x * d2x + d2y * d2y + d2z * d2z);
        out[q] = n2 < n1 ? 1 : 0;                            // argmin over (first, second): first on ties
    }
}

// ---- the 28 corner pairs of every cell: do the two corners side differently, and their field magnitudes
//      (mc_utils.py:170-223) ----
__global__ void vfn_grid_comb_kernel(const long long* choice, const float* norms, float* different, float* pair_norms, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * N * N;
    if (idx >= total) return;
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((long long)N * N));
    float nr[8];
    long long ch[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ii = i + CORNER[q][0], jj = j + CORNER[q][1], kk = k + CORNER[q][2];
        nr[q] = (ii < N && jj < N && kk < N) ? norms[((long long)ii * N + jj) * N + kk] : 0.f;
        ch[q] = choice[idx * 8 + q];
    }
    int c = 0;
    for (int a = 0; a < 7; ++a)
        for (int b = a + 1; b < 8; ++b, ++c) {
            different[idx * 28 + c] = ch[a] != ch[b] ? 1.f : 0.f;
            pair_norms[(idx * 28 + c) * 2] = nr[a];
            pair_norms[(idx * 28 + c) * 2 + 1] = nr[b];
        }
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int vfn_grid_divergence(const float* vt, int32_t n, float threshold, float* out, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(vt && out, "vfn_grid_divergence: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_divergence: resolution %d > 1024", n);
    hipLaunchKernelGGL(vfn_grid_divergence_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, vt, out, n, threshold);
    return vfn_check_launch("vfn_grid_divergence");
}

extern "C" int vfn_grid_smooth_axis(const float* in, float* out, int32_t n, int32_t axis, const float* weights_host, int32_t k, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(in && out && weights_host && in != out, "vfn_grid_smooth_axis: NULL argument or in-place call");
    VFN_REQUIRE(k >= 1 && k <= 15 && (k & 1) && axis >= 0 && axis <= 2 && n <= 1024, "vfn_grid_smooth_axis: bad k=%d / axis=%d / n=%d", k, axis, n);
    SmoothArgs a{};
    a.in = in; a.out = out; a.N = n; a.axis = axis; a.k = k;
    for (int t = 0; t < k; ++t) a.w[t] = weights_host[t];
    hipLaunchKernelGGL(vfn_grid_smooth_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_grid_smooth_axis");
}

extern "C" int vfn_grid_unify_direction(const float* divergence, const float* vt, int32_t n, int64_t* choice, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(divergence && vt && choice, "vfn_grid_unify_direction: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_unify_direction: resolution %d > 1024", n);
    hipLaunchKernelGGL(vfn_grid_unify_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, divergence, vt,
                       (long long*)choice, n);
    return vfn_check_launch("vfn_grid_unify_direction");
}

extern "C" int vfn_grid_comb_format(const int64_t* choice, const float* norms, int32_t n, float* different_side, float* pair_norms, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(choice && norms && different_side && pair_norms, "vfn_grid_comb_format: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_comb_format: resolution %d > 1024", n);
    hipLaunchKernelGGL(vfn_grid_comb_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)choice, norms, different_side, pair_norms, n);
    return vfn_check_launch("vfn_grid_comb_format");
}
