// vfn_grid.hip — the dense-grid stages between the vector-field queries and the mesh triangulation:
// evaluation/utils/mc_utils.py:34-86 (extract_divergence), :107-167 (unify_direction), :170-223 (make_comb_format) and
// evaluation/utils/guassian_smoothing.py:81-97 (smooth_vf).  The reference runs them as conv3d / gather chains on CPU
// tensors of res^3 x 3 floats; here each is an HBM-bound kernel whose every grid value crosses HBM ONCE and whose every
// store instruction writes whole consecutive lines:
//   divergence   a workgroup owns an 8 x 64 footprint of cells in (j, k) and marches along i; the corner terms of the
//                next plane are staged in LDS (each vector read, normalised and evaluated once per workgroup, not eight times), the
//                mask leaves as 256 contiguous bytes per wave.  Algorithmic 16 B / cell (12 in, 4 out).
//   smoothing    axes 0 and 1: the filter runs along a strided axis, so every FLOAT of the interleaved [.., 3] layout is
//                independent — a lane owns four consecutive floats and marches along the axis with the k taps in registers
//                (each input read once, 16-byte accesses); axis 2: whole rows staged in LDS.  24 B / cell / pass.
//   unify        one lane per cell for the mask and the stores, a whole wave per SURFACE cell for the 8-corner analysis (the wave
//                walks the set bits of its ballot); the eight int64 per cell leave through wave-wide 1 KiB stores (a lane's
//                own eight stores would touch 64 lines each).  69 B / cell.
//   comb         per-cell inputs (side bits, 8 corner norms) staged in LDS, the 28 + 56 floats per cell written as
//                float4 rows of the wave's contiguous output range.  336 B / cell out.
// Arithmetic per value is unchanged from the round-2 kernels (same expressions, same order, -ffp-contract=off): masks,
// choices and pair tables stay bit-identical to the reference's functions (tests/test_hip_parity.py).
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

// ------------------------------------------------------------------------------------------------------------------------
// divergence mask: 1 where the normalised field converges onto the cell (mc_utils.py:34-86)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int DV_TJ = 8, DV_TK = 64;                         // cells of the footprint; (TJ+1) x (TK+1) vectors per plane
constexpr int DV_PLANE = (DV_TJ + 1) * (DV_TK + 1);          // 585 vectors
constexpr int DV_PER_THREAD = (DV_PLANE + 255) / 256;        // 3

// What a vector contributes to a cell depends on which corner of the cell it is: corner (a, b, cc) takes
//   x = n0 (a ? +1 : -1) / sqrt(3) + n1 (b ? +1 : -1) / sqrt(3) + n2 (cc ? +1 : -1) / sqrt(3),   term = x |x| face_area,
// and a vector is corner (a, b, cc) of exactly one cell.  Negating all three signs negates x exactly (IEEE products and sums are
// symmetric), hence the term: the four patterns with a = 0, T[b][cc], are computed ONCE per vector when its plane is staged, and
// corner (1, b, cc) reads -T[1 - b][1 - cc].  A cell is then eight LDS reads and eight additions — in the same order and with the
// same values as the eight full evaluations it replaces, so the mask is bit for bit the one of the per-cell formulation.
__global__ __launch_bounds__(256) void vfn_grid_divergence_kernel(const float* __restrict__ vt, float* __restrict__ out, int N, float threshold,
                                                                  int seg_len) {
    __shared__ float T[3][DV_PLANE * 4];                     // the four a = 0 terms of every vector of three consecutive planes (28 KB)
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * DV_TK, j0 = blockIdx.y * DV_TJ;
    const int i0 = blockIdx.z * seg_len, i1 = min(N, i0 + seg_len);
    if (i0 >= i1) return;
    const long long NN = (long long)N * N;
    const float inv3 = 1.0f / sqrtf(3.0f);
    const float face_area = (float)(1.7320508075688772 / 4.0), shape_volume = (float)(1.4142135623730951 / 3.0);

    float r[DV_PER_THREAD][3];
    auto fetch = [&](int i) {                                // this thread's vectors of plane i -> registers (zeros outside the grid)
#pragma unroll
        for (int s = 0; s < DV_PER_THREAD; ++s) {
            const int v = tid + 256 * s;
            const int j = j0 + v / (DV_TK + 1), k = k0 + v % (DV_TK + 1);
            if (v < DV_PLANE && i < N && j < N && k < N) {
                const long long o = ((long long)i * NN + (long long)j * N + k) * 3;
                r[s][0] = vt[o]; r[s][1] = vt[o + 1]; r[s][2] = vt[o + 2];
            } else { r[s][0] = r[s][1] = r[s][2] = 0.f; }
        }
    };
    auto stash = [&](int buf) {                              // normalise (F.normalize: v / max(|v|, 1e-12)), evaluate the four terms, store
#pragma unroll
        for (int s = 0; s < DV_PER_THREAD; ++s) {
            const int v = tid + 256 * s;
            if (v < DV_PLANE) {
                const float nrm = fmaxf(sqrtf(r[s][0] * r[s][0] + r[s][1] * r[s][1] + r[s][2] * r[s][2]), 1e-12f);
                const float n0 = r[s][0] / nrm, n1 = r[s][1] / nrm, n2 = r[s][2] / nrm;
                float4 t;
                float* tp = &t.x;
#pragma unroll
                for (int bc = 0; bc < 4; ++bc) {
                    const int b = bc >> 1, cc = bc & 1;
                    const float x = n0 * (-inv3) + n1 * (b ? inv3 : -inv3) + n2 * (cc ? inv3 : -inv3);
                    tp[bc] = x * fabsf(x) * face_area;
                }
                *reinterpret_cast<float4*>(&T[buf][v * 4]) = t;
            }
        }
    };
    const int kk = tid & 63, jj = tid >> 6;                  // this thread's two cells of a plane: (jj, kk) and (jj + 4, kk)

    fetch(i0);
    stash(0);
    fetch(i0 + 1);
    int n = 0;
    for (int i = i0; i < i1; ++i, ++n) {
        const int b0 = n % 3, b1 = (n + 1) % 3;
        stash(b1);                                           // plane i + 1 (nobody reads buffer (n + 1) % 3 any more: one barrier per plane)
        if (i + 2 <= i1) fetch(i + 2);                       // in flight during the barrier and the arithmetic
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cj = jj + 4 * half;
            const int j = j0 + cj, k = k0 + kk;
            if (j >= N || k >= N) continue;
            float res = 0.f;
            if (i < N - 1 && j < N - 1 && k < N - 1) {
                float s = 0.f;
                // corner c of the 2x2x2 box = (a, b, cc) = (c >> 2, (c >> 1) & 1, c & 1), outward direction (2a-1, 2b-1, 2cc-1) / sqrt(3)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int a = c >> 2, b = (c >> 1) & 1, cc = c & 1;
                    const int vec = (cj + b) * (DV_TK + 1) + kk + cc;
                    s += a ? -T[b1][vec * 4 + 2 * (1 - b) + (1 - cc)] : T[b0][vec * 4 + 2 * b + cc];
                }
                res = s / shape_volume;
            }
            out[(long long)i * NN + (long long)j * N + k] = res > threshold ? 0.f : 1.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// one pass of the separable Gaussian along one axis, replicate padding (guassian_smoothing.py:81-97)
// ------------------------------------------------------------------------------------------------------------------------
struct SmoothArgs { const float* in; float* out; int N; int axis; int k; float w[16]; };

// any odd k <= 15 (the reference only uses 3 and 9): one thread per voxel, taps re-read through the caches
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

template <int V> struct Pack;
template <> struct Pack<4> { typedef float4 T; };
template <> struct Pack<1> { typedef float T; };
template <int V> __device__ __forceinline__ void fma_pack(typename Pack<V>::T& acc, float w, const typename Pack<V>::T& x);
template <> __device__ __forceinline__ void fma_pack<4>(float4& acc, float w, const float4& x) {
    acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
}
template <> __device__ __forceinline__ void fma_pack<1>(float& acc, float w, const float& x) { acc += w * x; }
template <int V> __device__ __forceinline__ typename Pack<V>::T zero_pack();
template <> __device__ __forceinline__ float4 zero_pack<4>() { return float4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ float zero_pack<1>() { return 0.f; }

// Axes 0 and 1.  The grid as a flat float array: element (outer, p, inner) at (outer * N + p) * inner_len + inner, the filter
// runs over p; inner_len = 3 N^2 (axis 0, outer_len 1) or 3 N (axis 1, outer_len N).  A lane owns V consecutive `inner` floats of
// one `outer` and marches p over [p_lo, p_hi) of its segment with the K taps in a register ring whose slots rotate at compile
// time (the march is unrolled K positions at a time): every input is read once, every access is V floats wide and consecutive
// across lanes.  Sums run over the taps in ascending order, like the per-voxel kernel.
template <int K, int V>
__global__ __launch_bounds__(256) void vfn_grid_smooth_march_kernel(const SmoothArgs a, long long inner_len, int outer_len, int seg_len) {
    typedef typename Pack<V>::T T;
    const long long cols = inner_len / V;
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols * outer_len) return;
    const long long outer = col / cols, inner = (col % cols) * V;
    const int N = a.N, H = K / 2;
    const int p_lo = blockIdx.y * seg_len, p_hi = min(N, p_lo + seg_len);
    if (p_lo >= p_hi) return;
    const float* src = a.in + outer * (long long)N * inner_len + inner;
    float* dst = a.out + outer * (long long)N * inner_len + inner;
    auto at = [&](int q) -> T {                              // replicate padding
        q = q < 0 ? 0 : (q >= N ? N - 1 : q);
        return *reinterpret_cast<const T*>(src + (long long)q * inner_len);
    };
    T win[K];                                                // position p_lo + m K + s: tap t sits in slot (s + t) % K
#pragma unroll
    for (int t = 0; t < K - 1; ++t) win[t] = at(p_lo + t - H);
    for (int p = p_lo; p < p_hi; p += K) {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (p + s < p_hi) {
                win[(s + K - 1) % K] = at(p + s + H);
                T acc = zero_pack<V>();
#pragma unroll
                for (int t = 0; t < K; ++t) fma_pack<V>(acc, a.w[t], win[(s + t) % K]);
                *reinterpret_cast<T*>(dst + (long long)(p + s) * inner_len) = acc;
            }
        }
    }
}

// Axis 2 (the innermost grid index): a row of N vectors = 3 N consecutive floats, the taps are 3 floats apart.  A workgroup
// stages SM_ROWS whole rows in LDS (each input read once, consecutive) and every lane produces consecutive output floats.
constexpr int SM_ROWS = 4;
template <int K>
__global__ __launch_bounds__(256) void vfn_grid_smooth_rows_kernel(const SmoothArgs a) {
    extern __shared__ float rows[];                          // SM_ROWS x 3N floats
    const int N = a.N, H = K / 2, row_len = 3 * N;
    const long long n_rows = (long long)N * N;
    const long long r0 = (long long)blockIdx.x * SM_ROWS;
    const int nr = (int)min((long long)SM_ROWS, n_rows - r0);
    const int floats = nr * row_len;
    const float* src = a.in + r0 * row_len;
    float* dst = a.out + r0 * row_len;
    if ((row_len & 3) == 0) {
        for (int f = threadIdx.x * 4; f < floats; f += 1024) *reinterpret_cast<float4*>(rows + f) = *reinterpret_cast<const float4*>(src + f);
    } else {
        for (int f = threadIdx.x; f < floats; f += 256) rows[f] = src[f];
    }
    __syncthreads();
    auto one = [&](const float* row_base, int e) -> float {  // e = 3 kpos + c within the row
        const int kpos = e / 3, c = e - kpos * 3;
        const float* row = row_base + c;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            int q = kpos + t - H;
            q = q < 0 ? 0 : (q >= N ? N - 1 : q);
            acc += a.w[t] * row[q * 3];
        }
        return acc;
    };
    for (int r = 0; r < nr; ++r) {
        const float* row = rows + r * row_len;
        float* drow = dst + (long long)r * row_len;
        if ((row_len & 3) == 0) {
            for (int e = threadIdx.x * 4; e < row_len; e += 1024)
                *reinterpret_cast<float4*>(drow + e) = float4{one(row, e), one(row, e + 1), one(row, e + 2), one(row, e + 3)};
        } else {
            for (int e = threadIdx.x; e < row_len; e += 256) drow[e] = one(row, e);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// per surface cell: the two most opposed corner vectors, and for every corner which of the two it sides with
// (mc_utils.py:107-167).  Side bits of a cell = one byte; the int64 [N^3, 8] table of the reference's interface is written
// from the bytes by the whole wave: 64 cells x 64 B = 4 KiB consecutive, four stores of 1 KiB.
// ------------------------------------------------------------------------------------------------------------------------
// The analysis of ONE surface cell by a whole wave: lanes 0..23 fetch the 24 floats of its eight corner vectors, lane L = 8 a + b
// evaluates the pair (a, b) of the 64 — the reference's distance matrix 1 - <v_a, v_b> in its own order of operations — a wave
// reduction picks the first maximum in a-major order (torch.argmax over the flattened 8 x 8 matrix), lanes 0..7 decide their
// corner's side.  Surface cells are a few percent of the grid: a wave walks the set bits of its ballot instead of sending all 64
// lanes through 64 pairs whenever one of them sits on the surface (which also took 176 VGPRs, i.e. two waves per SIMD for a
// kernel whose real job is to write 64 bytes per cell).
__device__ __forceinline__ unsigned cell_sides_wave(const float* __restrict__ vt, int N, long long idx, int lane) {
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((long long)N * N));
    float val = 0.f;
    if (lane < 24) {
        const int q = lane / 3, c = lane - 3 * q;
        const int ii = i + CORNER[q][0], jj = j + CORNER[q][1], kk = k + CORNER[q][2];
        if (ii < N && jj < N && kk < N) val = vt[(((long long)ii * N + jj) * N + kk) * 3 + c];
    }
    const int a = lane >> 3, b = lane & 7;
    const float a0 = __shfl(val, 3 * a, 64), a1 = __shfl(val, 3 * a + 1, 64), a2 = __shfl(val, 3 * a + 2, 64);
    const float b0 = __shfl(val, 3 * b, 64), b1 = __shfl(val, 3 * b + 1, 64), b2 = __shfl(val, 3 * b + 2, 64);
    float best = 1.0f - ((a0 * b0 + a1 * b1) + a2 * b2);
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }      // first maximum, as torch.argmax
    }
    const int f = bi >> 3, sc = bi & 7;
    const float fx = __shfl(val, 3 * f, 64), fy = __shfl(val, 3 * f + 1, 64), fz = __shfl(val, 3 * f + 2, 64);
    const float sx = __shfl(val, 3 * sc, 64), sy = __shfl(val, 3 * sc + 1, 64), sz = __shfl(val, 3 * sc + 2, 64);
    // lanes 0..7 hold (b0, b1, b2) = corner `lane` (b = lane there)
    const float d1x = fx - b0, d1y = fy - b1, d1z = fz - b2;
    const float d2x = sx - b0, d2y = sy - b1, d2z = sz - b2;
    const float n1 = sqrtf(d1x * d1x + d1y * d1y + d1z * d1z), n2 = sqrtf(d2x * d2x + d2y * d2y + d2z * d2z);
    return (unsigned)(__ballot(n2 < n1) & 0xffull);                              // argmin over (first, second): first on ties
}

__global__ __launch_bounds__(256) void vfn_grid_unify_kernel(const float* __restrict__ div, const float* __restrict__ vt, long long* __restrict__ choice,
                                                             unsigned char* __restrict__ sides, int N) {
    const long long total = (long long)N * N * N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long long first = idx - lane;                      // the wave's first cell
    unsigned bits = 0;
    unsigned long long todo = __ballot(idx < total && div[idx < total ? idx : 0] == 1.0f);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned m = cell_sides_wave(vt, N, first + src, lane);
        if (lane == src) bits = m;
    }
    if (sides && idx < total) sides[idx] = (unsigned char)bits;
    if (!choice) return;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;                        // 16-byte chunk of the wave's 4 KiB: cell c / 4, corners 2 (c % 4), + 1
        const unsigned m = (unsigned)__shfl((int)bits, c >> 2, 64);
        const int q0 = (c & 3) * 2;
        if (first + (c >> 2) < total)
            *reinterpret_cast<ll2*>(choice + (first + (c >> 2)) * 8 + q0) = ll2{(long long)((m >> q0) & 1u), (long long)((m >> (q0 + 1)) & 1u)};
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// the 28 corner pairs of every cell: do the two corners side differently, and their field magnitudes (mc_utils.py:170-223)
// ------------------------------------------------------------------------------------------------------------------------
__device__ __constant__ unsigned char PAIR_A[28] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6};
__device__ __constant__ unsigned char PAIR_B[28] = {1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};

template <bool FROM_BYTES>
__global__ __launch_bounds__(256) void vfn_grid_comb_kernel(const long long* __restrict__ choice, const unsigned char* __restrict__ sides,
                                                            const float* __restrict__ norms, float* __restrict__ different,
                                                            float* __restrict__ pair_norms, int N) {
    __shared__ float nr[4][64][9];                           // 8 corner norms per cell (+1: lanes 9 floats apart, no bank conflict)
    __shared__ unsigned mk[4][64];                           // side bits per cell
    __shared__ unsigned char pa[28], pb[28];
    const long long total = (long long)N * N * N;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 28) { pa[threadIdx.x] = PAIR_A[threadIdx.x]; pb[threadIdx.x] = PAIR_B[threadIdx.x]; }
    unsigned bits = 0;
    if (idx < total) {
        const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((long long)N * N));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ii = i + CORNER[q][0], jj = j + CORNER[q][1], kk = k + CORNER[q][2];
            nr[wave][lane][q] = (ii < N && jj < N && kk < N) ? norms[((long long)ii * N + jj) * N + kk] : 0.f;
        }
        if (FROM_BYTES) bits = sides[idx];
        else {
            // "!=" between two int64 entries of the table: with entries 0 / 1 (what unify_direction writes) one bit per entry holds
            // it; any other table is compared entry by entry below
            typedef long long ll2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const ll2 v = *reinterpret_cast<const ll2*>(choice + idx * 8 + 2 * h);
                bits |= (v[0] != 0 ? 1u : 0u) << (2 * h) | (v[1] != 0 ? 1u : 0u) << (2 * h + 1);
                if ((v[0] & ~1ll) | (v[1] & ~1ll)) bits |= 0x100u;      // an entry outside {0, 1}
            }
        }
    }
    mk[wave][lane] = bits;
    __syncthreads();
    const long long first = idx - lane;
    const int valid = (int)min(64ll, total - first);         // cells of this wave inside the grid
    if (valid <= 0) return;
    const bool general = !FROM_BYTES && __any((int)(bits & 0x100u));
    // different_side: 28 floats per cell, the wave's 64 cells = 1792 consecutive floats = 7 x (64 lanes x float4)
    float* d0 = different + first * 28;
    for (int it = 0; it < 7; ++it) {
        const int e0 = 4 * (it * 64 + lane);
        float val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + r, cell = e / 28, c = e - cell * 28;
            const unsigned m = mk[wave][cell];
            float x = (float)(((m >> pa[c]) ^ (m >> pb[c])) & 1u);
            if (general && cell < valid) x = choice[(first + cell) * 8 + pa[c]] != choice[(first + cell) * 8 + pb[c]] ? 1.f : 0.f;
            val[r] = x;
        }
        if (e0 + 3 < valid * 28) *reinterpret_cast<float4*>(d0 + e0) = float4{val[0], val[1], val[2], val[3]};
        else
            for (int r = 0; r < 4; ++r) if (e0 + r < valid * 28) d0[e0 + r] = val[r];
    }
    // pair_norms: 56 floats per cell = 14 x (64 lanes x float4)
    float* p0 = pair_norms + first * 56;
    for (int it = 0; it < 14; ++it) {
        const int e0 = 4 * (it * 64 + lane);
        float val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + r, cell = e / 56, rem = e - cell * 56, c = rem >> 1;
            val[r] = nr[wave][cell][(rem & 1) ? pb[c] : pa[c]];
        }
        if (e0 + 3 < valid * 56) *reinterpret_cast<float4*>(p0 + e0) = float4{val[0], val[1], val[2], val[3]};
        else
            for (int r = 0; r < 4; ++r) if (e0 + r < valid * 56) p0[e0 + r] = val[r];
    }
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

template <int K>
int launch_smooth(const SmoothArgs& a, hipStream_t s) {
    const long long N = a.N;
    if (a.axis == 2) {
        const size_t lds = (size_t)SM_ROWS * 3 * N * sizeof(float);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(vfn_grid_smooth_rows_kernel<K>), dim3((unsigned)((N * N + SM_ROWS - 1) / SM_ROWS)), dim3(256), lds, s, a);
        return VFN_OK;
    }
    const long long inner_len = a.axis == 0 ? 3 * N * N : 3 * N;
    const int outer_len = a.axis == 0 ? 1 : (int)N;
    const int v = (inner_len % 4 == 0) ? 4 : 1;
    const long long lanes = inner_len / v * outer_len;
    // enough waves to fill the chip: split the march into segments (each re-reads K - 1 planes) while there are fewer than ~2048
    int segs = (int)((2048ll * 64 + lanes - 1) / lanes);
    segs = segs < 1 ? 1 : (segs > (int)((N + 4 * K - 1) / (4 * K)) ? (int)((N + 4 * K - 1) / (4 * K)) : segs);
    int seg_len = (int)((N + segs - 1) / segs);
    seg_len = (seg_len + K - 1) / K * K;
    segs = (int)((N + seg_len - 1) / seg_len);
    const dim3 grid(blocks_for(lanes), (unsigned)segs);
    if (v == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(vfn_grid_smooth_march_kernel<K, 4>), grid, dim3(256), 0, s, a, inner_len, outer_len, seg_len);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(vfn_grid_smooth_march_kernel<K, 1>), grid, dim3(256), 0, s, a, inner_len, outer_len, seg_len);
    return VFN_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// lattice points: rows [row0, row0 + count) of the n^3 x 3 grid evaluation/methods.py:194-208 fills on the host, from its three AXIS
// TABLES (the lattice is separable: column 0 depends on i alone, column 1 on j, column 2 on k — whatever fp32 expression filled
// them, so the points are the caller's bit for bit).  12 B per point written, the tables (3 n floats) stay in L2.
// ------------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void vfn_lattice_points_kernel(const float* __restrict__ ax0, const float* __restrict__ ax1,
                                                                 const float* __restrict__ ax2, int n, long long row0, long long count,
                                                                 float* __restrict__ out) {
    // a lane owns four consecutive FLOATS of the [count, 3] output (16-byte stores): float f -> row f / 3, column f % 3
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long f0 = q * 4, total = count * 3;
    if (f0 >= total) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long f = f0 + e;
        const long long row = row0 + f / 3;
        const int c = (int)(f % 3);
        const int k = (int)(row % n), j = (int)((row / n) % n), i = (int)(row / ((long long)n * n));
        v[e] = f < total ? (c == 0 ? ax0[i] : (c == 1 ? ax1[j] : ax2[k])) : 0.f;
    }
    if (f0 + 4 <= total) *reinterpret_cast<float4*>(out + f0) = float4{v[0], v[1], v[2], v[3]};
    else for (int e = 0; e < 4 && f0 + e < total; ++e) out[f0 + e] = v[e];
}
}  // namespace

extern "C" int vfn_grid_lattice_points(const float* axis0, const float* axis1, const float* axis2, int32_t n, int64_t row0, int64_t count,
                                       float* points, void* stream) {
    VFN_REQUIRE(axis0 && axis1 && axis2 && points && n > 0 && n <= 2048 && row0 >= 0 && count >= 0 && row0 + count <= (int64_t)n * n * n,
                "vfn_grid_lattice_points: bad argument (n = %d, rows [%lld, %lld))", n, (long long)row0, (long long)(row0 + count));
    if (count == 0) return VFN_OK;
    const long long quads = (count * 3 + 3) / 4;
    hipLaunchKernelGGL(vfn_lattice_points_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, axis0, axis1, axis2, n,
                       (long long)row0, (long long)count, points);
    return vfn_check_launch("vfn_grid_lattice_points");
}

extern "C" int vfn_grid_divergence(const float* vt, int32_t n, float threshold, float* out, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(vt && out, "vfn_grid_divergence: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_divergence: resolution %d > 1024", n);
    const int bx = (n + DV_TK - 1) / DV_TK, by = (n + DV_TJ - 1) / DV_TJ;
    // planes per workgroup: segments of >= 16 planes (one extra plane each) until there are ~8 192 workgroups — five fit on a CU
    // (28 KB of LDS each), and 2 048 of them would be 1.6 rounds of the chip, i.e. a fifth of it idle in the second
    int segs = (8192 + bx * by - 1) / (bx * by);
    segs = segs < 1 ? 1 : (segs > (n + 15) / 16 ? (n + 15) / 16 : segs);
    const int seg_len = (n + segs - 1) / segs;
    hipLaunchKernelGGL(vfn_grid_divergence_kernel, dim3(bx, by, (n + seg_len - 1) / seg_len), dim3(256), 0, (hipStream_t)stream, vt, out, n, threshold,
                       seg_len);
    return vfn_check_launch("vfn_grid_divergence");
}

extern "C" int vfn_grid_smooth_axis(const float* in, float* out, int32_t n, int32_t axis, const float* weights_host, int32_t k, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(in && out && weights_host && in != out, "vfn_grid_smooth_axis: NULL argument or in-place call");
    VFN_REQUIRE(k >= 1 && k <= 15 && (k & 1) && axis >= 0 && axis <= 2 && n <= 1024, "vfn_grid_smooth_axis: bad k=%d / axis=%d / n=%d", k, axis, n);
    SmoothArgs a{};
    a.in = in; a.out = out; a.N = n; a.axis = axis; a.k = k;
    for (int t = 0; t < k; ++t) a.w[t] = weights_host[t];
    // the two filters the reference uses (k = 3 with sigma 1, k = 9 with sigma 2: evaluation/methods.py:214-221) have register-window /
    // LDS-row kernels; rows must fit the LDS budget of the axis-2 kernel
    const bool fits = (size_t)SM_ROWS * 3 * n * sizeof(float) <= 64 * 1024;
    if (k == 3 && (axis != 2 || fits)) launch_smooth<3>(a, (hipStream_t)stream);
    else if (k == 9 && (axis != 2 || fits)) launch_smooth<9>(a, (hipStream_t)stream);
    else hipLaunchKernelGGL(vfn_grid_smooth_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_grid_smooth_axis");
}

extern "C" int vfn_grid_unify_direction(const float* divergence, const float* vt, int32_t n, int64_t* choice, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(divergence && vt && choice, "vfn_grid_unify_direction: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_unify_direction: resolution %d > 1024", n);
    hipLaunchKernelGGL(vfn_grid_unify_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, divergence, vt,
                       (long long*)choice, (unsigned char*)nullptr, n);
    return vfn_check_launch("vfn_grid_unify_direction");
}

extern "C" int vfn_grid_unify_direction_sides(const float* divergence, const float* vt, int32_t n, uint8_t* sides, int64_t* choice, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(divergence && vt && sides, "vfn_grid_unify_direction_sides: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_unify_direction_sides: resolution %d > 1024", n);
    hipLaunchKernelGGL(vfn_grid_unify_kernel, dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream, divergence, vt,
                       (long long*)choice, (unsigned char*)sides, n);
    return vfn_check_launch("vfn_grid_unify_direction_sides");
}

extern "C" int vfn_grid_comb_format(const int64_t* choice, const float* norms, int32_t n, float* different_side, float* pair_norms, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(choice && norms && different_side && pair_norms, "vfn_grid_comb_format: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_comb_format: resolution %d > 1024", n);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(vfn_grid_comb_kernel<false>), dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)choice, (const unsigned char*)nullptr, norms, different_side, pair_norms, n);
    return vfn_check_launch("vfn_grid_comb_format");
}

extern "C" int vfn_grid_comb_format_sides(const uint8_t* sides, const float* norms, int32_t n, float* different_side, float* pair_norms, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(sides && norms && different_side && pair_norms, "vfn_grid_comb_format_sides: NULL argument");
    VFN_REQUIRE(n <= 1024, "vfn_grid_comb_format_sides: resolution %d > 1024", n);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(vfn_grid_comb_kernel<true>), dim3(blocks_for((long long)n * n * n)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)nullptr, (const unsigned char*)sides, norms, different_side, pair_norms, n);
    return vfn_check_launch("vfn_grid_comb_format_sides");
}
