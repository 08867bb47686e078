// vfn_mlp_bwd.hip — backward of the fused MLPs: autograd of
// models/vector_field/vector_field_network.py:177-208 and rendering_network.py:62-108 in the
// shipped training regime (eval-mode BatchNorm, train/vector_field_nerf_train.py:140-141).
//
// Two kernel families, both fp32 MFMA (v_mfma_f32_32x32x2_f32):
//
//  (1) vfn_mlp_bwd_kernel — the dX chain.  One workgroup = 64 points walks the layers backwards with the
//      64x256 gradient tile resident in LDS: dX_l = dY_l * W'_l streams the TRANSPOSED packed tiles from L2
//      exactly like the forward streams W'; the epilogue loads the saved post-activation output of the layer
//      below (ReLU mask / tanh derivative), writes dY_{l-1} to HBM for the weight-gradient kernels and into
//      the tile for the next step.  The 3-channel heads enter as rank-3 updates in the epilogue.
//
//  (2) vfn_dw_kernel — weight gradients dW'_l = dY_l^T X_l.  Persistent workgroups, each reduces a
//      contiguous slab of points with the WHOLE 256x256 gradient held in accumulator registers (4 waves x
//      128x128 = 256 VGPRs each): both MFMA operands are row-major global loads in their natural layout
//      (A[i=n][k=m] = dY[m][n], B[k=m][j=k] = X[m][k]), no transposes, no LDS.  Bias gradients fall out of the
//      A operands (column sums).  Each workgroup writes one partial slab; slabs are summed afterwards.
#include <string.h>
#include "vfn_common.h"
#include "vfn_plan.h"
#include "vfn_mlp_core.h"
using namespace vfn;

namespace {

enum : int { MASK_RELU = 0, MASK_TANH = 1 };

struct BwdArgs {
    VfnNetPlan vf;
    VfnNetPlan rn;
    const float* vf_w;    // forward pack (head rows)
    const float* vf_wb;   // backward pack
    const float* rn_w;
    const float* rn_wb;
    const float* saved;   // [slots][M][256] post-activation outputs from the training forward
    float* dy;            // [slots][M][256] pre-activation gradients (output)
    const float* d_colors;  // [M,3]
    const float* colors;    // [M,3]
    const float* d_vec;     // gradient wrt the tanh'ed vector columns, row stride vec_stride
    const float* vec;       // the tanh'ed vector columns, row stride vec_stride
    const float* d_feats;   // VF-only mode: gradient wrt the tanh'ed features, row stride vec_stride (or NULL)
    float* dz_rgb;          // [M,4] out: pre-sigmoid gradient of the RGB head
    float* dz_vec;          // [M,4] out: pre-tanh gradient of the vector head
    long long n_points;
    int vec_stride;
    int fused;              // 1: rendering net + VF net; 0: VF net only
};

__device__ __forceinline__ float head_weight(const float* __restrict__ wbase, uint32_t head_w_off, int n, int col) {
    // forward head pack: w[kb16][lane = 16*q + n][j], k = 16*kb16 + 4*q + j
    return wbase[head_w_off + ((((col >> 4) * 64) + (((col & 15) >> 2) * 16) + n) << 2) + (col & 3)];
}

// One backward step for this wave's output tiles:
//   v[row][col] = (DO_MMA ? sum_n tile[row][n] * W'[n][col] : 0) + (head ? sum_{c<3} s_dz[row][c] * Whead[c][col] : 0)
//   dy = v * f'(saved[mask_slot][row][col]);  dY[mask_slot] <- dy;  tile <- dy
template <int NT>
__device__ __forceinline__ void bwd_step_t(bool do_mma, int n_red_tiles, uint32_t bw_off, const float* __restrict__ wb,
                                           bool head_np, uint32_t head_w_off, const float* __restrict__ head_w,
                                           const float* s_dz, float* s_tile, const float* __restrict__ saved_slot,
                                           float* __restrict__ dy_slot, int mask_kind, long long row0, long long n_rows,
                                           int tile0, int lane) {
    f32x16 acc[2][2];
    acc[0][0] = splat16(0.f); acc[0][1] = splat16(0.f); acc[1][0] = splat16(0.f); acc[1][1] = splat16(0.f);
    if (do_mma) {
        const int nb = 4 * n_red_tiles;
        const f32x4* w0 = reinterpret_cast<const f32x4*>(wb + bw_off) + (size_t)tile0 * nb * 64;
        const f32x4* w1 = w0 + (size_t)nb * 64;
        mma_segment<NT, true>(acc, s_tile, nb, w0, w1, lane);
    }
    int c = lane & 31, h = lane >> 5;
    // make the lane coordinates opaque here: otherwise hipcc hoists the ~190 lane-constant LDS / global offsets of
    // this epilogue out of the layer loop and spills them around every MFMA phase
    asm volatile("" : "+v"(c), "+v"(h));
    float hw[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (head_np) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) hw[nt][ch] = head_weight(head_w, head_w_off, ch, 32 * (tile0 + nt) + c);
    }
    // the saved activations of this tile through a bounds-checked buffer descriptor: rows past the end of the
    // batch read as 0 without branches (a guarded plain load makes hipcc serialise every element)
    const int rows_in = (int)min((long long)TM, n_rows - row0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(saved_slot + (size_t)row0 * ACT_LD), 0, rows_in * ACT_LD * 4, 0x00020000);
    float* dy_tile = dy_slot + (size_t)row0 * ACT_LD;
    __syncthreads();  // every wave has finished reading the tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float xs[NT][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                xs[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (row * ACT_LD + 32 * (tile0 + nt) + c) * 4, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            float dz0 = 0.f, dz1 = 0.f, dz2 = 0.f;
            if (head_np) { dz0 = s_dz[row * 4 + 0]; dz1 = s_dz[row * 4 + 1]; dz2 = s_dz[row * 4 + 2]; }
            const bool in = row < rows_in;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 32 * (tile0 + nt) + c;
                float v = acc[mt][nt][r];
                if (head_np) v += dz0 * hw[nt][0] + dz1 * hw[nt][1] + dz2 * hw[nt][2];
                const float x = xs[nt][r];
                const float dyv = (mask_kind == MASK_RELU) ? (x > 0.f ? v : 0.f) : v * (1.0f - x * x);
                if (in) dy_tile[row * ACT_LD + col] = dyv;
                s_tile[act_idx(row, col)] = in ? dyv : 0.f;
            }
        }
    }
    __syncthreads();
}

// lp (by value: taking addresses of kernel-argument members would force a private-memory copy of the plans)
// supplies the reduction width (n_tiles) and the transposed tiles; has_head adds the rank-3 head update.
__device__ __forceinline__ void bwd_step(bool do_mma, const VfnLayerPlan lp, const float* wb, bool has_head,
                                         uint32_t head_w_off, const float* head_w, const float* s_dz, float* s_tile,
                                         const float* saved_slot, float* dy_slot, int mask_kind, long long row0,
                                         long long n_rows, int n_out_tiles, int wave, int lane) {
    const int tile0 = 2 * wave;
    const int nt = min(2, max(0, n_out_tiles - tile0));  // wave-uniform
    if (nt == 2)
        bwd_step_t<2>(do_mma, lp.n_tiles, lp.bw_off, wb, has_head, head_w_off, head_w, s_dz, s_tile, saved_slot, dy_slot,
                      mask_kind, row0, n_rows, tile0, lane);
    else if (nt == 1)
        bwd_step_t<1>(do_mma, lp.n_tiles, lp.bw_off, wb, has_head, head_w_off, head_w, s_dz, s_tile, saved_slot, dy_slot,
                      mask_kind, row0, n_rows, tile0, lane);
    else { __syncthreads(); __syncthreads(); }
}

__global__ __launch_bounds__(NTHREADS, 2) void vfn_mlp_bwd_kernel(const BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[TM * ACT_LD + TM * 4];
    float* s_tile = smem;
    float* s_dz = smem + TM * ACT_LD;  // [64][4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long row0 = (long long)blockIdx.x * TM;
    const long long n_rows = a.n_points;
    const size_t slot = (size_t)n_rows * ACT_LD;
    const int vfH = a.vf.n_hidden, n_plain = vfH - a.vf.feat_layer;
    bool have_feat_grad = false;

    if (a.fused) {
        const int rnH = a.rn.n_hidden;
        // (a) RGB head: dZ = dC * c (1 - c)
        if (tid < TM) {
            const long long row = row0 + tid;
            float dz[3] = {0.f, 0.f, 0.f};
            if (row < n_rows) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float cc = a.colors[row * 3 + ch];
                    dz[ch] = a.d_colors[row * 3 + ch] * cc * (1.0f - cc);
                    a.dz_rgb[row * 4 + ch] = dz[ch];
                }
                a.dz_rgb[row * 4 + 3] = 0.f;
            }
            s_dz[tid * 4 + 0] = dz[0]; s_dz[tid * 4 + 1] = dz[1]; s_dz[tid * 4 + 2] = dz[2]; s_dz[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        // (b) gradient wrt the last hidden output of the rendering net = rank-3 update, ReLU-masked
        bwd_step(false, a.rn.hidden[0], nullptr, true, a.rn.head_w_off, a.rn_w, s_dz, s_tile, a.saved + (size_t)(vfH + rnH - 1) * slot,
                 a.dy + (size_t)(vfH + rnH - 1) * slot, MASK_RELU, row0, n_rows, VFN_HIDDEN / 32, wave, lane);
        // (c) hidden layers rnH-1 .. 1
        for (int hh = rnH - 1; hh >= 1; --hh)
            bwd_step(true, a.rn.hidden[hh], a.rn_wb, false, 0u, nullptr, s_dz, s_tile, a.saved + (size_t)(vfH + hh - 1) * slot,
                     a.dy + (size_t)(vfH + hh - 1) * slot, MASK_RELU, row0, n_rows, a.rn.hidden[hh].nkb_act / 4, wave, lane);
        // (d) layer 0: gradient wrt the features, through tanh -> feature slot of the VF net
        bwd_step(true, a.rn.hidden[0], a.rn_wb, false, 0u, nullptr, s_dz, s_tile, a.saved + (size_t)(vfH - 1) * slot,
                 a.dy + (size_t)(vfH - 1) * slot, MASK_TANH, row0, n_rows, a.rn.hidden[0].nkb_act / 4, wave, lane);
        have_feat_grad = true;
    } else if (a.d_feats && a.vf.feat_layer) {
        // VF-only: dZ_f = dF * (1 - F^2) straight from the caller's gradient
        const float* F = a.saved + (size_t)(vfH - 1) * slot;
        float* dyf = a.dy + (size_t)(vfH - 1) * slot;
        for (int i = tid; i < TM * ACT_LD; i += NTHREADS) {
            const int row = i >> 8, col = i & 255;
            float v = 0.f;
            if (row0 + row < n_rows) {
                const float f = F[(size_t)(row0 + row) * ACT_LD + col];
                v = a.d_feats[(size_t)(row0 + row) * a.vec_stride + col] * (1.0f - f * f);
                dyf[(size_t)(row0 + row) * ACT_LD + col] = v;
            }
            s_tile[act_idx(row, col)] = v;
        }
        have_feat_grad = true;
        __syncthreads();
    }

    // vector head: dZ = dV * (1 - v^2)
    if (tid < TM) {
        const long long row = row0 + tid;
        float dz[3] = {0.f, 0.f, 0.f};
        if (row < n_rows) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float t = a.vec[row * a.vec_stride + ch];
                dz[ch] = a.d_vec[row * a.vec_stride + ch] * (1.0f - t * t);
                a.dz_vec[row * 4 + ch] = dz[ch];
            }
            a.dz_vec[row * 4 + 3] = 0.f;
        }
        s_dz[tid * 4 + 0] = dz[0]; s_dz[tid * 4 + 1] = dz[1]; s_dz[tid * 4 + 2] = dz[2]; s_dz[tid * 4 + 3] = 0.f;
    }
    __syncthreads();
    // (e) last Linear of the VF net: feature block (MFMA) + vector head (rank 3) -> last plain hidden output
    bwd_step(have_feat_grad, a.vf.hidden[vfH - 1], a.vf_wb, true, a.vf.head_w_off, a.vf_w, s_dz, s_tile,
             a.saved + (size_t)(n_plain - 1) * slot, a.dy + (size_t)(n_plain - 1) * slot, MASK_RELU, row0, n_rows,
             VFN_HIDDEN / 32, wave, lane);
    // (f) plain hidden layers n_plain-1 .. 1 (layer 0 consumes only the encoding: no dX)
    for (int hh = n_plain - 1; hh >= 1; --hh)
        bwd_step(true, a.vf.hidden[hh], a.vf_wb, false, 0u, nullptr, s_dz, s_tile, a.saved + (size_t)(hh - 1) * slot,
                 a.dy + (size_t)(hh - 1) * slot, MASK_RELU, row0, n_rows, a.vf.hidden[hh].nkb_act / 4, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------------
struct DwArgs {
    const float* dy;      // [M][ld_dy]
    const float* x;       // [M][ld_x]
    float* dw_part;       // [G][n_rows_out][ld_out]
    float* db_part;       // [G][n_rows_out]   (NULL to skip)
    long long n_points;
    int ld_dy, n_valid;   // dY row stride, valid dY columns (others read as 0)
    int ld_x, k_valid;    // X row stride, valid X columns
    int ld_out;           // row stride of the output slab (= 32 * total k tiles)
    int n_out;            // rows of the output slab (= 32 * total n tiles)
};

// Wave (wn, wk) of a WN x WK wave grid owns n tiles [wn*NN, wn*NN+NN) and k tiles [wk*KK, wk*KK+KK).
template <int NN, int KK, int WN, int WK>
__global__ __launch_bounds__(256, 1) void vfn_dw_kernel(const DwArgs a) {
    static_assert(WN * WK == 4, "four waves per workgroup");
    constexpr int U = 4;  // row pairs per load group
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int c = lane & 31, h = lane >> 5;
    const int G = gridDim.x, g = blockIdx.x;
    // contiguous slab of row pairs for this workgroup
    const long long pairs = (a.n_points + 1) / 2;
    const long long per = (pairs + G - 1) / G;
    const long long p0 = g * per, p1 = min(pairs, p0 + per);

    f32x16 acc[NN][KK];
#pragma unroll
    for (int i = 0; i < NN; ++i)
#pragma unroll
        for (int t = 0; t < KK; ++t) acc[i][t] = splat16(0.f);
    float bsum[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) bsum[i] = 0.f;

    // Slab-relative, bounds-checked buffer descriptors: rows past the slab / batch end and columns past
    // n_valid / k_valid get an out-of-range offset and read as 0 — no branches, no per-element waits.
    const long long r_base = 2 * p0;
    const long long rows_slab = max(0LL, min(2 * (p1 - p0), a.n_points - r_base));
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy + (size_t)r_base * a.ld_dy), 0, (int)(rows_slab * a.ld_dy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (size_t)r_base * a.ld_x), 0, (int)(rows_slab * a.ld_x * 4), 0x00020000);
    constexpr unsigned OOB = 0x7fffffffu;
    int ncol[NN];
    unsigned noff[NN], koff[KK];
#pragma unroll
    for (int i = 0; i < NN; ++i) { ncol[i] = 32 * (wn * NN + i) + c; noff[i] = ncol[i] < a.n_valid ? (unsigned)ncol[i] * 4u : OOB; }
#pragma unroll
    for (int t = 0; t < KK; ++t) { const int kc = 32 * (wk * KK + t) + c; koff[t] = kc < a.k_valid ? (unsigned)kc * 4u : OOB; }

    float a0v[U][NN], b0v[U][KK], a1v[U][NN], b1v[U][KK];
    auto load_group = [&](long long p, float (&av)[U][NN], float (&bv)[U][KK]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned r = (unsigned)(2 * (p - p0 + u) + h);
            const unsigned ra = r * (unsigned)a.ld_dy * 4u, rb = r * (unsigned)a.ld_x * 4u;
#pragma unroll
            for (int i = 0; i < NN; ++i)
                av[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dy, noff[i] == OOB ? OOB : ra + noff[i], 0, 0));
#pragma unroll
            for (int t = 0; t < KK; ++t)
                bv[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, koff[t] == OOB ? OOB : rb + koff[t], 0, 0));
        }
    };
    auto compute = [&](const float (&av)[U][NN], const float (&bv)[U][KK]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < NN; ++i) {
                bsum[i] += av[u][i];
#pragma unroll
                for (int t = 0; t < KK; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][t], acc[i][t], 0, 0, 0);
            }
        }
    };
    // two named register sets, no copies: the loads of one set stay in flight under the MFMAs of the other
    // (rows past the slab end have out-of-range offsets -> zeros, so over-running the slab is harmless)
    load_group(p0, a0v, b0v);
    for (long long p = p0; p < p1; p += 2 * U) {
        load_group(p + U, a1v, b1v);
        compute(a0v, b0v);
        load_group(p + 2 * U, a0v, b0v);
        compute(a1v, b1v);
    }
    // partial slab: D row = n (MFMA i index), col = k (j index)
    float* out = a.dw_part + (size_t)g * a.n_out * a.ld_out;
#pragma unroll
    for (int i = 0; i < NN; ++i)
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * (wn * NN + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)n * a.ld_out + 32 * (wk * KK + t) + c] = acc[i][t][r];
            }
    if (a.db_part && wk == 0) {
#pragma unroll
        for (int i = 0; i < NN; ++i) {
            const float s = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (h == 0) a.db_part[(size_t)g * a.n_out + ncol[i]] = s;
        }
    }
}

// The two thin shapes with 16-byte loads on their wide operand.  One buffer_load_b128 hands a lane four consecutive
// columns c0 + 4 i + j of its row: they feed FOUR tiles (tile j takes column 4 i + j as its row/column i — a permutation of
// which output index sits in which tile, undone when the slab is written), so the wide operand costs one load per row
// pair instead of four, in 512-byte pieces.  Waves (ws, wx) = (half of the workgroup's rows, 128-column
// span of the wide operand); the two row halves are added through LDS, one slab per workgroup.
//   WIDE_A: A = dY[M][256] wide (n = 128 wx + 4 i + j), B = X[M][<=64] by scalar loads (2 tiles)      -> [256][64] slab
//   else  : A = dY[M][<=32] by scalar loads (1 tile), B = X[M][256] wide (k = 128 wx + 4 i + j)       -> [32][256] slab
template <bool WIDE_A, bool XH = false>     // XH (narrow-A shape only): X rows hold 256 f16 values in their first 512 bytes
__global__ __launch_bounds__(256, 1) void vfn_dw_thin_kernel(const DwArgs a) {
    constexpr int U = 8;
    constexpr int NA = WIDE_A ? 4 : 1, NB = WIDE_A ? 2 : 4;
    __shared__ float s_red[2][NA * NB * 16 + NA][64];      // the ws = 1 waves' accumulators (and column sums)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ws = wave >> 1, wx = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    const int G = gridDim.x, g = blockIdx.x;
    // the workgroup's slab of row pairs (the same partition as vfn_dw_kernel), split in two halves of whole load groups
    const long long pairs = (a.n_points + 1) / 2;
    const long long per = (pairs + G - 1) / G;
    const long long s0 = g * per, s1 = min(pairs, s0 + per);
    const long long half = (((s1 - s0 + 1) / 2 + 2 * U - 1) / (2 * U)) * (2 * U);
    const long long p0 = ws == 0 ? s0 : min(s1, s0 + half), p1 = ws == 0 ? min(s1, s0 + half) : s1;

    f32x16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[i][t] = splat16(0.f);
    float bsum[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) bsum[i] = 0.f;

    const long long r_base = 2 * p0;
    const long long rows_slab = max(0LL, min(2 * (p1 - p0), a.n_points - r_base));
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy + (size_t)r_base * a.ld_dy), 0, (int)(rows_slab * a.ld_dy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (size_t)r_base * a.ld_x), 0, (int)(rows_slab * a.ld_x * 4), 0x00020000);
    constexpr unsigned OOB = 0x7fffffffu;
    // wide operand: columns 128 wx + 4 c .. + 3 (all valid: the wide side is a full 256-column matrix); narrow: 32 t + c
    const unsigned wide_off = (unsigned)(128 * wx + 4 * c) * 4u;
    unsigned nar_off[WIDE_A ? NB : NA];
#pragma unroll
    for (int t = 0; t < (WIDE_A ? NB : NA); ++t) {
        const int col = 32 * t + c;
        nar_off[t] = col < (WIDE_A ? a.k_valid : a.n_valid) ? (unsigned)col * 4u : OOB;
    }

    float av[2][U][NA], bv[2][U][NB];
    auto load_group = [&](long long p, float (&aa)[U][NA], float (&bb)[U][NB]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned r = (unsigned)(2 * (p - p0 + u) + h);
            const unsigned ra = r * (unsigned)a.ld_dy * 4u, rb = r * (unsigned)a.ld_x * 4u;
            if (WIDE_A) {
                // (cast the whole vector: __builtin_bit_cast on one element of an ext_vector reads element 0)
                const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, ra + wide_off, 0, 0));
#pragma unroll
                for (int i = 0; i < NA; ++i) aa[u][i] = q[i];
#pragma unroll
                for (int t = 0; t < NB; ++t)
                    bb[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, nar_off[t] == OOB ? OOB : rb + nar_off[t], 0, 0));
            } else {
                aa[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dy, nar_off[0] == OOB ? OOB : ra + nar_off[0], 0, 0));
                f32x4 q;
                if (XH) {
                    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                    q = __builtin_convertvector(__builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rs_x, rb + wide_off / 2, 0, 0)), f32x4);
                } else {
                    q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, rb + wide_off, 0, 0));
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) bb[u][t] = q[t];
            }
        }
    };
    auto compute = [&](const float (&aa)[U][NA], const float (&bb)[U][NB]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                bsum[i] += aa[u][i];
#pragma unroll
                for (int t = 0; t < NB; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u][i], bb[u][t], acc[i][t], 0, 0, 0);
            }
    };
    load_group(p0, av[0], bv[0]);
    for (long long p = p0; p < p1; p += 2 * U) {      // rows past the slab read as zero: over-running it is harmless
        load_group(p + U, av[1], bv[1]);
        compute(av[0], bv[0]);
        load_group(p + 2 * U, av[0], bv[0]);
        compute(av[1], bv[1]);
    }
    // the second half's partial sums go through LDS to the first half's waves
    if (ws == 1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[wx][(i * NB + t) * 16 + r][lane] = acc[i][t][r];
            s_red[wx][NA * NB * 16 + i][lane] = bsum[i];
        }
    }
    __syncthreads();
    if (ws == 1) return;
    // D row = A index (r & 3) + 8 (r >> 2) + 4 h, col = B index c; wide tiles: index i of tile j is column 128 wx + 4 i + j
    float* out = a.dw_part + (size_t)g * a.n_out * a.ld_out;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ai = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int n = WIDE_A ? 128 * wx + 4 * ai + i : ai;
                const int k = WIDE_A ? 32 * t + c : 128 * wx + 4 * c + t;
                out[(size_t)n * a.ld_out + k] = acc[i][t][r] + s_red[wx][(i * NB + t) * 16 + r][lane];
            }
    if (a.db_part && (WIDE_A || wx == 0)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float sum = bsum[i] + s_red[wx][NA * NB * 16 + i][lane];
            sum += __shfl_xor(sum, 32, 64);
            const int n = WIDE_A ? 128 * wx + 4 * c + i : c;
            if (h == 0) a.db_part[(size_t)g * a.n_out + n] = sum;
        }
    }
}

int plan_or_error(int kind, const vfn_net_geom* g, VfnNetPlan* p, const char* what) {
    char err[256] = {0};
    int rc = vfn_make_plan(kind, g, p, err, sizeof(err));
    if (rc != VFN_OK) vfn_set_error("%s: %s", what, err);
    return rc;
}

}  // namespace

extern "C" int vfn_mlp_bwd_chain(const vfn_net_geom* vf_geom, const float* vf_packed, const float* vf_packed_bwd,
                                 const vfn_net_geom* rn_geom, const float* rn_packed, const float* rn_packed_bwd,
                                 const float* saved, float* dy, const float* d_colors, const float* colors,
                                 const float* d_vec, const float* vec, const float* d_feats, int32_t vec_stride,
                                 int64_t n_points, float* dz_rgb, float* dz_vec, void* stream) {
    BwdArgs a = {};
    int rc = plan_or_error(VFN_NET_VF, vf_geom, &a.vf, "vfn_mlp_bwd_chain");
    if (rc != VFN_OK) return rc;
    a.fused = rn_geom != nullptr;
    if (a.fused) {
        rc = plan_or_error(VFN_NET_RENDER, rn_geom, &a.rn, "vfn_mlp_bwd_chain");
        if (rc != VFN_OK) return rc;
        VFN_REQUIRE(rn_packed && rn_packed_bwd && d_colors && colors && dz_rgb, "vfn_mlp_bwd_chain: NULL rendering-net argument");
        VFN_REQUIRE(vf_geom->feature_dims == VFN_HIDDEN, "vfn_mlp_bwd_chain: fused mode needs feature_dims == %d", VFN_HIDDEN);
    }
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed && vf_packed_bwd && saved && dy && d_vec && vec && dz_vec, "vfn_mlp_bwd_chain: NULL argument");
    VFN_REQUIRE(vec_stride >= 3, "vfn_mlp_bwd_chain: vec_stride must be >= 3");
    a.vf_w = vf_packed; a.vf_wb = vf_packed_bwd; a.rn_w = rn_packed; a.rn_wb = rn_packed_bwd;
    a.saved = saved; a.dy = dy; a.d_colors = d_colors; a.colors = colors; a.d_vec = d_vec; a.vec = vec;
    a.d_feats = d_feats; a.dz_rgb = dz_rgb; a.dz_vec = dz_vec; a.n_points = n_points; a.vec_stride = vec_stride;
    const long long blocks = (n_points + TM - 1) / TM;
    hipLaunchKernelGGL(vfn_mlp_bwd_kernel, dim3((unsigned)blocks), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_mlp_bwd_chain");
}

// dW'[n][k] = sum_m dY[m][n] X[m][k] as `groups` partial slabs [groups][n_out][ld_out] (+ column sums of dY).
// shape: 0 = 256 x 256 (hidden layer, act inputs), 1 = 256 x 64 (aux inputs, 40 valid), 2 = 32 x 256 (3-channel head)
extern "C" int vfn_weight_grad_partials(int32_t shape, const float* dy, int32_t ld_dy, int32_t n_valid, const float* x,
                                        int32_t ld_x, int32_t k_valid, int64_t n_points, int32_t groups, float* dw_part,
                                        float* db_part, int32_t x_f16, void* stream) {
    VFN_REQUIRE(dy && x && dw_part, "vfn_weight_grad_partials: NULL argument");
    VFN_REQUIRE(groups >= 1 && groups <= 4096, "vfn_weight_grad_partials: groups=%d", groups);
    VFN_REQUIRE(!x_f16 || shape == 2, "vfn_weight_grad_partials: x_f16 is implemented for shape 2 (the 3-channel heads) only");
    DwArgs a = {};
    a.dy = dy; a.x = x; a.dw_part = dw_part; a.db_part = db_part; a.n_points = n_points;
    a.ld_dy = ld_dy; a.n_valid = n_valid; a.ld_x = ld_x; a.k_valid = k_valid;
    hipStream_t s = (hipStream_t)stream;
    if (shape == 0) {
        a.ld_out = 256; a.n_out = 256;
        hipLaunchKernelGGL((vfn_dw_kernel<4, 4, 2, 2>), dim3(groups), dim3(256), 0, s, a);
    } else if (shape == 1) {
        a.ld_out = 64; a.n_out = 256;
        // 16-byte loads on dY when it is a full, aligned 256-column matrix (the shipped layers)
        const bool wide = n_valid == 256 && (ld_dy & 3) == 0 && ((uintptr_t)dy & 15) == 0;
        if (wide) hipLaunchKernelGGL((vfn_dw_thin_kernel<true>), dim3(groups), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((vfn_dw_kernel<2, 2, 4, 1>), dim3(groups), dim3(256), 0, s, a);
    } else if (shape == 2) {
        a.ld_out = 256; a.n_out = 32;
        const bool wide = k_valid == 256 && (ld_x & 3) == 0 && ((uintptr_t)x & 15) == 0;
        VFN_REQUIRE(!x_f16 || wide, "vfn_weight_grad_partials: f16 rows need the full, aligned 256-column X of shape 2");
        if (wide && x_f16) hipLaunchKernelGGL((vfn_dw_thin_kernel<false, true>), dim3(groups), dim3(256), 0, s, a);
        else if (wide) hipLaunchKernelGGL((vfn_dw_thin_kernel<false>), dim3(groups), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((vfn_dw_kernel<1, 2, 1, 4>), dim3(groups), dim3(256), 0, s, a);
    } else {
        vfn_set_error("vfn_weight_grad_partials: unknown shape %d", shape);
        return VFN_ERR_INVALID;
    }
    return vfn_check_launch("vfn_weight_grad_partials");
}
