// vfn_mlp_core.h — device helpers shared by the forward (vfn_mlp.hip) and backward (vfn_mlp_bwd.hip) MLP kernels:
// LDS tile geometry, swizzle, the fp32-MFMA K-segment loop, accumulator stores and the 16x16x4 head.
#pragma once
#include <hip/hip_runtime.h>
#include "vfn_plan.h"

namespace vfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = VFN_TM;       // 64 rows per workgroup
constexpr int ACT_LD = 256;      // floats per activation row
constexpr int AUX_LD = 44;       // floats per aux row (40 used; 44 keeps ds_read_b128 conflict-free)
constexpr int NTHREADS = 256;
constexpr int SMEM_FLOATS = TM * ACT_LD + TM * AUX_LD + TM * 3 + TM * 3;
enum : int { ACT_RELU = 0, ACT_TANH = 1, ACT_NONE = 2 };

// float index of (row, col) inside the swizzled activation tile: 16-byte chunks of a row are
// XOR-ed with (row & 15) so that 16 lanes reading the same logical chunk of 16 different rows hit
// 16 different 4-bank groups (ds_read_b128), and a half-wave writing 32 consecutive columns of
// one row stays conflict-free (ds_write_b32).
__device__ __forceinline__ int act_idx(int row, int col) {
    return row * ACT_LD + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3));
}

__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = v;
    return r;
}

// One K segment: nkb blocks of 8 k's.  A fragments come from LDS (act: swizzled, aux: linear),
// B fragments from the packed weight stream of this wave's NT column tiles.
template <int NT, bool SWZ>
__device__ __forceinline__ void mma_segment(f32x16 (&acc)[2][2], const float* __restrict__ lds, int nkb,
                                            const f32x4* __restrict__ w0, const f32x4* __restrict__ w1, int lane) {
    const int r = lane & 31;
    const int h = lane >> 5;
    const int ld = SWZ ? ACT_LD : AUX_LD;
    const float* row0 = lds + r * ld;
    const float* row1 = lds + (r + 32) * ld;
    const int sw = SWZ ? (lane & 15) : 0;

    auto load_a = [&](int kb, f32x4& a0, f32x4& a1) {
        const int ch = ((2 * kb + h) ^ sw) << 2;
        a0 = *reinterpret_cast<const f32x4*>(row0 + ch);
        a1 = *reinterpret_cast<const f32x4*>(row1 + ch);
    };
    f32x4 a0, a1, b0, b1 = {0.f, 0.f, 0.f, 0.f};
    if (nkb <= 0) return;
    load_a(0, a0, a1);
    b0 = w0[lane];
    if (NT == 2) b1 = w1[lane];
    for (int kb = 0; kb < nkb; ++kb) {
        f32x4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
        if (kb + 1 < nkb) {
            load_a(kb + 1, na0, na1);
            nb0 = w0[(kb + 1) * 64 + lane];
            if (NT == 2) nb1 = w1[(kb + 1) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            if (NT == 2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            if (NT == 2) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

// acc <- bias + A[64 x K] * W'^T for this wave's column tiles of one hidden layer.
template <int NT>
__device__ __forceinline__ void layer_mma(f32x16 (&acc)[2][2], const VfnLayerPlan& lp, const float* __restrict__ wbase,
                                          const float* s_act, const float* s_aux, int tile0, int lane) {
    const int kbt = lp.nkb_act + lp.nkb_aux;
    const float* bias = wbase + lp.b_off;
    const float bv0 = bias[tile0 * 32 + (lane & 31)];
    const float bv1 = (NT == 2) ? bias[(tile0 + 1) * 32 + (lane & 31)] : 0.f;
    acc[0][0] = splat16(bv0); acc[1][0] = splat16(bv0);
    acc[0][1] = splat16(bv1); acc[1][1] = splat16(bv1);
    const f32x4* w0 = reinterpret_cast<const f32x4*>(wbase + lp.w_off) + (size_t)tile0 * kbt * 64;
    const f32x4* w1 = w0 + (size_t)kbt * 64;
    mma_segment<NT, true>(acc, s_act, lp.nkb_act, w0, w1, lane);
    mma_segment<NT, false>(acc, s_aux, lp.nkb_aux, w0 + (size_t)lp.nkb_act * 64, w1 + (size_t)lp.nkb_act * 64, lane);
}

__device__ __forceinline__ float act_fn(float v, int kind) {
    return kind == ACT_RELU ? fmaxf(v, 0.f) : (kind == ACT_TANH ? tanhf(v) : v);
}

// Store this wave's accumulators into the activation tile (after the workgroup has finished
// reading it), applying the activation.  D layout of v_mfma_f32_32x32x2_f32:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <int NT>
__device__ __forceinline__ void store_tile_lds(const f32x16 (&acc)[2][2], float* s_act, int tile0, int lane, int kind) {
    int c = lane & 31, h = lane >> 5;
    asm volatile("" : "+v"(c), "+v"(h));  // keep the lane-constant offsets from being hoisted out of the layer loop
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                s_act[act_idx(row, 32 * (tile0 + nt) + c)] = act_fn(acc[mt][nt][r], kind);
            }
}

// Same accumulators to global memory: out[(row0 + row) * stride + col_off + col].
template <int NT>
__device__ __forceinline__ void store_tile_global(const f32x16 (&acc)[2][2], float* out, long long row0, long long n_rows,
                                                  int stride, int col_off, int tile0, int lane, int kind) {
    int c = lane & 31, h = lane >> 5;
    asm volatile("" : "+v"(c), "+v"(h));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < n_rows) out[row * stride + col_off + 32 * (tile0 + nt) + c] = act_fn(acc[mt][nt][r], kind);
            }
}

// 3-channel head on v_mfma_f32_16x16x4_f32: wave w owns rows 16w..16w+15, K = 256 from the act tile.
// Returns D (col = lane & 15 = channel, row = 16w + 4*(lane >> 4) + reg).
__device__ __forceinline__ f32x4 head_mma(const VfnNetPlan& np, const float* __restrict__ wbase, const float* s_act,
                                          int wave, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const int row = 16 * wave + i;
    const float* arow = s_act + row * ACT_LD;
    const int sw = row & 15;
    const f32x4* w = reinterpret_cast<const f32x4*>(wbase + np.head_w_off);
    const float bv = wbase[np.head_b_off + i];
    f32x4 d0 = {bv, bv, bv, bv};
    f32x4 d1 = {0.f, 0.f, 0.f, 0.f};
    const int nkb = (int)np.head_nkb16;
    f32x4 a = *reinterpret_cast<const f32x4*>(arow + (((q) ^ sw) << 2));
    f32x4 b = w[lane];
    for (int kb = 0; kb < nkb; ++kb) {
        f32x4 na = a, nb = b;
        if (kb + 1 < nkb) {
            na = *reinterpret_cast<const f32x4*>(arow + (((4 * (kb + 1) + q) ^ sw) << 2));
            nb = w[(kb + 1) * 64 + lane];
        }
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], d1, 0, 0, 0);
        a = na; b = nb;
    }
    return d0 + d1;
}


}  // namespace vfn
