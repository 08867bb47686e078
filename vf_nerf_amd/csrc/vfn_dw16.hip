// vfn_dw16.hip — weight gradients of the 256 x 256 hidden layers on the bf16 matrix cores.
//
// Same function as vfn_dw_kernel<4,4,2,2> (vfn_mlp_bwd.hip): dW'[n][k] = sum_m dY[m][n] X[m][k] over a slab of points, one
// partial 256 x 256 slab per workgroup, plus db'[n] = sum_m dY[m][n] — the autograd of the Linear layers of
// models/vector_field/vector_field_network.py:177-208 and rendering_network.py:62-108 under
// train/vector_field_nerf_train.py:252.  The fp32 MFMA runs at 1/16 of the bf16 rate, so the product is evaluated on
// split operands: v = hi + lo with hi = bf16_trunc(v), lo = bf16(v - hi) (16 significant bits, fp32's exponent range:
// gradients of 1e-8 need no scaling, which is why this is bf16 and not the forward's f16 split), and
//     dY X ~= dY_hi X_hi + dY_hi X_lo + dY_lo X_hi
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 MFMAs per K=16 block instead of 8 fp32 MFMAs.  The dropped
// lo x lo term and the 16-bit operands leave ~2^-16 relative error per product, unbiased, under a sum over 10^5..10^6
// points; the gradient tests hold the result to 1e-3 of each tensor's magnitude and observe ~1e-5.
//
// Structure: the reduction runs over the POINT index, which is the slow index of both row-major inputs, so both MFMA
// operands need a transpose.  Per step of 32 points the four waves load 32 rows of dY and of X (buffer_load_dwordx4,
// one full 1 KiB row per wave-instruction, rows beyond the slab read as zero), split them in registers and write four
// [32][256] bf16 images (dY|X) x (hi|lo) to LDS with ds_write_b64; the operands are then fetched with the hardware
// transposing read ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group a 4-row x 16-column block
// arrives column-major, two reads give a lane its 8 consecutive points of one column.  Image rows are 576 bytes apart
// (512 + 64): consecutive rows shift by 16 banks, which makes the four rows x two column groups of a 32-lane half hit
// 64 distinct banks.  One wave per SIMD: wave (wn, wk) owns the 128 x 128 block of the gradient (16 accumulator tiles =
// 256 AGPRs); the next step's global loads are in flight under the current step's 96 MFMAs, its split + LDS writes run
// between the two K-blocks; images are double-buffered (144 KiB), one barrier per step.
#include <string.h>
#include "vfn_common.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DW_STEP = 32;                     // points per step (two K-blocks of 16)
constexpr int DW_ROW = 576;                     // bytes between image rows (256 bf16 + 64: 16-bank shift per row)
constexpr int DW_IMG = DW_STEP * DW_ROW;        // one [32][256] bf16 image
constexpr int DW_BUF = 4 * DW_IMG;              // (dY | X) x (hi | lo)

struct Dw16Args {
    const float* dy;      // [M][256]
    const float* x;       // [M][256] fp32, or (XH) 256 f16 values in the first 512 bytes of every 1 KiB row
    float* dw_part;       // [G][256][256]
    float* db_part;       // [G][256] or NULL
    long long n_points;
    int ld_dy, ld_x;      // floats between rows (>= 256, multiples of 4: whole 16-byte pieces); 256 for the dense matrices
    // FOLD (vfn_weight_grad_partials_bf16_fold): x is the previous layer's pre-BatchNorm output z; the operand is
    // fold_post * max(z * scale + shift, 0) on the first fold_n of the 256 columns (fold_coef = [4][fold_n]: scale | shift | ..), fold_post * z behind them
    const float* fold_coef;
    int fold_n;
    float fold_post;
};

typedef __attribute__((address_space(3))) s4 lds_s4;

// 8 consecutive points (K slots) of this lane's column: two transposed 4-row reads
__device__ __forceinline__ bf8 tr_frag(const unsigned char* img, int off0) {
    const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + off0));
    const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + off0 + 4 * DW_ROW));
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf8, v);
}

// four fp32 values -> 4 bf16 "hi" (truncated: exactly representable, so v - hi is exact) and 4 bf16 "lo" (rounded)
__device__ __forceinline__ void split4(const f32x4v v, uint2& hi, uint2& lo) {
    // (through the integer vector: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    const unsigned u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
    hi.x = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    hi.y = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
    const f32x2v r01 = {v[0] - __builtin_bit_cast(float, u0 & 0xffff0000u), v[1] - __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const f32x2v r23 = {v[2] - __builtin_bit_cast(float, u2 & 0xffff0000u), v[3] - __builtin_bit_cast(float, u3 & 0xffff0000u)};
    lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf2));
    lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf2));
}

template <bool XH, bool FOLD = false>
__global__ __launch_bounds__(256, 1) void vfn_dw16_kernel(const Dw16Args a) {
    static_assert(!(XH && FOLD), "the fold reads fp32 pre-activations");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * DW_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;                 // this wave's 128 x 128 block of the gradient
    const int G = gridDim.x, g = blockIdx.x;
    const long long steps = (a.n_points + DW_STEP - 1) / DW_STEP;
    const long long per = (steps + G - 1) / G;
    const long long s0 = g * per, s1 = min(steps, s0 + per);

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                    // columns 4*lane .. 4*lane+3 of dY over the rows this wave stages

    // slab-relative descriptors: rows past the end of the batch read as zero
    const long long r_base = s0 * DW_STEP;
    const long long rows_slab = max(0LL, min((s1 - s0) * DW_STEP, a.n_points - r_base));
    const int sdy = a.ld_dy * 4, sx = a.ld_x * 4;            // bytes between rows
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (size_t)r_base * a.ld_dy), 0,
                                                                           (int)(rows_slab * sdy), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)r_base * a.ld_x), 0,
                                                                          (int)(rows_slab * sx), 0x00020000);
    [[maybe_unused]] f32x4v f_sc = {1.f, 1.f, 1.f, 1.f}, f_sh = {0.f, 0.f, 0.f, 0.f}, f_lo = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FOLD) {                                    // a lane stages columns 4 lane .. 4 lane + 3 of every row: its coefficients once
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * lane + c;
            const bool bn = k < a.fold_n;
            f_sc[c] = bn ? a.fold_coef[k] : 1.0f;
            f_sh[c] = bn ? a.fold_coef[a.fold_n + k] : 0.0f;
            f_lo[c] = bn ? 0.0f : -__builtin_inff();
        }
    }
    u32x4 ld_dy[8], ld_x[8];                                 // this wave's 8 rows of the step in flight
    auto issue = [&](long long s) {
        const int row0 = (int)(s - s0) * DW_STEP + 8 * wave;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            ld_dy[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, lane * 16, (row0 + r) * sdy, 0);
            if (XH) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs_x, lane * 8, (row0 + r) * sx, 0);
                ld_x[r] = u32x4{h[0], h[1], 0u, 0u};
            } else {
                ld_x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane * 16, (row0 + r) * sx, 0);
            }
        }
    };
    auto stage = [&](int buf) {                              // split + write this wave's rows into the images of `buf`
        unsigned char* base = lds + buf * DW_BUF + (8 * wave) * DW_ROW + lane * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x4v vd = __builtin_bit_cast(f32x4v, ld_dy[r]);
            f32x4v vx;
            if (XH) {
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 h = {ld_x[r][0], ld_x[r][1]};
                vx = __builtin_convertvector(__builtin_bit_cast(half4, h), f32x4v);
            } else {
                vx = __builtin_bit_cast(f32x4v, ld_x[r]);
            }
            if constexpr (FOLD) {     // the expression of vfn_bstat_relu_rows, per value (rows past the slab's end read z = 0: their dY is zero)
                // (the product must be ROUNDED to fp32 before the split below, as the stored activation is: left to -ffp-contract the multiply
                //  fuses into split4's  v - hi  — an fma on the unrounded product — and the low halves differ.  The empty asm pins the value.)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = a.fold_post * fmaxf(fmaf(vx[c], f_sc[c], f_sh[c]), f_lo[c]);
                    asm volatile("" : "+v"(t));
                    vx[c] = t;
                }
            }
            uint2 hi, lo;
            split4(vd, hi, lo);
            *reinterpret_cast<uint2*>(base + 0 * DW_IMG + r * DW_ROW) = hi;
            *reinterpret_cast<uint2*>(base + 1 * DW_IMG + r * DW_ROW) = lo;
            split4(vx, hi, lo);
            *reinterpret_cast<uint2*>(base + 2 * DW_IMG + r * DW_ROW) = hi;
            *reinterpret_cast<uint2*>(base + 3 * DW_IMG + r * DW_ROW) = lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) bsum[c] += vd[c];
        }
    };
    // transposed-read address of this lane inside a 32-column tile and a 16-row K-block (T10): lane 4q+p of a 16-lane
    // group supplies row q, columns 4p..4p+3; groups 0/1 = columns 0-15 / 16-31, lane halves = rows +0 / +8
    const int q = (lane & 15) >> 2, p = lane & 3, cg = (lane >> 4) & 1, h = lane >> 5;
    const int tr_off = (8 * h + q) * DW_ROW + (16 * cg + 4 * p) * 2;

    auto mma = [&](int buf, int kb) {
        const unsigned char* img = lds + buf * DW_BUF + kb * 16 * DW_ROW + tr_off;
        bf8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = tr_frag(img + 0 * DW_IMG, (4 * wn + i) * 64);
            al[i] = tr_frag(img + 1 * DW_IMG, (4 * wn + i) * 64);
            bh[i] = tr_frag(img + 2 * DW_IMG, (4 * wk + i) * 64);
            bl[i] = tr_frag(img + 3 * DW_IMG, (4 * wk + i) * 64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[t], acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[t], acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[t], acc[i][t], 0, 0, 0);
            }
    };

    if (s0 < s1) {
        issue(s0);
        stage(0);
        __syncthreads();
    }
    for (long long s = s0; s < s1; ++s) {
        const int buf = (int)(s - s0) & 1;
        const bool more = s + 1 < s1;
        if (more) issue(s + 1);
        mma(buf, 0);
        if (more) stage(buf ^ 1);       // the other buffer was last read in step s-1; every wave passed that step's barrier
        mma(buf, 1);
        __syncthreads();
    }

    // partial slab: D row = n (A operand's row), column = k (B operand's column)
    float* out = a.dw_part + (size_t)g * 256 * 256;
    const int c = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * (4 * wn + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)n * 256 + 32 * (4 * wk + t) + c] = acc[i][t][r];
            }
    if (a.db_part) {                   // column sums of dY: four waves each hold the sums of the rows they staged
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) red[wave * 256 + 4 * lane + cc] = bsum[cc];
        __syncthreads();
        a.db_part[(size_t)g * 256 + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
    }
}

}  // namespace

extern "C" int vfn_weight_grad_partials_bf16_fold(const float* dy, int32_t ld_dy, const float* z_prev, int32_t ldz, const float* coef_prev,
                                                  int32_t n_prev, float post_prev, int64_t n_points, int32_t groups, float* dw_part,
                                                  float* db_part, void* stream) {
    VFN_REQUIRE(dy && z_prev && coef_prev && dw_part, "vfn_weight_grad_partials_bf16_fold: NULL argument");
    VFN_REQUIRE(groups >= 1 && groups <= 4096 && n_prev >= 1 && n_prev <= 256, "vfn_weight_grad_partials_bf16_fold: groups=%d n_prev=%d", groups, n_prev);
    VFN_REQUIRE(ld_dy >= 256 && ldz >= 256 && (ld_dy & 3) == 0 && (ldz & 3) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)z_prev & 15) == 0,
                "vfn_weight_grad_partials_bf16_fold: 256 columns in whole 16-byte pieces of every row (ld_dy=%d, ldz=%d)", ld_dy, ldz);
    VFN_REQUIRE(n_points >= 0 && n_points * (int64_t)(ld_dy > ldz ? ld_dy : ldz) * 4 < (int64_t)groups << 31,
                "vfn_weight_grad_partials_bf16_fold: slab larger than 2 GiB");
    Dw16Args a = {};
    a.dy = dy; a.x = z_prev; a.dw_part = dw_part; a.db_part = db_part; a.n_points = n_points; a.ld_dy = ld_dy; a.ld_x = ldz;
    a.fold_coef = coef_prev; a.fold_n = n_prev; a.fold_post = post_prev;
    hipLaunchKernelGGL((vfn_dw16_kernel<false, true>), dim3(groups), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_weight_grad_partials_bf16_fold");
}

extern "C" int vfn_weight_grad_partials_bf16(const float* dy, const float* x, int64_t n_points, int32_t groups,
                                             float* dw_part, float* db_part, int32_t x_f16, void* stream) {
    return vfn_weight_grad_partials_bf16_ld(dy, 256, x, 256, n_points, groups, dw_part, db_part, x_f16, stream);
}

extern "C" int vfn_weight_grad_partials_bf16_ld(const float* dy, int32_t ld_dy, const float* x, int32_t ld_x, int64_t n_points, int32_t groups,
                                                float* dw_part, float* db_part, int32_t x_f16, void* stream) {
    VFN_REQUIRE(dy && x && dw_part, "vfn_weight_grad_partials_bf16: NULL argument");
    VFN_REQUIRE(groups >= 1 && groups <= 4096, "vfn_weight_grad_partials_bf16: groups=%d", groups);
    VFN_REQUIRE(ld_dy >= 256 && ld_x >= 256 && (ld_dy & 3) == 0 && (ld_x & 3) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0,
                "vfn_weight_grad_partials_bf16: 256 columns in whole 16-byte pieces of every row (ld_dy=%d, ld_x=%d)", ld_dy, ld_x);
    VFN_REQUIRE(!x_f16 || ld_x == 256, "vfn_weight_grad_partials_bf16: the f16 rows of X are 1 KiB apart");
    VFN_REQUIRE(n_points >= 0 && n_points * (int64_t)(ld_dy > ld_x ? ld_dy : ld_x) * 4 < (int64_t)groups << 31,
                "vfn_weight_grad_partials_bf16: slab larger than 2 GiB");
    Dw16Args a = {};
    a.dy = dy; a.x = x; a.dw_part = dw_part; a.db_part = db_part; a.n_points = n_points; a.ld_dy = ld_dy; a.ld_x = ld_x;
    if (x_f16) hipLaunchKernelGGL(vfn_dw16_kernel<true>, dim3(groups), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(vfn_dw16_kernel<false>, dim3(groups), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_weight_grad_partials_bf16");
}
