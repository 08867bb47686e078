// vfn_dwf.hip — weight gradients of the training step from the FRAGMENT-ORDERED workspace.
//
// dW'[n][k] = sum_m dY[m][n] X[m][k], db'[n] = sum_m dY[m][n] per slab of points — the autograd of the Linear layers of
// models/vector_field/vector_field_network.py:177-208 and rendering_network.py:62-108 under
// train/vector_field_nerf_train.py:252 — like vfn_dw16.hip (256 x 256 layers, split-bf16 products on
// v_mfma_f32_32x32x16_bf16, LDS images read back with the transposing ds_read_b64_tr_b16), but for the layout the f16x3
// training forward and the bf16 dX chain now WRITE their tiles in:
//
//   slot s, group G = m >> 5 (the 32 points of one wave), 32 KiB per group:
//       piece (t, q) = registers 4q .. 4q+3 of output tile t, exactly as the producing wave holds them:
//       [4 t + q][lane 0..63][16 B]   lane = 32 g + i  ->  point 32 G + i, columns 32 t + 8 q + 4 g .. + 3        (fp32)
//       [4 t + q][lane 0..63][ 8 B]   the same four values as f16 (activations) or bf16 (gradients)               (16 bit)
//   so that every store instruction of the producers and every load instruction here moves 1 KiB (512 B) of consecutive
//   bytes; the row-major layout made each store touch 32 lines with 32 (16) bytes each (Backward).
//
// The transposing read wants image rows 576 bytes apart (rows shift by 16 banks); fragment-ordered pieces arrive with the
// POINT on the lane index, i.e. a wave-instruction writes 32 different image rows at one column group, which on that
// stride is an 8-way bank conflict.  So the 8-byte chunks of every image row are XOR-swizzled with the row number:
// chunk' = chunk ^ ((row >> 1) & 7) — a permutation inside each aligned run of 8 chunks, which keeps the transposed reads
// conflict-free (each 32-lane half still covers 4 rows x 16 distinct dwords, 16 banks apart) and makes the writes
// conflict-free too (16 lanes = 8 row pairs x 2 bank halves).
//
// One kernel, three shapes (what vfn_weight_grad_partials calls shapes 0 / 1 / 2):
//   0   256 x 256    A = dY (8 tiles), B = X (8 tiles): wave (wn, wk) owns 4 x 4 tiles
//   1   256 x 64     A = dY (8 tiles), B = the encoding tile aux[M][40] (row-major fp32, zero-padded to 2 tiles): 2 x 2 per wave
//   2    32 x 256    A = the head's pre-activation gradient dz[M][4] (row-major fp32, 1 tile), B = X (8 tiles): 1 x 2 per wave
// and the operand forms: dY fragment fp32 | fragment bf16 | dz rows; X fragment fp32 | fragment f16 | [M][256] fp32 rows (the
// tanh'ed features, which stay row-major) | aux rows.  A bf16 dY has no low half (two products per K-block instead of
// three); an f16 X splits exactly into bf16 hi + lo.
//
// dY form 3, "scaled f16" (what the chain writes with dy_flags bit 3): the 16 values a lane holds of a gradient tile (one point,
// 16 of the tile's 32 columns: the lane's accumulator registers) are stored as f16(dY * 2^k) with k chosen per lane and tile so
// that the largest magnitude lands in [2^14, 2^15): 11 significant bits whatever the gradient's scale, no overflow, no state
// carried between steps.  The four 512-byte pieces of tile t sit where the bf16 form has them; byte 16384 + 64 t + lane of the
// group holds k + 113 (255: all 16 values are zero).  A slab's workgroup reads the exponent bytes of its groups first, takes K =
// the smallest k among them, multiplies every value by 2^(K - k) <= 1 on its way into LDS (values 2^-24 below the slab's largest
// vanish — they could not move a sum that is compared to the tensor's largest entry) and multiplies its result by 2^-K:
// operands with ONE common scale, so the products run on v_mfma_f32_32x32x16_f16 with f16 activations as they are — one product
// per K-block, exact 11 x 11-bit operands — and an fp32 X (feature rows, encoding tile) as f16 hi + lo (two products).
//
// Steps.  A step stages one workspace group (32 points = two K-blocks of 16) of both operands into one of two LDS buffers while the
// matrix cores work on the other; the next step's pieces are requested right after the split, so they have two K-blocks and a
// barrier to land in (the stream is latency-bound per CU).  The default training form (dY form 3 x f16 activations) has single
// images — no lo halves — so a buffer holds 64 rows of each and a step takes TWO groups: 64 KiB per workgroup in flight out of the
// same registers and LDS (WIDE in the kernel; 560 -> 512 us per batched 256 x 256 launch at 524 288 points).
#include <string.h>
#include <type_traits>
#include "vfn_common.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int F_STEP = 32;                      // points per step = one workspace group (two K-blocks of 16)
constexpr int F_ROW = 576;                      // bytes between image rows
constexpr int F_IMG = F_STEP * F_ROW;           // one [32][<=256] bf16 image
constexpr int F_BUF = 4 * F_IMG;                // (dY | X) x (hi | lo)
constexpr int F_GROUP = 32768;                  // bytes of one workspace group

enum : int { X_FRAG32 = 0, X_FRAG16 = 1, X_ROWS32 = 2, X_AUX40 = 3 };
enum : int { DY_FRAG32 = 0, DY_FRAGBF16 = 1, DY_DZ4 = 2, DY_FRAGF16S = 3 };
constexpr int F_EXP_OFF = 16384;                // exponent bytes of a group (form 3): [tile 0..7][lane 0..63], behind its 16 KiB of 16-bit pieces

struct DwfArgs {
    const void* dy;
    const void* x;
    float* dw_part;       // [G][n_out][ld_out]
    float* db_part;       // [G][n_out] or NULL
    long long n_points;
};
// Several products of the SAME shape and operand forms in one launch (blockIdx.y picks the product, blockIdx.x its slab): a net's
// 256 x 256 weight gradients need not be one launch of 256 slabs each — eight of them as ONE launch of 8 x 32 workgroups fill
// the chip just the same and write an eighth of the partial slabs (64 MB per product and launch otherwise, read back by the un-fold).
constexpr int DWF_BATCH = 8;
struct DwfBatch { DwfArgs v[DWF_BATCH]; const int* n_dev; };   // n_dev: optional device-side point count (clamps every product's n_points)

typedef __attribute__((address_space(3))) s4 lds_s4;

__device__ __forceinline__ bf8 tr_frag(const unsigned char* img, int off1, int off2) {
    const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + off1));
    const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + off2));
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf8, v);
}

// four fp32 values -> 4 bf16 "hi" (truncated: v - hi is exact) and 4 bf16 "lo" (rounded)
__device__ __forceinline__ void split4(const f32x4v v, uint2& hi, uint2& lo) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    const unsigned u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
    hi.x = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    hi.y = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
    const f32x2v r01 = {v[0] - __builtin_bit_cast(float, u0 & 0xffff0000u), v[1] - __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const f32x2v r23 = {v[2] - __builtin_bit_cast(float, u2 & 0xffff0000u), v[3] - __builtin_bit_cast(float, u3 & 0xffff0000u)};
    lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf2));
    lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf2));
}

// four fp32 values -> 4 f16 "hi" (rounded) and 4 f16 "lo" (the residual, rounded)
__device__ __forceinline__ void split4h(const f32x4v v, uint2& hi, uint2& lo) {
    const half4 h = __builtin_convertvector(v, half4);
    const f32x4v r = v - __builtin_convertvector(h, f32x4v);
    const half4 l = __builtin_convertvector(r, half4);
    const u32x2 hu = __builtin_bit_cast(u32x2, h), lu = __builtin_bit_cast(u32x2, l);
    hi.x = hu[0]; hi.y = hu[1]; lo.x = lu[0]; lo.y = lu[1];
}

// byte offset of 8-byte chunk `chunk` of image row `row` (chunks XOR-swizzled inside aligned runs of 8)
__device__ __forceinline__ int img_off(int row, int chunk) { return row * F_ROW + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int SHAPE, int XM, int DM>
__global__ __launch_bounds__(256, 1) void vfn_dwf_kernel(const DwfBatch batch) {
    DwfArgs a = batch.v[blockIdx.y];
    if (batch.n_dev) {          // the live number of points is known to the device only (csrc/vfn_train.hip): slabs are cut from THAT count
        const long long nd = (long long)*batch.n_dev;
        a.n_points = nd < a.n_points ? nd : a.n_points;
    }
    constexpr bool A_FRAG = DM != DY_DZ4;
    constexpr bool B_FRAG = XM == X_FRAG32 || XM == X_FRAG16;
    constexpr bool F16 = DM == DY_FRAGF16S;                     // tile-scaled f16 gradient: f16 matrix instruction, one common scale per slab
    constexpr bool A_LO = DM != DY_FRAGBF16 && !F16;            // a 16-bit gradient has no low half
    constexpr bool B_LO = !(F16 && XM == X_FRAG16);             // f16 activations enter an f16 product as they are
    // the default training form (tile-scaled f16 gradient x f16 activations: 32 KiB of pieces per group and operand pair... 16 KiB each)
    // takes TWO workspace groups per step: its images are single (no low halves), so a buffer holds 64 rows of each, and a step
    // requests 64 KiB per workgroup instead of 32 — the stream is latency-bound per CU
    constexpr bool WIDE = SHAPE == 0 && F16 && XM == X_FRAG16;
    constexpr int GPS = WIDE ? 2 : 1;                           // groups per step
    constexpr int NA = SHAPE == 0 ? 4 : (SHAPE == 1 ? 2 : 1);   // A / B tiles per wave
    constexpr int NB = SHAPE == 0 ? 4 : 2;
    constexpr int LD_OUT = SHAPE == 1 ? 64 : 256, N_OUT = SHAPE == 2 ? 32 : 256;
    static_assert((SHAPE == 1) == (XM == X_AUX40) && (SHAPE == 2) == (DM == DY_DZ4), "shape / operand form mismatch");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * F_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, g = blockIdx.x;
    const long long groups_total = (a.n_points + F_STEP - 1) / F_STEP;
    const long long steps = (groups_total + GPS - 1) / GPS;
    const long long per = (steps + G - 1) / G;
    const long long s0 = g * per, s1 = min(steps, s0 + per);
    const int at0 = SHAPE == 0 ? 4 * (wave >> 1) : (SHAPE == 1 ? 2 * wave : 0);     // first A / B tile of this wave
    const int bt0 = SHAPE == 0 ? 4 * (wave & 1) : (SHAPE == 1 ? 0 : 2 * wave);

    f32x16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    float bsum[A_FRAG ? 8 : 1][4];                              // column sums of dY over this lane's points, per staged piece
#pragma unroll
    for (int r = 0; r < (A_FRAG ? 8 : 1); ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) bsum[r][c] = 0.f;

    // images with fewer than 256 columns: the columns nobody stages must read as zero
    if (XM == X_AUX40 || DM == DY_DZ4) {
        for (int i = tid; i < 2 * F_BUF / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
    }

    const long long n_steps = max(0LL, s1 - s0);
    const long long g_base = s0 * GPS;                                              // first group of the slab
    const long long n_groups = max(0LL, min(groups_total, s1 * GPS) - g_base);      // groups it holds (the last step may hold one)
    // form 3: the smallest tile exponent of this slab = the common scale of its operands
    int bmin = 255;
    if constexpr (F16) {
        const unsigned char* gp = static_cast<const unsigned char*>(a.dy) + (size_t)g_base * F_GROUP + F_EXP_OFF;
        for (long long idx = tid; idx < n_groups * 32; idx += 256) {             // 512 exponent bytes per group = 32 x 16 bytes
            const uint4 e = *reinterpret_cast<const uint4*>(gp + (size_t)(idx >> 5) * F_GROUP + (idx & 31) * 16);
            const unsigned w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) bmin = min(bmin, (int)((w[j] >> (8 * k)) & 0xffu));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bmin = min(bmin, __shfl_xor(bmin, o, 64));
        int* red = reinterpret_cast<int*>(lds);
        __syncthreads();
        if (lane == 0) red[wave] = bmin;
        __syncthreads();
        bmin = __builtin_amdgcn_readfirstlane(min(min(red[0], red[1]), min(red[2], red[3])));
        __syncthreads();
        if (XM == X_AUX40) {            // (the reduction used the first bytes of the zero-filled images)
            if (tid < 4) red[tid] = 0;
            __syncthreads();
        }
    }
    // slab-relative descriptors: groups / rows past the end of the slab read as zero
    const long long r_base = g_base * F_STEP;
    const long long rows_slab = max(0LL, min(n_steps * F_STEP, a.n_points - r_base));
    const __amdgpu_buffer_rsrc_t rs_a = A_FRAG
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(static_cast<const unsigned char*>(a.dy)) + (size_t)g_base * F_GROUP, 0,
                                            (int)(n_groups * F_GROUP), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(static_cast<const unsigned char*>(a.dy)) + (size_t)r_base * 16, 0,
                                            (int)(rows_slab * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = B_FRAG
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(static_cast<const unsigned char*>(a.x)) + (size_t)g_base * F_GROUP, 0,
                                            (int)(n_groups * F_GROUP), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(static_cast<const unsigned char*>(a.x)) +
                                                (size_t)r_base * (XM == X_AUX40 ? 160 : 1024), 0,
                                            (int)(rows_slab * (XM == X_AUX40 ? 160 : 1024)), 0x00020000);

    // The thin shapes (the encoding tile's 256 x 64, the head's 32 x 256) move 16-20 KiB per step and workgroup and multiply for ~130
    // cycles: with one step's pieces in flight they are latency-bound (the head's launch read its 322 MB at 1.1 TB/s).  Their accumulators
    // are small, so they keep a RING of four register sets: three steps' pieces are on their way while one is split and multiplied.
    constexpr int RING = SHAPE != 0 ? 4 : 1;
    u32x4 ld_a_all[RING][A_FRAG ? 8 * GPS : 1], ld_b_all[RING][(B_FRAG || XM == X_ROWS32) ? 8 * GPS : 2];
    unsigned ld_e_all[RING][2 * GPS] = {};                      // form 3: this lane's exponent bytes of the wave's two tiles (2 wave, 2 wave + 1)
    // ``part``: 1 = the dY pieces (and their exponent bytes), 2 = the X pieces, 3 = both.  The default training form (WIDE) requests and
    // splits the two operands half a step apart (STAGGER below), every other form both together.  ``setc``: which register set.
    auto issue = [&](auto setc, long long s, int part = 3) {
        auto& ld_a = ld_a_all[decltype(setc)::value];
        auto& ld_b = ld_b_all[decltype(setc)::value];
        auto& ld_e = ld_e_all[decltype(setc)::value];
        const int st = (int)(s - s0) * GPS;                     // slab-relative group (a group past the slab's end reads as zero)
        if (part & 1) {
        if constexpr (F16) {
#pragma unroll
            for (int gq = 0; gq < GPS; ++gq) {
                ld_e[2 * gq + 0] = __builtin_amdgcn_raw_buffer_load_b8(rs_a, lane, (st + gq) * F_GROUP + F_EXP_OFF + (2 * wave) * 64, 0);
                ld_e[2 * gq + 1] = __builtin_amdgcn_raw_buffer_load_b8(rs_a, lane, (st + gq) * F_GROUP + F_EXP_OFF + (2 * wave + 1) * 64, 0);
            }
        }
        if constexpr (A_FRAG) {
#pragma unroll
            for (int gq = 0; gq < GPS; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int pc = 8 * wave + r;
#ifdef VFN_DWF_PROBE_HALF_DY
                // TIMING PROBE (tools/probe_dwf_half_dy.sh; never in the product build): every second dY piece is not requested at all — the bytes
                // an ideal, decode-free compaction of the ReLU-masked half of dY would save.  Results are wrong by construction.
                if (r & 1) { ld_a[8 * gq + r] = u32x4{0u, 0u, 0u, 0u}; continue; }
#endif
                if (DM == DY_FRAG32) ld_a[8 * gq + r] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, (st + gq) * F_GROUP + pc * 1024, 0);
                else { const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs_a, lane * 8, (st + gq) * F_GROUP + pc * 512, 0); ld_a[8 * gq + r] = u32x4{h[0], h[1], 0u, 0u}; }
            }
        } else {
            ld_a[0] = tid < 32 ? __builtin_amdgcn_raw_buffer_load_b128(rs_a, tid * 16, st * F_STEP * 16, 0) : u32x4{0u, 0u, 0u, 0u};
        }
        }
        if (!(part & 2)) return;
        if constexpr (B_FRAG) {
#pragma unroll
            for (int gq = 0; gq < GPS; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int pc = 8 * wave + r;
                if (XM == X_FRAG32) ld_b[8 * gq + r] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, lane * 16, (st + gq) * F_GROUP + pc * 1024, 0);
                else { const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs_b, lane * 8, (st + gq) * F_GROUP + pc * 512, 0); ld_b[8 * gq + r] = u32x4{h[0], h[1], 0u, 0u}; }
            }
        } else if constexpr (XM == X_ROWS32) {
#pragma unroll
            for (int r = 0; r < 8; ++r) ld_b[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, lane * 16, (st * F_STEP + 8 * wave + r) * 1024, 0);
        } else {            // aux rows: 32 x 40 floats = 320 16-byte chunks
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 256 * k;
                const int row = idx / 10, ch = idx - 10 * row;
                ld_b[k] = idx < 320 ? __builtin_amdgcn_raw_buffer_load_b128(rs_b, (st * F_STEP + row) * 160 + ch * 16, 0, 0) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    const int gi = lane >> 5, pi = lane & 31;                   // fragment pieces: lane half (column group) and point inside the group
    auto stage = [&](auto setc, int buf, long long s, int part = 3, bool valid = true) {      // split + write what this thread loaded into the images of `buf`
                                                                                              // (valid = false: a step past the slab's end -> zeros)
        auto& ld_a = ld_a_all[decltype(setc)::value];
        auto& ld_b = ld_b_all[decltype(setc)::value];
        auto& ld_e = ld_e_all[decltype(setc)::value];
        unsigned char* base = lds + buf * F_BUF;
        if (part & 1) {
        if constexpr (A_FRAG) {
#pragma unroll
            for (int gq = 0; gq < GPS; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool live = valid && (s * GPS + gq) * F_STEP + pi < a.n_points;    // the last group may be partial: its missing points count as zero
                u32x4& la = ld_a[8 * gq + r];
                const int pc = 8 * wave + r, t = pc >> 2, q = pc & 3;
                f32x4v v;
                if (DM == DY_FRAG32) v = __builtin_bit_cast(f32x4v, la);
                else if (F16) {
                    // rescale from the lane's own exponent to the slab's: 2^(bmin - b) <= 1, exact in f16 down to 2^-24
                    const int b = (int)(ld_e[2 * gq + (r >> 2)] & 0xffu);
                    const int d = b - bmin;
                    const _Float16 f = (b == 255 || d > 24) ? (_Float16)0.f : (_Float16)__builtin_bit_cast(float, (unsigned)(127 - d) << 23);
                    const half4 hv = __builtin_bit_cast(half4, u32x2{la[0], la[1]}) * half4{f, f, f, f};
                    const u32x2 hu = __builtin_bit_cast(u32x2, hv);
                    la[0] = hu[0]; la[1] = hu[1];
                    v = __builtin_convertvector(hv, f32x4v);
                }
                else { const u32x2 h = {la[0], la[1]}; v = __builtin_convertvector(__builtin_bit_cast(bf4, h), f32x4v); }
                if (!live) v = f32x4v{0.f, 0.f, 0.f, 0.f};
                const int off = img_off(32 * gq + pi, 8 * t + 2 * q + gi);
                if (DM == DY_FRAG32) {
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<uint2*>(base + 0 * F_IMG + off) = hi;
                    *reinterpret_cast<uint2*>(base + 1 * F_IMG + off) = lo;
                } else {
                    *reinterpret_cast<uint2*>(base + 0 * F_IMG + off) = live ? uint2{la[0], la[1]} : uint2{0u, 0u};
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) bsum[r][c] += v[c];
            }
        } else if (tid < 32) {
            const f32x4v v = __builtin_bit_cast(f32x4v, ld_a[0]);
            uint2 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<uint2*>(base + 0 * F_IMG + img_off(tid, 0)) = hi;
            *reinterpret_cast<uint2*>(base + 1 * F_IMG + img_off(tid, 0)) = lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) bsum[0][c] += v[c];
        }
        }
        if (!(part & 2)) return;
        if constexpr (B_FRAG) {
#pragma unroll
            for (int gq = 0; gq < GPS; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool live = valid && (s * GPS + gq) * F_STEP + pi < a.n_points;
                const u32x4 lb = ld_b[8 * gq + r];
                const int pc = 8 * wave + r, t = pc >> 2, q = pc & 3;
                f32x4v v;
                if (XM == X_FRAG32) v = __builtin_bit_cast(f32x4v, lb);
                else { const u32x2 h = {lb[0], lb[1]}; v = __builtin_convertvector(__builtin_bit_cast(half4, h), f32x4v); }
                if (!live) v = f32x4v{0.f, 0.f, 0.f, 0.f};
                const int off = img_off(32 * gq + pi, 8 * t + 2 * q + gi);
                if constexpr (!B_LO) {        // f16 activations under an f16 product: the stored bits
                    *reinterpret_cast<uint2*>(base + 2 * F_IMG + off) = live ? uint2{lb[0], lb[1]} : uint2{0u, 0u};
                } else {
                    uint2 hi, lo;
                    if (F16) split4h(v, hi, lo); else split4(v, hi, lo);
                    *reinterpret_cast<uint2*>(base + 2 * F_IMG + off) = hi;
                    *reinterpret_cast<uint2*>(base + 3 * F_IMG + off) = lo;
                }
            }
        } else if constexpr (XM == X_ROWS32) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint2 hi, lo;
                if (F16) split4h(__builtin_bit_cast(f32x4v, ld_b[r]), hi, lo); else split4(__builtin_bit_cast(f32x4v, ld_b[r]), hi, lo);
                const int off = img_off(8 * wave + r, lane);
                *reinterpret_cast<uint2*>(base + 2 * F_IMG + off) = hi;
                *reinterpret_cast<uint2*>(base + 3 * F_IMG + off) = lo;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + 256 * k;
                if (idx < 320) {
                    const int row = idx / 10, ch = idx - 10 * row;
                    uint2 hi, lo;
                    if (F16) split4h(__builtin_bit_cast(f32x4v, ld_b[k]), hi, lo); else split4(__builtin_bit_cast(f32x4v, ld_b[k]), hi, lo);
                    *reinterpret_cast<uint2*>(base + 2 * F_IMG + img_off(row, ch)) = hi;
                    *reinterpret_cast<uint2*>(base + 3 * F_IMG + img_off(row, ch)) = lo;
                }
            }
        }
    };
    // transposed-read addresses of this lane inside a 32-column tile and a 16-row K-block (cdna_hip_programming.md T10): lane
    // 4 q + p of a 16-lane group supplies row q, columns 4 p .. 4 p + 3; groups 0 / 1 = columns 0-15 / 16-31, lane halves =
    // rows +0 / +8; the second read takes rows +4.  The swizzle term (row >> 1) & 7 does not depend on the K-block.
    const int q_ = (lane & 15) >> 2, p_ = lane & 3, cg = (lane >> 4) & 1, h = lane >> 5;
    const int tr1 = img_off(8 * h + q_, 4 * cg + p_), tr2 = img_off(8 * h + q_ + 4, 4 * cg + p_);

    auto mma = [&](int buf, int kb) {
        const unsigned char* img = lds + buf * F_BUF + kb * 16 * F_ROW;
        bf8 ah[NA], al[NA], bh[NB], bl[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ah[i] = tr_frag(img + 0 * F_IMG + (at0 + i) * 64, tr1, tr2);
            if (A_LO) al[i] = tr_frag(img + 1 * F_IMG + (at0 + i) * 64, tr1, tr2);
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            bh[t] = tr_frag(img + 2 * F_IMG + (bt0 + t) * 64, tr1, tr2);
            if (B_LO) bl[t] = tr_frag(img + 3 * F_IMG + (bt0 + t) * 64, tr1, tr2);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                if constexpr (F16) {
                    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bh[t]), acc[i][t], 0, 0, 0);
                    if (B_LO) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bl[t]), acc[i][t], 0, 0, 0);
                } else {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[t], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[t], acc[i][t], 0, 0, 0);
                    if (A_LO) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[t], acc[i][t], 0, 0, 0);
                }
            }
    };

    // The pieces of step s + 1 are requested as soon as the registers are free — right after step s's pieces were split into LDS —
    // and not at the start of step s + 1's own trip: they then have the second K-block of this step, the barrier and the first K-block
    // of the next step to land in (two K-blocks of 16 MFMAs instead of one; the stream is latency-bound per CU).
    // STAGGER (the default training form).  With both operands of a step requested together and split together, nothing is in flight
    // between the arrival of a step's pieces and the end of their split, and every 64-KiB burst pays the memory latency in full: the
    // launch streamed 3.9 TB/s where the same access pattern, kept continuously in flight, streams 5.7-5.9 (tools/micro/load_width.hip).
    // Here the dY pieces of step s + 2 are requested in the MIDDLE of step s (after the dY pieces of s + 1 were split) and the X pieces at
    // its END (after the X pieces of s + 1): one operand's pieces are always on their way, each with a whole step to arrive, out of the
    // same registers.
    constexpr bool STAGGER = WIDE;
    typedef std::integral_constant<int, 0> S0;
    if constexpr (STAGGER) {
        // (no branch around a request or a split: at a join the compiler's s_waitcnt placement must be right for the path that issued nothing,
        //  i.e. it would wait for the requests just made as well.  Steps past the slab's end read zeros — the descriptors are slab-relative
        //  — and their split is masked.)
        if (s0 < s1) {
            issue(S0{}, s0);
            stage(S0{}, 0, s0);
            issue(S0{}, s0 + 1);
            __syncthreads();
        }
        for (long long s = s0; s < s1; ++s) {
            const int buf = (int)(s - s0) & 1;
            const bool more = s + 1 < s1;
#pragma unroll
            for (int kb = 0; kb < GPS; ++kb) mma(buf, kb);
            __builtin_amdgcn_sched_barrier(0);        // (fences: left alone the scheduler sinks the first requests to the end of the step)
            stage(S0{}, buf ^ 1, s + 1, 1, more);     // the other buffer was last read in step s-1; every wave passed that step's barrier
            issue(S0{}, s + 2, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = GPS; kb < 2 * GPS; ++kb) mma(buf, kb);
            __builtin_amdgcn_sched_barrier(0);
            stage(S0{}, buf ^ 1, s + 1, 2, more);
            issue(S0{}, s + 2, 2);
            __syncthreads();
        }
    } else if constexpr (RING == 4) {
        // step j of the slab (j = s - s0) comes through register set j % 4 into LDS buffer j & 1.  Nothing below is conditional (see above):
        // the slab is walked in whole groups of four steps, steps past its end read zeros and multiply zeros.
        typedef std::integral_constant<int, 1> S1;
        typedef std::integral_constant<int, 2> S2;
        typedef std::integral_constant<int, 3> S3;
        if (s0 < s1) {
            issue(S0{}, s0); issue(S1{}, s0 + 1); issue(S2{}, s0 + 2); issue(S3{}, s0 + 3);
            stage(S0{}, 0, s0);
            issue(S0{}, s0 + 4);
            __syncthreads();
        }
        const long long n_quads = (n_steps + 3) >> 2;
        for (long long qd = 0; qd < n_quads; ++qd) {
            const long long s = s0 + 4 * qd;
            auto body = [&](auto nextc, int buf, long long sj) {      // step sj is in `buf`; the next step's pieces sit in set `nextc`
#pragma unroll
                for (int kb = 0; kb < GPS; ++kb) mma(buf, kb);
                __builtin_amdgcn_sched_barrier(0);
                stage(nextc, buf ^ 1, sj + 1, 3, sj + 1 < s1);
                issue(nextc, sj + 5);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kb = GPS; kb < 2 * GPS; ++kb) mma(buf, kb);
                __syncthreads();
            };
            body(S1{}, 0, s);
            body(S2{}, 1, s + 1);
            body(S3{}, 0, s + 2);
            body(S0{}, 1, s + 3);
        }
    } else {
    if (s0 < s1) {
        issue(S0{}, s0);
        stage(S0{}, 0, s0);
        if (s0 + 1 < s1) issue(S0{}, s0 + 1);
        __syncthreads();
    }
    for (long long s = s0; s < s1; ++s) {
        const int buf = (int)(s - s0) & 1;
        const bool more = s + 1 < s1;
#pragma unroll
        for (int kb = 0; kb < GPS; ++kb) mma(buf, kb);
        if (more) stage(S0{}, buf ^ 1, s + 1);      // the other buffer was last read in step s-1; every wave passed that step's barrier
        if (s + 2 < s1) issue(S0{}, s + 2);
#pragma unroll
        for (int kb = GPS; kb < 2 * GPS; ++kb) mma(buf, kb);
        __syncthreads();
    }
    }

    // partial slab: D row = n (A operand's row), column = k (B operand's column)
    float* out = a.dw_part + (size_t)g * N_OUT * LD_OUT;
    const int c = lane & 31;
    // form 3: back from the slab's common scale 2^(bmin - 113) (an all-zero slab has nothing to scale)
    const float unscale = (F16 && bmin != 255) ? __builtin_bit_cast(float, (unsigned)(240 - bmin) << 23) : 1.0f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            if (F16) asm volatile("" : "+a"(acc[i][t]));       // the accumulators stay in the AGPRs (the scaling below would pull all 256 into VGPRs)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * (at0 + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)n * LD_OUT + 32 * (bt0 + t) + c] = F16 ? acc[i][t][r] * unscale : acc[i][t][r];
            }
        }
    if (a.db_part) {
        if constexpr (A_FRAG) {                // piece r of this wave = columns 32 t + 8 q + 4 g + c; sum over the 32 points of a lane half
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    float v = bsum[r][cc];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    const int pc = 8 * wave + r, t = pc >> 2, q = pc & 3;
                    if (pi == 0) a.db_part[(size_t)g * N_OUT + 32 * t + 8 * q + 4 * gi + cc] = F16 ? v * unscale : v;
                }
        } else if (wave == 0) {                // dz rows: threads 0..31 hold one row each; the head has 4 (3 used) outputs
            float keep[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float v = lane < 32 ? bsum[0][cc] : 0.f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                keep[cc] = v;
            }
            if (lane < 32) a.db_part[(size_t)g * N_OUT + lane] = lane < 4 ? keep[lane & 3] : 0.f;
        }
    }
}

template <int SHAPE, int XM, int DM>
int launch(const DwfBatch& a, int n, int groups, hipStream_t s) {
    hipLaunchKernelGGL((vfn_dwf_kernel<SHAPE, XM, DM>), dim3(groups, n), dim3(256), 0, s, a);
    return vfn_check_launch("vfn_weight_grad_frag");
}

}  // namespace

extern "C" int vfn_weight_grad_frag(int32_t shape, const void* dy, int32_t dy_form, const void* x, int32_t x_form, int64_t n_points,
                                    int32_t groups, float* dw_part, float* db_part, void* stream) {
    return vfn_internal_weight_grad_frag_batch(shape, dy_form, x_form, 1, &dy, &x, &dw_part, &db_part, n_points, groups, stream);
}

int vfn_internal_weight_grad_frag_batch(int32_t shape, int32_t dy_form, int32_t x_form, int32_t n, const void* const* dy, const void* const* x,
                                        float* const* dw_part, float* const* db_part, int64_t n_points, int32_t groups, void* stream) {
    return vfn_internal_weight_grad_frag_batch_dev(shape, dy_form, x_form, n, dy, x, dw_part, db_part, n_points, nullptr, groups, stream);
}

int vfn_internal_weight_grad_frag_batch_dev(int32_t shape, int32_t dy_form, int32_t x_form, int32_t n, const void* const* dy, const void* const* x,
                                            float* const* dw_part, float* const* db_part, int64_t n_points, const int32_t* n_dev, int32_t groups,
                                            void* stream) {
    VFN_REQUIRE(n >= 1 && n <= DWF_BATCH && dy && x && dw_part && db_part, "vfn_weight_grad_frag: bad batch (n=%d)", n);
    VFN_REQUIRE(groups >= 1 && groups <= 4096, "vfn_weight_grad_frag: groups=%d", groups);
    VFN_REQUIRE(n_points >= 0 && n_points < (1ll << 20) * groups, "vfn_weight_grad_frag: slab larger than 1 GiB");
    DwfBatch a = {};
    a.n_dev = n_dev;
    for (int i = 0; i < n; ++i) {
        VFN_REQUIRE(dy[i] && x[i] && dw_part[i], "vfn_weight_grad_frag: NULL argument");
        a.v[i].dy = dy[i]; a.v[i].x = x[i]; a.v[i].dw_part = dw_part[i]; a.v[i].db_part = db_part[i]; a.v[i].n_points = n_points;
    }
    hipStream_t s = (hipStream_t)stream;
    if (shape == 0) {
        VFN_REQUIRE(dy_form == DY_FRAG32 || dy_form == DY_FRAGBF16 || dy_form == DY_FRAGF16S, "vfn_weight_grad_frag: shape 0 takes a fragment-ordered dY");
        if (dy_form == DY_FRAGF16S) {
            if (x_form == X_FRAG32) return launch<0, X_FRAG32, DY_FRAGF16S>(a, n, groups, s);
            if (x_form == X_FRAG16) return launch<0, X_FRAG16, DY_FRAGF16S>(a, n, groups, s);
            if (x_form == X_ROWS32) return launch<0, X_ROWS32, DY_FRAGF16S>(a, n, groups, s);
        } else if (dy_form == DY_FRAG32) {
            if (x_form == X_FRAG32) return launch<0, X_FRAG32, DY_FRAG32>(a, n, groups, s);
            if (x_form == X_FRAG16) return launch<0, X_FRAG16, DY_FRAG32>(a, n, groups, s);
            if (x_form == X_ROWS32) return launch<0, X_ROWS32, DY_FRAG32>(a, n, groups, s);
        } else {
            if (x_form == X_FRAG32) return launch<0, X_FRAG32, DY_FRAGBF16>(a, n, groups, s);
            if (x_form == X_FRAG16) return launch<0, X_FRAG16, DY_FRAGBF16>(a, n, groups, s);
            if (x_form == X_ROWS32) return launch<0, X_ROWS32, DY_FRAGBF16>(a, n, groups, s);
        }
    } else if (shape == 1) {
        VFN_REQUIRE(x_form == X_AUX40, "vfn_weight_grad_frag: shape 1 takes the [M][40] encoding tile as X");
        if (dy_form == DY_FRAG32) return launch<1, X_AUX40, DY_FRAG32>(a, n, groups, s);
        if (dy_form == DY_FRAGBF16) return launch<1, X_AUX40, DY_FRAGBF16>(a, n, groups, s);
        if (dy_form == DY_FRAGF16S) return launch<1, X_AUX40, DY_FRAGF16S>(a, n, groups, s);
    } else if (shape == 2) {
        VFN_REQUIRE(dy_form == DY_DZ4, "vfn_weight_grad_frag: shape 2 takes the [M][4] head gradient as dY");
        if (x_form == X_FRAG32) return launch<2, X_FRAG32, DY_DZ4>(a, n, groups, s);
        if (x_form == X_FRAG16) return launch<2, X_FRAG16, DY_DZ4>(a, n, groups, s);
        if (x_form == X_ROWS32) return launch<2, X_ROWS32, DY_DZ4>(a, n, groups, s);
    }
    vfn_set_error("vfn_weight_grad_frag: unsupported combination shape=%d dy_form=%d x_form=%d", shape, dy_form, x_form);
    return VFN_ERR_INVALID;
}
