// vfn_bstat.hip — the two MLPs with nn.BatchNorm1d in TRAINING mode (batch statistics), layer at a time.
//
// Reference: models/vector_field/vector_field_network.py:146-208 and rendering_network.py:62-108 after
// VectorFieldNerf.train() (models/nerf/vector_field_nerf.py:139-150) — the regime the trainer enters when the
// directional-derivative loss weight is non-zero (train/vector_field_nerf_train.py:140-141).  Batch statistics couple
// every row of a layer's output, so — unlike the eval-mode path, where BatchNorm folds into the weights and all layers
// fuse into one launch — each layer needs the column means / variances of the whole batch before the next one can
// start, and its backward needs two more column sums before dZ exists.  The path is therefore a sequence of
//   * vfn_linear_rows          C = act(A W^T + b) or C = A W, exact fp32 on v_mfma_f32_32x32x2f32, optional per-block
//                              column sums of z and z^2 (the batch statistics, no atomics),
//   * vfn_colsum_finish        per-block partial sums -> double column sums,
//   * vfn_bstat_finalize       sums -> mean / biased variance -> scale, shift, mean, rstd; running statistics update,
//   * vfn_bstat_relu_rows      h = post * relu(z * scale + shift),
//   * vfn_bstat_relu_bwd_sums  per-block partials of sum g', sum g' x_hat  (g' = post * g * [z * scale + shift > 0]),
//   * vfn_bstat_relu_bwd_rows  dz = gamma rstd (g' - mean g' - x_hat mean(g' x_hat)),
//   * vfn_act_bwd_rows         tanh / sigmoid backward (or a one-hot seed for the autograd.grad rows),
//   * vfn_embed_rows(_bwd)     positional encoding and its derivative wrt the point,
// and the weight gradients reuse vfn_weight_grad_partials + vfn_unfold_weight_grads.  Activations live in HBM as
// row-major fp32 matrices whose leading dimensions are multiples of 4 floats and whose pad columns hold zeros.
//
// Roofline: the fp32 matrix pipe (32x32x2: 64 cycles per MFMA per SIMD, 157 TFLOP/s nominal) bounds the GEMM; all other
// kernels are one pass over [M, n] fp32 and HBM-bound.
#include <stdlib.h>
#include "vfn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM_ROWS = 128;    // rows per workgroup (4 waves x 32)
constexpr int GM_KC = 32;       // contraction chunk staged in LDS
constexpr int GM_LD = 36;       // LDS row stride (floats): 16-byte aligned, conflict-free b128 reads
[[maybe_unused]] constexpr int ACT_NONE = 0;
constexpr int ACT_TANH = 1, ACT_SIGMOID = 2;

struct GemmArgs {
    const float* a; const float* w; const float* bias; float* c; float* stats_part;
    long long m;
    int lda, ldw, ldc;
    int n_out;      // output columns (all launches together)
    int k_in;       // contraction length actually present in W
    int k_pad;      // contraction length to run over A (multiple of 8, pad columns of A are zero)
    int n0;         // first output column of this launch
    int act;
    int stats_ld;   // = n_out
    // vfn_linear_rows_dx_sums (the split kernels with SUMS): stats_part = per-block partials of the NEXT BatchNorm backward's two column sums
    // over the first stats_ld columns of C, from that layer's pre-BatchNorm outputs zp and coefficients coef_p [4][stats_ld]
    const float* zp; const float* coef_p;
    int ldzp;
    float post_p;
    unsigned long long* probe;      // VFN_GEMM_PROBE: per-workgroup cycle counts of the phases of the chunk loop (debug)
    uint32_t* status;               // split f16 form: bit 0 is set when an element of A leaves the range its scaled halves cover (|a| >= 1023)
    // W split ONCE per call into its two 16-bit planes by vfn_gemm_split_w_kernel (the two-plane arithmetics; caller's scratch): per block of
    // 256 output columns and chunk of 32 k, [plane hi | mid][256 n][32 k] u16 — what a workgroup stages, in the order it stages it
    const unsigned short* wp;
    int wp_chunks;
    // FOLD (vfn_linear_rows_fold, round 6): A is the PREVIOUS layer's pre-BatchNorm output z; the operand the product multiplies is
    // fold_post * max(z * scale + shift, 0) for the first fold_n columns (fold_coef = that layer's [4][fold_n] coefficients) and
    // fold_post * z for the others (the skip layer's re-injected encoding) — formed when the A fragment is read, so the activated
    // matrix is never written to or read from HBM.
    const float* fold_coef;
    int fold_n;
    float fold_post;
};

// TRANS = false: B(k, n) = W[n][k]  (nn.Linear weight, C = A W^T);  TRANS = true: B(k, n) = W[k][n]  (C = A W).
template <int NT, bool TRANS>
__global__ __launch_bounds__(256, 2) void vfn_linear_rows_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float s_b[NT * 32 * GM_LD];
    __shared__ float s_red[2][4][NT * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, g = lane >> 5;
    constexpr int NCOL = NT * 32;
    constexpr int PER = NCOL * GM_KC / 256;      // W elements each thread stages per chunk
    constexpr unsigned OOB = 0x7fffffffu;        // an offset past every buffer: the load returns 0, no branch

    // bounds-checked buffer descriptors: A restricted to this workgroup's rows (rows past m read as zero), W whole
    const long long blk_row0 = (long long)blockIdx.x * GM_ROWS;
    const long long blk_rows = min((long long)GM_ROWS, a.m - blk_row0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.a + (size_t)blk_row0 * a.lda), 0, (int)(blk_rows * a.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)((long long)(TRANS ? a.k_in : a.n_out) * a.ldw * 4), 0x00020000);
    const unsigned a_row_off = (unsigned)(32 * wave + c) * (unsigned)a.lda * 4u;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // element e of the [NCOL][32] chunk -> (column n, contraction index kk).  Plain: consecutive lanes walk k (coalesced
    // rows of W[n][k]).  Transposed: a wave covers 16 columns x 4 k (coalesced rows of W[k][n], and the LDS stores of its 64
    // lanes fall on distinct banks: 36 n + kk = 4 (e>>2 & 15) + (e & 3) mod 32).
    auto stage_index = [&](int e, int& n, int& kk) {
        if (!TRANS) { kk = e & 31; n = e >> 5; }
        else { const int rest = e >> 6; kk = (e & 3) + 4 * (rest & 7); n = ((e >> 2) & 15) + 16 * (rest >> 3); }
    };
    // per-thread W offsets of chunk 0 (bytes); a chunk further on adds kc * (TRANS ? ldw : 1) * 4
    unsigned w_off[PER];
    int w_kk[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        int n, kk;
        stage_index(tid + 256 * r, n, kk);
        const int col = a.n0 + n;
        w_kk[r] = kk;
        w_off[r] = col < a.n_out ? (TRANS ? ((unsigned)kk * (unsigned)a.ldw + (unsigned)col) * 4u : ((unsigned)col * (unsigned)a.ldw + (unsigned)kk) * 4u) : OOB;
    }
    const unsigned w_step = (TRANS ? (unsigned)a.ldw : 1u) * 4u;
    float wreg[PER];
    auto fetch_w = [&](int kc) {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const unsigned off = (w_off[r] == OOB || kc + w_kk[r] >= a.k_in) ? OOB : w_off[r] + (unsigned)kc * w_step;
            wreg[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, off, 0, 0));
        }
    };
    auto stage_w = [&]() {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            int n, kk;
            stage_index(tid + 256 * r, n, kk);
            s_b[n * GM_LD + kk] = wreg[r];
        }
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 av[4];
    auto fetch_a = [&](int kc) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int k = kc + 8 * kb + 4 * g;
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs_a, k < a.k_pad ? a_row_off + (unsigned)k * 4u : OOB, 0, 0);
            av[kb] = __builtin_bit_cast(f32x4, raw);
        }
    };

    fetch_w(0);
    for (int kc = 0; kc < a.k_pad; kc += GM_KC) {
        __syncthreads();               // every wave is done with the previous chunk
        stage_w();
        __syncthreads();
        // this chunk's A fragments first, then the next chunk's W: loads return in order, so the MFMAs below wait only for
        // the four A loads while the W prefetch stays in flight underneath them
        fetch_a(kc);
        if (kc + GM_KC < a.k_pad) fetch_w(kc + GM_KC);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kc + 8 * kb >= a.k_pad) break;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_b + (32 * j + c) * GM_LD + 8 * kb + 4 * g);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb][t], bv[t], acc[j], 0, 0, 0);
            }
        }
    }

    // epilogue: D row = (r&3) + 8 (r>>2) + 4 g, col = c
    const long long row0 = (long long)blockIdx.x * GM_ROWS + 32 * wave;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = a.n0 + 32 * j + c;
        const bool col_ok = col < a.n_out;
        const float b = (a.bias && col_ok) ? a.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const float z = acc[j][r] + b;
            if (row < a.m && col_ok) {
                s1 += z;
                s2 += z * z;
                float y = z;
                if (a.act == ACT_TANH) y = tanhf(z);
                else if (a.act == ACT_SIGMOID) y = 1.0f / (1.0f + expf(-z));
                a.c[(size_t)row * a.ldc + col] = y;
            }
        }
        if (a.stats_part) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (g == 0) { s_red[0][wave][32 * j + c] = s1; s_red[1][wave][32 * j + c] = s2; }
        }
    }
    if (a.stats_part) {
        __syncthreads();
        for (int i = tid; i < 2 * NCOL; i += 256) {
            const int which = i / NCOL, n = i - which * NCOL;
            const int col = a.n0 + n;
            if (col < a.n_out) {
                const float s = s_red[which][0][n] + s_red[which][1][n] + s_red[which][2][n] + s_red[which][3][n];
                a.stats_part[((size_t)blockIdx.x * 2 + which) * a.stats_ld + col] = s;
            }
        }
    }
}

// ---- the same GEMM on the 16-bit matrix cores with SPLIT operands (round 4) --------------------------------------------------
// Every fp32 operand is carried as two 16-bit halves and a product is a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_*
// with fp32 accumulation: 3 MFMAs of 32 cycles per K = 16 instead of 8 fp32 MFMAs of 64 cycles (5.3x fewer matrix cycles).
//   F16 (forward GEMMs: batch-normalised activations, weights of O(0.1)): hi = f16(v), lo = f16(v - hi): 22 significant bits; A rides
//        at 2^6 x its value so that its low halves stay out of the f16 denormals (undone in the epilogue), the weights' low halves
//        may be denormal (the MFMA honours them: an absolute resolution of 6e-8);
//   BF16 (backward GEMMs dX = dZ W: gradients of any magnitude): hi = the top 16 bits (truncation), lo = bf16(v - hi): 16 significant
//        bits with fp32's exponent range, no scaling needed — the dX chain of the fused kernels (csrc/vfn_bwd16.hip) is the same split.
// Same tiling as the exact kernel (128 rows x NT x 32 columns per workgroup, W staged in LDS in chunks of 32 k, A fragments from
// global memory by bounds-checked 16-byte loads), same epilogue (bias, activation, per-workgroup column sums of z and z^2).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
constexpr int GM_LDH = 40;      // LDS row stride of a 16-bit W plane (elements): 80 bytes, 16-byte aligned

// ARITH: 0 = split f16 (3 products, 22 bits), 1 = split bf16 (3 products, 16 bits), 2 = bf16 in THREE parts (hi | mid | lo = 24 bits,
// six products hh hm mh hl lh mm: fp32-equivalent at fp32's exponent range, 192 instead of 512 matrix cycles per K = 16)
template <int ARITH>
__device__ __forceinline__ void split3(float v, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    if constexpr (ARITH == 0) {
        const _Float16 h = (_Float16)v;
        hi = __builtin_bit_cast(unsigned short, h);
        mid = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
        lo = 0;
    } else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        hi = (unsigned short)(u >> 16);
        const float r = v - __builtin_bit_cast(float, u & 0xffff0000u);
        if constexpr (ARITH == 1) {
            mid = __builtin_bit_cast(unsigned short, (__bf16)r);
            lo = 0;
        } else {
            const unsigned ur = __builtin_bit_cast(unsigned, r);
            mid = (unsigned short)(ur >> 16);
            lo = __builtin_bit_cast(unsigned short, (__bf16)(r - __builtin_bit_cast(float, ur & 0xffff0000u)));
        }
    }
}

// SUMS (vfn_linear_rows_dx_sums): C is the gradient wrt the previous layer's activated output; the per-block partials of sum g' and
// sum g' x_hat of that layer's BatchNorm backward (g' = post g [z scale + shift > 0], x_hat = (z - mean) rstd) are taken from C while it is
// in registers — what vfn_bstat_relu_bwd_sums computes in a pass of its own over g and z (2.3 TB/s, 16 % of a training-mode step).
// PK: W arrives as pre-split planes (GemmArgs::wp): every workgroup used to fetch all of W as fp32 dwords and split it again — 32 loads and
// ~300 VALU instructions per thread and chunk, 40 % of a workgroup's cycles (round 5 phase probe) — for values that are the same for all of them.
template <int NT, bool TRANS, int ARITH, bool SUMS = false, bool PK = false, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void vfn_linear_rows16_kernel(const GemmArgs a) {
    constexpr bool THREE = ARITH == 2;
    static_assert(!FOLD || (!TRANS && !SUMS), "the fold is a forward product's");
    static_assert(!(PK && THREE), "pre-split planes exist for the two-plane arithmetics");
    constexpr int NCOL = NT * 32;
    constexpr int A_LD = 36;                     // floats per staged A row (32 k + 4): conflict-free 16-byte fragment reads
    __shared__ __attribute__((aligned(16))) unsigned short s_hi[NCOL * GM_LDH];
    __shared__ __attribute__((aligned(16))) unsigned short s_mid[NCOL * GM_LDH];
    __shared__ __attribute__((aligned(16))) unsigned short s_lo[THREE ? NCOL * GM_LDH : 8];
    __shared__ __attribute__((aligned(16))) float s_a[GM_ROWS * A_LD];
    __shared__ __attribute__((aligned(16))) float s_fold[FOLD ? 3 * GM_KC : 4];      // the chunk's scale | shift | lower bound per k
    // (the column-sum partials of the epilogue reuse the A tile: 8 KiB of its 18; with three weight planes two workgroups still fit a CU)
    static_assert(2 * 4 * NCOL <= GM_ROWS * A_LD, "s_red must fit into the A tile");
    float (*s_red)[4][NCOL] = reinterpret_cast<float (*)[4][NCOL]>(s_a);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, g = lane >> 5;
    constexpr int PER = NCOL * 16 / 256;         // PAIRS of consecutive k each thread stages per chunk of 32 k
    constexpr unsigned OOB = 0x7fffffffu;
    constexpr float A_SCALE = ARITH == 0 ? 64.0f : 1.0f;

    const unsigned rb = blockIdx.x;          // this workgroup's block of 128 rows
    const int n0 = a.n0;
    const long long blk_row0 = (long long)rb * GM_ROWS;
    const long long blk_rows = min((long long)GM_ROWS, a.m - blk_row0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.a + (size_t)blk_row0 * a.lda), 0, (int)(blk_rows * a.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, (int)((long long)(TRANS ? a.k_in : a.n_out) * a.ldw * 4), 0x00020000);

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    auto stage_index = [&](int e, int& n, int& kk) {
        if (!TRANS) { kk = 2 * (e & 15); n = e >> 4; }
        else { n = e % NCOL; kk = 2 * (e / NCOL); }
    };
    float wreg[PER][2];
    // W: the per-thread part of an element's address is the same for all PER x 2 loads of a chunk (TRANS: the thread's column and its
    // first k; else its first row and its k pair) and goes into the buffer instruction's vector offset once; what changes from load to
    // load is uniform and rides in the scalar offset.  (With the whole address recomputed and range-checked per load in vector
    // instructions, ISSUING the 32 loads of a chunk took 2 500 cycles per workgroup — as long as its matrix instructions.)
    static_assert(256 % NCOL == 0, "a thread keeps its column over the passes");
    const int w_kk = TRANS ? 2 * (tid / NCOL) : 2 * (tid & 15);      // this thread's k within a chunk (plus the uniform part, TRANS)
    const int w_n = TRANS ? tid % NCOL : tid >> 4;                  // its column (plus 16 r without TRANS)
    const unsigned w_voff = TRANS ? ((unsigned)w_kk * (unsigned)a.ldw + (unsigned)(n0 + w_n)) * 4u
                                  : ((unsigned)(n0 + w_n) * (unsigned)a.ldw + (unsigned)w_kk) * 4u;
    auto fetch_w = [&](int kc) {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bool ok;
                unsigned soff;
                if constexpr (TRANS) {
                    const int ku = kc + 2 * r * (256 / NCOL) + q;      // uniform part of k
                    ok = n0 + w_n < a.n_out && w_kk + ku < a.k_in;
                    soff = (unsigned)ku * (unsigned)a.ldw * 4u;
                } else {
                    ok = n0 + w_n + 16 * r < a.n_out && w_kk + kc + q < a.k_in;
                    soff = ((unsigned)(16 * r) * (unsigned)a.ldw + (unsigned)(kc + q)) * 4u;
                }
                // (the range check of a raw buffer access covers the vector offset only: invalid elements get the out-of-range one)
                wreg[r][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, ok ? w_voff : OOB, soff, 0));
            }
        }
    };
    auto stage_w = [&]() {
        if constexpr (TRANS && NCOL == 256) {
            // the thread holds the chunk's 32 k of ONE column: 64 bytes per plane, four 16-byte stores instead of sixteen 4-byte ones
            typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4w ph, pm, pl;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    unsigned short h0, m0, l0, h1, m1, l1;
                    split3<ARITH>(wreg[4 * i + t][0], h0, m0, l0);
                    split3<ARITH>(wreg[4 * i + t][1], h1, m1, l1);
                    ph[t] = (unsigned)h0 | ((unsigned)h1 << 16);
                    pm[t] = (unsigned)m0 | ((unsigned)m1 << 16);
                    pl[t] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
                *reinterpret_cast<u32x4w*>(s_hi + w_n * GM_LDH + 8 * i) = ph;
                *reinterpret_cast<u32x4w*>(s_mid + w_n * GM_LDH + 8 * i) = pm;
                if constexpr (THREE) *reinterpret_cast<u32x4w*>(s_lo + w_n * GM_LDH + 8 * i) = pl;
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            int n, kk;
            stage_index(tid + 256 * r, n, kk);
            unsigned short h0, m0, l0, h1, m1, l1;
            split3<ARITH>(wreg[r][0], h0, m0, l0);
            split3<ARITH>(wreg[r][1], h1, m1, l1);
            *reinterpret_cast<unsigned*>(s_hi + n * GM_LDH + kk) = (unsigned)h0 | ((unsigned)h1 << 16);
            *reinterpret_cast<unsigned*>(s_mid + n * GM_LDH + kk) = (unsigned)m0 | ((unsigned)m1 << 16);
            if constexpr (THREE) *reinterpret_cast<unsigned*>(s_lo + n * GM_LDH + kk) = (unsigned)l0 | ((unsigned)l1 << 16);
        }
    };
    // PK: the chunk's two planes as 16-byte pieces (8 k of one column), NCOL * 4 per plane, straight from the planes to LDS rows
    typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
    constexpr int WPIECES = PK ? (NCOL * 4 + 255) / 256 : 1;
    [[maybe_unused]] u32x4p preg[2][WPIECES];
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(PK ? a.wp + (size_t)(n0 / 256) * a.wp_chunks * 2 * 8192 : nullptr), 0,
        PK ? a.wp_chunks * 2 * 8192 * 2 : 0, 0x00020000);
    [[maybe_unused]] auto fetch_wp = [&](int kc) {
        const unsigned soff = (unsigned)(kc / GM_KC) * 2u * 8192u * 2u;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < WPIECES; ++i) {
                const int q = tid + 256 * i;                    // piece: column q >> 2, k 8 (q & 3) ..
                preg[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wp, q < NCOL * 4 ? (unsigned)q * 16u : OOB, soff + (unsigned)pl * 8192u * 2u, 0);
            }
    };
    [[maybe_unused]] auto stage_wp = [&]() {
#pragma unroll
        for (int i = 0; i < WPIECES; ++i) {
            const int q = tid + 256 * i;
            if (q < NCOL * 4) {
                *reinterpret_cast<u32x4p*>(s_hi + (q >> 2) * GM_LDH + 8 * (q & 3)) = preg[0][i];
                *reinterpret_cast<u32x4p*>(s_mid + (q >> 2) * GM_LDH + 8 * (q & 3)) = preg[1][i];
            }
        }
    };
    // A: the workgroup's 128 rows x 32 k of a chunk through LDS.  Eight consecutive lanes fetch one row's 128 bytes (whole cache
    // lines; 32 rows per pass, four passes), instead of every lane fetching 32-byte pieces of its own row 1 KiB from its
    // neighbour's — that pattern, not the matrix pipe, bounded the first version of this kernel at 1.6 TB/s.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 areg[4];
    const int a_r = tid >> 3, a_q = tid & 7;     // row within a pass, 16-byte piece of the row's chunk
    [[maybe_unused]] float freg = 0.f;           // FOLD: this thread's entry of the chunk's coefficient table (threads 0 .. 95)
    auto fetch_a = [&](int kc) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 32 * ps + a_r, k = kc + 4 * a_q;
            areg[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, k < a.k_pad ? ((unsigned)row * (unsigned)a.lda + (unsigned)k) * 4u : OOB, 0, 2);
        }
        if constexpr (FOLD) {
            if (tid < 3 * GM_KC) {
                const int which = tid / GM_KC, k = kc + tid % GM_KC;
                const bool bn = k < a.fold_n;
                // scale | shift of the BatchNorm'ed columns; identity without a ReLU (lower bound -inf) for the columns behind them
                freg = which == 0 ? (bn ? a.fold_coef[k] : 1.0f) : (which == 1 ? (bn ? a.fold_coef[a.fold_n + k] : 0.0f) : (bn ? 0.0f : -__builtin_inff()));
            }
        }
    };
    [[maybe_unused]] float a_max = 0.f;       // split f16 form: largest |A| this thread staged (the range report at the end)
    [[maybe_unused]] const bool fold_row_ok = 32 * wave + c < blk_rows;       // FOLD: this lane's fragment row exists
    auto stage_a = [&]() {
        if constexpr (FOLD) {
            if (tid < 3 * GM_KC) s_fold[tid] = freg;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            *reinterpret_cast<u32x4*>(s_a + (32 * ps + a_r) * A_LD + 4 * a_q) = areg[ps];
            if constexpr (ARITH == 0 && !FOLD) {      // (FOLD: the range report looks at the ACTIVATED operand, below)
                const f32x4 v = __builtin_bit_cast(f32x4, areg[ps]);
                a_max = fmaxf(fmaxf(a_max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
        }
    };

    if constexpr (PK) fetch_wp(0); else fetch_w(0);
    fetch_a(0);
    // The phase probe is a BUILD option (tools/build_unit_variant.sh probe vfn_bstat -DVFN_GEMM_PROBE_BUILD, then VFN_GEMM_PROBE=1): as a
    // run-time switch its seven uniform branches cut every iteration of the chunk loop into seven scheduling regions (round 5).
#ifdef VFN_GEMM_PROBE_BUILD
    constexpr bool PROBE = true;
#else
    constexpr bool PROBE = false;
#endif
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, t_prev = (PROBE && a.probe) ? __builtin_amdgcn_s_memtime() : 0;
    auto mark = [&](int which) {
        if (PROBE && a.probe) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            ph[which] += t - t_prev;
            t_prev = t;
        }
    };
    for (int kc = 0; kc < a.k_pad; kc += GM_KC) {
        __syncthreads();               // every wave is done with the previous chunk
        mark(0);
        if (PROBE && a.probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mark(1); }
        if constexpr (PK) stage_wp(); else stage_w();
        stage_a();
        mark(2);
        __syncthreads();
        mark(3);
        if (kc + GM_KC < a.k_pad) {                                                   // in flight underneath this chunk's matrix work
            if constexpr (PK) fetch_wp(kc + GM_KC); else fetch_w(kc + GM_KC);
            fetch_a(kc + GM_KC);
        }
        mark(4);
        // (Hand-pipelining this section — B fragments one tile ahead, the second K-block's A fragment split behind the first's last tile —
        // was tried and bought 2 %: the launch is bound by its HBM streams, see the phase table in profiles/r04/linear_rows_microbench.txt,
        // and the eight-tile forms have no registers for the extra fragments.)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (no early exit for a chunk that holds 16 k or fewer: A beyond k_pad is staged as zeros — out-of-range loads — and the branch
            //  kept the second K-block's operand split from being scheduled under the first one's matrix instructions)
            // (the bf16 forms keep three operand planes per K-block: with both blocks' splits in flight they spill — fenced apart)
            if constexpr (ARITH != 0) __builtin_amdgcn_sched_barrier(0);
            const float* ap = s_a + (32 * wave + c) * A_LD + 16 * ks + 8 * g;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
            u16x8 ah, am, al;
            [[maybe_unused]] f32x4 fs0, fs1, fh0, fh1, fl0, fl1;
            if constexpr (FOLD) {
                const float* fp = s_fold + 16 * ks + 8 * g;
                fs0 = *reinterpret_cast<const f32x4*>(fp); fs1 = *reinterpret_cast<const f32x4*>(fp + 4);
                fh0 = *reinterpret_cast<const f32x4*>(fp + GM_KC); fh1 = *reinterpret_cast<const f32x4*>(fp + GM_KC + 4);
                fl0 = *reinterpret_cast<const f32x4*>(fp + 2 * GM_KC); fl1 = *reinterpret_cast<const f32x4*>(fp + 2 * GM_KC + 4);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                unsigned short h, m, l;
                float av = (t < 4 ? a0[t & 3] : a1[t & 3]);
                if constexpr (FOLD) {
                    // the expression of vfn_bstat_relu_rows (post * max(fma(z, scale, shift), 0)): the operand is that pass's output, bit for bit
                    av = __fmul_rn(a.fold_post, fmaxf(fmaf(av, t < 4 ? fs0[t & 3] : fs1[t & 3], t < 4 ? fh0[t & 3] : fh1[t & 3]), t < 4 ? fl0[t & 3] : fl1[t & 3]));
                    // (a row past the end of the matrix reads z = 0, i.e. max(shift, 0) — any size: it must neither reach the range report nor
                    //  the product as anything but the zero an unfolded operand reads there)
                    av = fold_row_ok ? av : 0.f;
                    if constexpr (ARITH == 0) a_max = fmaxf(a_max, fabsf(av));
                }
                av *= A_SCALE;
                // split f16 form: an element beyond the range (reported below) is SATURATED to the largest finite half, as the fused f16x3
                // kernels saturate — the flagged call then returns clamped values, never inf - inf = NaN products (which the BatchNorm running
                // statistics of a training-mode forward would keep)
                if constexpr (ARITH == 0) av = __builtin_amdgcn_fmed3f(av, -65504.0f, 65504.0f);
                split3<ARITH>(av, h, m, l);
                ah[t] = h; am[t] = m; al[t] = l;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int o = (32 * j + c) * GM_LDH + 16 * ks + 8 * g;
                const u16x8 bh = *reinterpret_cast<const u16x8*>(s_hi + o);
                const u16x8 bm = *reinterpret_cast<const u16x8*>(s_mid + o);
                if constexpr (ARITH == 0) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bh), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bm), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, am), __builtin_bit_cast(h8, bh), acc[j], 0, 0, 0);
                } else {
                    // smallest terms first
                    if constexpr (THREE) {
                        const u16x8 bl = *reinterpret_cast<const u16x8*>(s_lo + o);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, am), __builtin_bit_cast(b8, bm), acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bl), acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, al), __builtin_bit_cast(b8, bh), acc[j], 0, 0, 0);
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bm), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, am), __builtin_bit_cast(b8, bh), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bh), acc[j], 0, 0, 0);
                }
            }
        }
        mark(5);
    }

    // epilogue: D row = (r&3) + 8 (r>>2) + 4 g, col = c  (as the exact kernel; the f16 form undoes A's scale first)
    if (a.stats_part) __syncthreads();          // every wave has read its last A fragments: the tile becomes s_red
    const long long row0 = (long long)rb * GM_ROWS + 32 * wave;
    [[maybe_unused]] const int live = (int)min((long long)32, a.m - row0);            // rows of this wave inside the matrix
    [[maybe_unused]] float znext[16];
    [[maybe_unused]] auto fetch_z = [&](int j) {     // the previous layer's pre-BatchNorm outputs under tile j of C (the accumulators' rows)
        const int col = n0 + 32 * j + c;
        const float* zp0 = a.zp + (size_t)row0 * a.ldzp + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * g;
            znext[r] = (lr < live && col < a.stats_ld) ? zp0[lr * a.ldzp] : 0.f;
        }
    };
    // FAST epilogue (round 5): a wave whose 32 rows and whose NCOL columns all exist, no activation — every 256-wide layer product of a
    // training-mode step.  C leaves through range-checked buffer stores whose per-lane offsets are computed ONCE (16 registers: the
    // accumulator rows; the tile's column block rides in the instruction's immediate offset), the previous layer's z arrives the same way.
    // The general path below guards every one of a lane's 128 values with its own row / column test and forms a 64-bit address for it:
    // ~2 000 instructions and 250 divergent branches per wave, half the life of a workgroup (profiles/r04/linear_rows_microbench.txt
    // took that for store bandwidth).
    const bool whole = live == 32 && n0 + NCOL <= a.n_out && a.act == ACT_NONE;
    if (whole && (!SUMS || !a.stats_part || n0 + NCOL <= a.stats_ld)) {
        const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(a.c + (size_t)row0 * a.ldc, 0, (int)(32u * (unsigned)a.ldc * 4u), 0x00020000);
        // (per lane: its column and the 4 g rows; the accumulator's row — uniform — rides in the scalar offset, the tile's column block in the
        //  immediate: no register per row.  A buffer's range check does not see the scalar offset: it is not needed, the wave's 32 rows exist.)
        const unsigned vbase = ((unsigned)(4 * g) * (unsigned)a.ldc + (unsigned)(n0 + c)) * 4u;
        const unsigned ldc4 = (unsigned)a.ldc * 4u;
        auto srow = [](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)); };
        if constexpr (SUMS) {
            if (a.stats_part) {
                const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.zp) + (size_t)row0 * a.ldzp, 0,
                                                                                      (int)(32u * (unsigned)a.ldzp * 4u), 0x00020000);
                const unsigned zbase = ((unsigned)(4 * g) * (unsigned)a.ldzp + (unsigned)(n0 + c)) * 4u;
                const unsigned ldz4 = (unsigned)a.ldzp * 4u;
                float zq[2][16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zq[0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, zbase, srow(r) * ldz4, 0));
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = n0 + 32 * j + c;
                    const float sc = a.coef_p[col], sh = a.coef_p[a.stats_ld + col], mean = a.coef_p[2 * a.stats_ld + col], rstd = a.coef_p[3 * a.stats_ld + col];
                    if (j + 1 < NT) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            zq[(j + 1) & 1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, zbase + 128u * (unsigned)(j + 1), srow(r) * ldz4, 0));
                    }
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[j][r] * (1.0f / A_SCALE);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c, vbase + 128u * (unsigned)j, srow(r) * ldc4, 2);
                        const float z = zq[j & 1][r];
                        const float g1 = fmaf(z, sc, sh) > 0.f ? a.post_p * v : 0.f;
                        s1 += g1;
                        s2 += g1 * ((z - mean) * rstd);
                    }
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (g == 0) { s_red[0][wave][32 * j + c] = s1; s_red[1][wave][32 * j + c] = s2; }
                }
            }
        }
        if (!SUMS || !a.stats_part) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + 32 * j + c;
                const float b = a.bias ? a.bias[col] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z = acc[j][r] * (1.0f / A_SCALE) + b;
                    s1 += z;
                    s2 += z * z;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z), rs_c, vbase + 128u * (unsigned)j, srow(r) * ldc4, 2);
                }
                if (a.stats_part) {
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (g == 0) { s_red[0][wave][32 * j + c] = s1; s_red[1][wave][32 * j + c] = s2; }
                }
            }
        }
    } else {
    if constexpr (SUMS) {
        if (a.stats_part) fetch_z(0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + 32 * j + c;
        const bool col_ok = col < a.n_out;
        const float b = (a.bias && col_ok) ? a.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        if constexpr (SUMS) {
            if (a.stats_part) {
                // (tile j's z values are fetched while tile j - 1 is summed: see the loop head below)
                const bool st_ok = col < a.stats_ld;
                const float sc = st_ok ? a.coef_p[col] : 0.f, sh = st_ok ? a.coef_p[a.stats_ld + col] : 0.f;
                const float mean = st_ok ? a.coef_p[2 * a.stats_ld + col] : 0.f, rstd = st_ok ? a.coef_p[3 * a.stats_ld + col] : 0.f;
                float* c0 = a.c + (size_t)row0 * a.ldc + col;
                float zq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zq[r] = znext[r];
                if (j + 1 < NT) fetch_z(j + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * g;
                    const float v = acc[j][r] * (1.0f / A_SCALE);
                    if (lr < live && col_ok) __builtin_nontemporal_store(v, &c0[lr * a.ldc]);
                    if (lr < live && st_ok) {
                        const float g1 = fmaf(zq[r], sc, sh) > 0.f ? a.post_p * v : 0.f;
                        s1 += g1;
                        s2 += g1 * ((zq[r] - mean) * rstd);
                    }
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (g == 0) { s_red[0][wave][32 * j + c] = s1; s_red[1][wave][32 * j + c] = s2; }
                continue;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = row0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const float z = acc[j][r] * (1.0f / A_SCALE) + b;
            if (row < a.m && col_ok) {
                s1 += z;
                s2 += z * z;
                float y = z;
                if (a.act == ACT_TANH) y = tanhf(z);
                else if (a.act == ACT_SIGMOID) y = 1.0f / (1.0f + expf(-z));
                __builtin_nontemporal_store(y, &a.c[(size_t)row * a.ldc + col]);
            }
        }
        if (a.stats_part) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (g == 0) { s_red[0][wave][32 * j + c] = s1; s_red[1][wave][32 * j + c] = s2; }
        }
    }
    }       // (general epilogue)
    if (a.stats_part) {
        __syncthreads();
        for (int i = tid; i < 2 * NCOL; i += 256) {
            const int which = i / NCOL, n = i - which * NCOL;
            const int col = n0 + n;
            if (col < a.stats_ld) {          // (= n_out, or the summed layer's width in the SUMS form)
                const float s = s_red[which][0][n] + s_red[which][1][n] + s_red[which][2][n] + s_red[which][3][n];
                a.stats_part[((size_t)rb * 2 + which) * a.stats_ld + col] = s;
            }
        }
    }
    if constexpr (ARITH == 0) {
        // A rides at 64x its value as two f16 halves: from |a| = 1023.5 on the high half is infinite.  Report it like the fused f16x3
        // kernels report a saturated activation (vfn_f16x3_set_status): the facade's range guard then moves the model to the exact products.
        if (a.status && !(a_max < 1023.0f)) atomicOr(a.status, 1u);         // (also true for NaN)
    }
    if (PROBE && a.probe && tid == 0 && rb < 4096) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        mark(6);          // (the epilogue)
#pragma unroll
        for (int q = 0; q < 7; ++q) a.probe[rb * 8 + q] = ph[q];
    }
}

// VFN_GEMM_PROBE=1 (debugging aid, single-threaded use): wave 0 of every workgroup of the split kernels adds up the shader cycles
// (s_memtime) it spends in each phase of the chunk loop; every launch is then followed by a synchronisation and one line on stderr.
// This is what showed the launch to be bound by its HBM streams (profiles/r04/linear_rows_microbench.txt).
constexpr unsigned PROBE_BLOCKS = 4096;
unsigned long long* gemm_probe_buffer(hipStream_t s) {
    #ifdef VFN_GEMM_PROBE_BUILD
    static const bool on = getenv("VFN_GEMM_PROBE") != nullptr;
#else
    static const bool on = false;
#endif
    static unsigned long long* buf = nullptr;
    if (!on) return nullptr;
    if (!buf && hipMalloc(&buf, PROBE_BLOCKS * 8 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(buf, 0, PROBE_BLOCKS * 8 * sizeof(unsigned long long), s);
    return buf;
}
void gemm_probe_report(const GemmArgs& a, unsigned blocks, int trans, int arith, int sums, hipStream_t s) {
    unsigned long long* host = (unsigned long long*)malloc(PROBE_BLOCKS * 8 * sizeof(unsigned long long));
    if (!host) return;
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(host, a.probe, PROBE_BLOCKS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned nb = blocks < PROBE_BLOCKS ? blocks : PROBE_BLOCKS;
    for (unsigned b = 0; b < nb; ++b)
        for (int q = 0; q < 7; ++q) sum[q] += (double)host[b * 8 + q];
    fprintf(stderr, "gemm16 probe (TRANS=%d ARITH=%d SUMS=%d m=%lld n=%d k=%d): cycles per workgroup: barrier1 %.0f  wait_loads %.0f  "
            "stage %.0f  barrier2 %.0f  fetch_issue %.0f  compute %.0f  epilogue %.0f\n", trans, arith, sums, a.m, a.n_out, a.k_in,
            sum[0] / nb, sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb, sum[6] / nb);
    free(host);
}

// W -> its two 16-bit planes, once per call: out[block of 256 columns][chunk of 32 k][plane][256 n][32 k] (zeros past n_out / k_in)
template <int ARITH>
__global__ __launch_bounds__(256) void vfn_gemm_split_w_kernel(const float* w, int ldw, int trans, int n_out, int k_in, int chunks, unsigned short* out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;            // one (column, k) of one block and chunk
    const int k = idx & 31, n = (idx >> 5) & 255, bc = idx >> 13;
    const int ci = bc % chunks, nb = bc / chunks;
    const int col = nb * 256 + n, kk = ci * 32 + k;
    float v = 0.f;
    if (col < n_out && kk < k_in) v = trans ? w[(size_t)kk * ldw + col] : w[(size_t)col * ldw + kk];
    unsigned short hi, mid, lo;
    split3<ARITH>(v, hi, mid, lo);
    unsigned short* o = out + (size_t)bc * 2 * 8192 + n * 32 + k;
    o[0] = hi;
    o[8192] = mid;
}

template <bool TRANS, int ARITH, bool SUMS = false>
void launch_gemm16(GemmArgs a, hipStream_t s) {
    a.probe = gemm_probe_buffer(s);
    a.status = ARITH == 0 ? vfn_internal_f16x3_status() : nullptr;
    const unsigned blocks = (unsigned)((a.m + GM_ROWS - 1) / GM_ROWS);
    float* const stats = a.stats_part;
    if constexpr (ARITH != 2) {
        if (a.wp) {         // (the caller's scratch: vfn_linear_rows_wplanes_bytes)
            a.wp_chunks = (a.k_pad + GM_KC - 1) / GM_KC;
            const int nblocks = (a.n_out + 255) / 256;
            hipLaunchKernelGGL((vfn_gemm_split_w_kernel<ARITH>), dim3((unsigned)(nblocks * a.wp_chunks * 32)), dim3(256), 0, s, a.w, a.ldw, (int)TRANS,
                               a.n_out, a.k_in, a.wp_chunks, const_cast<unsigned short*>(a.wp));
        }
    } else {
        a.wp = nullptr;
    }
    for (int n0 = 0; n0 < a.n_out; n0 += 256) {
        a.n0 = n0;
        if (SUMS) a.stats_part = n0 == 0 ? stats : nullptr;      // the summed columns (<= 256) all sit in the first launch
        const int tiles = (min(a.n_out - n0, 256) + 31) / 32;
        if constexpr (!TRANS && !SUMS && ARITH == 0) {
            if (a.fold_coef) {             // (vfn_linear_rows_fold has checked: more than four tiles of output columns)
                // (W from its pre-split planes only: the form that fetches and splits W itself has no registers left for the coefficients)
                hipLaunchKernelGGL((vfn_linear_rows16_kernel<8, false, 0, false, true, true>), dim3(blocks), dim3(256), 0, s, a);
                continue;
            }
        }
        if constexpr (ARITH != 2) {
            if (a.wp && tiles > 4) {       // the 256-column launches read the planes; narrower ones split on the fly as before
                hipLaunchKernelGGL((vfn_linear_rows16_kernel<8, TRANS, ARITH, SUMS, true>), dim3(blocks), dim3(256), 0, s, a);
                continue;
            }
        }
        if (tiles > 4) hipLaunchKernelGGL((vfn_linear_rows16_kernel<8, TRANS, ARITH, SUMS>), dim3(blocks), dim3(256), 0, s, a);
        else if (tiles > 2) hipLaunchKernelGGL((vfn_linear_rows16_kernel<4, TRANS, ARITH, SUMS>), dim3(blocks), dim3(256), 0, s, a);
        else if (tiles > 1) hipLaunchKernelGGL((vfn_linear_rows16_kernel<2, TRANS, ARITH, SUMS>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((vfn_linear_rows16_kernel<1, TRANS, ARITH, SUMS>), dim3(blocks), dim3(256), 0, s, a);
    }
    if (a.probe) gemm_probe_report(a, blocks, (int)TRANS, ARITH, (int)SUMS, s);
}

template <bool TRANS>
void launch_gemm(GemmArgs a, hipStream_t s) {
    const unsigned blocks = (unsigned)((a.m + GM_ROWS - 1) / GM_ROWS);
    for (int n0 = 0; n0 < a.n_out; n0 += 256) {
        a.n0 = n0;
        const int tiles = (min(a.n_out - n0, 256) + 31) / 32;
        if (tiles > 4) hipLaunchKernelGGL((vfn_linear_rows_kernel<8, TRANS>), dim3(blocks), dim3(256), 0, s, a);
        else if (tiles > 2) hipLaunchKernelGGL((vfn_linear_rows_kernel<4, TRANS>), dim3(blocks), dim3(256), 0, s, a);
        else if (tiles > 1) hipLaunchKernelGGL((vfn_linear_rows_kernel<2, TRANS>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((vfn_linear_rows_kernel<1, TRANS>), dim3(blocks), dim3(256), 0, s, a);
    }
}

// ------------------------------------------------------------------------------------------------
// column sums: per-block partials [P][n_sets][n] (fp32) -> double [n_sets][n]
// ------------------------------------------------------------------------------------------------
__global__ void vfn_colsum_finish_kernel(const float* part, int n_parts, int width, double* sums) {
    __shared__ double s_acc[256];
    const int col = blockIdx.x;
    double acc = 0.0;
    for (int p = threadIdx.x; p < n_parts; p += 256) acc += (double)part[(size_t)p * width + col];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[col] = s_acc[0];
}

// sums[0][c] = sum z, sums[1][c] = sum z^2 over m rows -> coef[0..3][n] = scale, shift, mean, rstd; running stats
__global__ void vfn_bstat_finalize_kernel(const double* sums, long long m, int n, const float* gamma, const float* beta,
                                          float eps, float momentum, float* running_mean, float* running_var, float* coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const double mean = sums[c] / (double)m;
    double var = sums[n + c] / (double)m - mean * mean;      // biased (what normalises)
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * rstd;
    coef[c] = scale;
    coef[n + c] = beta[c] - (float)mean * scale;
    coef[2 * n + c] = (float)mean;
    coef[3 * n + c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = m > 1 ? var * ((double)m / (double)(m - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// ------------------------------------------------------------------------------------------------
// row-wise kernels: one thread per column (consecutive lanes = consecutive columns), EW_ROWS rows per workgroup
// ------------------------------------------------------------------------------------------------
constexpr int EW_ROWS = 64;

// 16-byte form of the two passes that only map rows (n a multiple of 4, 16-byte aligned rows): a lane owns four consecutive columns, the
// lanes of a workgroup cover 1024 / n... rows at a time; every access is a whole b128 (the one-column-per-thread forms reach 3.3-4 TB/s)
__global__ __launch_bounds__(256) void vfn_bstat_relu_rows4_kernel(const float* z, int ldz, const float* coef, long long m, int n,
                                                                    float post, float* h, int ldh) {
    const int q = n >> 2;                                   // 16-byte pieces per row
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    const int piece = threadIdx.x % q, rsub = threadIdx.x / q, rstep = 256 / q;
    if (rsub >= rstep) return;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(coef + 4 * piece), sh = *reinterpret_cast<const f32x4*>(coef + n + 4 * piece);
    for (long long r = r0 + rsub; r < r1; r += rstep) {
        const f32x4 zv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + (size_t)r * ldz + 4 * piece));
        f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = post * fmaxf(fmaf(zv[t], sc[t], sh[t]), 0.f);
        *reinterpret_cast<f32x4*>(h + (size_t)r * ldh + 4 * piece) = o;
    }
}

__global__ __launch_bounds__(256) void vfn_bstat_relu_bwd_rows4_kernel(const float* gr, int ldg, const float* z, int ldz, const float* coef,
                                                                        const double* sums, long long m, int n, float post, float* dz,
                                                                        int lddz) {
    const int q = n >> 2;
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    const int piece = threadIdx.x % q, rsub = threadIdx.x / q, rstep = 256 / q;
    if (rsub >= rstep) return;
    const int c0 = 4 * piece;
    const f32x4 scale = *reinterpret_cast<const f32x4*>(coef + c0), sh = *reinterpret_cast<const f32x4*>(coef + n + c0);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(coef + 2 * n + c0), rstd = *reinterpret_cast<const f32x4*>(coef + 3 * n + c0);
    f32x4 ga, gb;
#pragma unroll
    for (int t = 0; t < 4; ++t) { ga[t] = (float)(sums[c0 + t] / (double)m); gb[t] = (float)(sums[n + c0 + t] / (double)m); }
    for (long long r = r0 + rsub; r < r1; r += rstep) {
        const f32x4 zv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + (size_t)r * ldz + c0));
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gr + (size_t)r * ldg + c0));
        f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float g1 = fmaf(zv[t], scale[t], sh[t]) > 0.f ? post * gv[t] : 0.f;
            o[t] = scale[t] * (g1 - ga[t] - (zv[t] - mean[t]) * rstd[t] * gb[t]);
        }
        *reinterpret_cast<f32x4*>(dz + (size_t)r * lddz + c0) = o;
    }
}

__global__ __launch_bounds__(256) void vfn_bstat_relu_rows_kernel(const float* z, int ldz, const float* coef, long long m, int n,
                                                                   float post, float* h, int ldh) {
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    for (int c = threadIdx.x; c < n; c += 256) {
        const float sc = coef[c], sh = coef[n + c];
        for (long long r = r0; r < r1; ++r) h[(size_t)r * ldh + c] = post * fmaxf(fmaf(z[(size_t)r * ldz + c], sc, sh), 0.f);
    }
}

// The ReLU mask is re-derived from z with the forward's own expression (fmaf(z, scale, shift) > 0): one read less per pass
// than looking at the stored activation.
__global__ __launch_bounds__(256) void vfn_bstat_relu_bwd_sums_kernel(const float* gr, int ldg, const float* z, int ldz, const float* coef,
                                                                       long long m, int n, float post, float* part) {
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    for (int c = threadIdx.x; c < n; c += 256) {
        const float sc = coef[c], sh = coef[n + c], mean = coef[2 * n + c], rstd = coef[3 * n + c];
        float s1 = 0.f, s2 = 0.f;
        for (long long r = r0; r < r1; ++r) {
            const float zv = z[(size_t)r * ldz + c];
            const float g1 = fmaf(zv, sc, sh) > 0.f ? post * gr[(size_t)r * ldg + c] : 0.f;
            s1 += g1;
            s2 += g1 * ((zv - mean) * rstd);
        }
        part[((size_t)blockIdx.x * 2) * n + c] = s1;
        part[((size_t)blockIdx.x * 2 + 1) * n + c] = s2;
    }
}

__global__ __launch_bounds__(256) void vfn_bstat_relu_bwd_rows_kernel(const float* gr, int ldg, const float* z, int ldz, const float* coef,
                                                                       const double* sums, long long m, int n, float post, float* dz,
                                                                       int lddz) {
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    for (int c = threadIdx.x; c < n; c += 256) {
        const float scale = coef[c], sh = coef[n + c], mean = coef[2 * n + c], rstd = coef[3 * n + c];     // scale = gamma * rstd
        const float ga = (float)(sums[c] / (double)m), gb = (float)(sums[n + c] / (double)m);
        for (long long r = r0; r < r1; ++r) {
            const float zv = z[(size_t)r * ldz + c];
            const float g1 = fmaf(zv, scale, sh) > 0.f ? post * gr[(size_t)r * ldg + c] : 0.f;
            dz[(size_t)r * lddz + c] = scale * (g1 - ga - (zv - mean) * rstd * gb);
        }
    }
}

__global__ __launch_bounds__(256) void vfn_act_bwd_rows_kernel(int act, const float* dy, int lddy, const float* y, int ldy, long long m,
                                                                int n, int onehot, float* dz, int lddz) {
    const long long r0 = (long long)blockIdx.x * EW_ROWS, r1 = min(m, r0 + EW_ROWS);
    for (int c = threadIdx.x; c < n; c += 256) {
        for (long long r = r0; r < r1; ++r) {
            const float g = dy ? dy[(size_t)r * lddy + c] : (c == onehot ? 1.f : 0.f);
            const float v = y[(size_t)r * ldy + c];
            dz[(size_t)r * lddz + c] = act == ACT_TANH ? g * (1.f - v * v) : (act == ACT_SIGMOID ? g * v * (1.f - v) : g);
        }
    }
}

// dst[row][col0 + j] = scale * PE(src3[row])[j]: [x, sin(2^k x), cos(2^k x)]_{k<L} (models/helpers/embedder.py:11-37)
__global__ __launch_bounds__(256) void vfn_embed_rows_kernel(const float* src3, int ld_src, int rows_per_src, long long m, int multires,
                                                              float scale, float* dst, int ld_dst, int col0) {
    const long long row = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    if (row >= m) return;
    const float* sp = src3 + (size_t)(row / rows_per_src) * ld_src;
    const float x[3] = {sp[0], sp[1], sp[2]};
    float* out = dst + (size_t)row * ld_dst + col0;
    if (part == 0) { out[0] = scale * x[0]; out[1] = scale * x[1]; out[2] = scale * x[2]; }
    for (int q = part; q < 3 * multires; q += 4) {
        const int k = q / 3, cc = q - 3 * k;
        float s, cs;
        sincosf(x[cc] * (float)(1 << k), &s, &cs);
        out[3 + 6 * k + cc] = scale * s;
        out[3 + 6 * k + 3 + cc] = scale * cs;
    }
}

struct EmbedBwdPiece { const float* d; int ld, col0; float scale; };

// d_src[row][c] (+)= sum over pieces of scale * (d[c] + sum_k 2^k (cos(2^k x) d[3+6k+c] - sin(2^k x) d[3+6k+3+c]))
__global__ __launch_bounds__(256) void vfn_embed_rows_bwd_kernel(const float* src3, long long m, int multires, EmbedBwdPiece p0,
                                                                  EmbedBwdPiece p1, float* d_src3, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * m) return;
    const long long row = i / 3;
    const int c = (int)(i - 3 * row);
    const float x = src3[i];
    float total = accumulate ? d_src3[i] : 0.f;
    const EmbedBwdPiece ps[2] = {p0, p1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!ps[q].d) continue;
        const float* d = ps[q].d + (size_t)row * ps[q].ld + ps[q].col0;
        float acc = d[c];
        for (int k = 0; k < multires; ++k) {
            const float f = (float)(1 << k);
            float s, cs;
            sincosf(x * f, &s, &cs);
            acc += f * (cs * d[3 + 6 * k + c] - s * d[3 + 6 * k + 3 + c]);
        }
        total += ps[q].scale * acc;
    }
    d_src3[i] = total;
}

inline unsigned ew_blocks(long long m) { return (unsigned)((m + EW_ROWS - 1) / EW_ROWS); }
// rows of `ld` floats from a 16-byte aligned base, every row 16-byte aligned (the b128 forms of the row passes)
inline bool rows16(const void* p, int ld) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0; }

}  // namespace

extern "C" int64_t vfn_linear_rows_wplanes_bytes(int32_t n_out, int32_t k_in) {
    if (n_out < 1 || k_in < 1) return VFN_ERR_INVALID;
    const long long k_pad = (k_in + 7) & ~7;
    return ((long long)(n_out + 255) / 256) * ((k_pad + GM_KC - 1) / GM_KC) * 2 * 8192 * 2;
}

extern "C" int vfn_linear_rows(int32_t transpose_w, const float* a, int32_t lda, const float* w, int32_t ldw, const float* bias,
                               int64_t m, int32_t n_out, int32_t k_in, int32_t act, float* c, int32_t ldc, float* stats_part,
                               void* stream) {
    return vfn_linear_rows_ws(transpose_w, a, lda, w, ldw, bias, m, n_out, k_in, act, c, ldc, stats_part, nullptr, stream);
}

extern "C" int vfn_linear_rows_ws(int32_t transpose_w, const float* a, int32_t lda, const float* w, int32_t ldw, const float* bias,
                                  int64_t m, int32_t n_out, int32_t k_in, int32_t act, float* c, int32_t ldc, float* stats_part,
                                  void* wplanes, void* stream) {
    VFN_REQUIRE(a && w && c, "vfn_linear_rows: NULL argument");
    VFN_REQUIRE(n_out >= 1 && k_in >= 1 && act >= 0 && act <= 2, "vfn_linear_rows: n_out=%d k_in=%d act=%d", n_out, k_in, act);
    const int k_pad = (k_in + 7) & ~7;
    VFN_REQUIRE((lda & 3) == 0 && lda >= k_pad && ((uintptr_t)a & 15) == 0,
                "vfn_linear_rows: A needs 16-byte aligned rows with lda (%d) >= %d (k rounded up to 8; pad columns zero)", lda, k_pad);
    VFN_REQUIRE(ldc >= n_out && ldw >= ((transpose_w & 1) ? n_out : k_in), "vfn_linear_rows: ldc=%d ldw=%d too small", ldc, ldw);
    VFN_REQUIRE((long long)((transpose_w & 1) ? k_in : n_out) * ldw * 4 < (1ll << 31), "vfn_linear_rows: W larger than 2 GiB");
    if (m <= 0) return VFN_OK;
    GemmArgs g = {};
    g.a = a; g.w = w; g.bias = bias; g.c = c; g.stats_part = stats_part; g.m = m; g.lda = lda; g.ldw = ldw; g.ldc = ldc;
    g.n_out = n_out; g.k_in = k_in; g.k_pad = k_pad; g.act = act; g.stats_ld = n_out;
    g.wp = static_cast<const unsigned short*>(wplanes);
    const int arith = transpose_w & 6;           // 2: split f16 (22 bits); 4: split bf16 (16 bits); 6: bf16 in three parts (24 bits); 0: exact fp32
    hipStream_t s = (hipStream_t)stream;
    const bool tr = (transpose_w & 1) != 0;
    if (arith == 2) { if (tr) launch_gemm16<true, 0>(g, s); else launch_gemm16<false, 0>(g, s); }
    else if (arith == 4) { if (tr) launch_gemm16<true, 1>(g, s); else launch_gemm16<false, 1>(g, s); }
    else if (arith == 6) { if (tr) launch_gemm16<true, 2>(g, s); else launch_gemm16<false, 2>(g, s); }
    else if (tr) launch_gemm<true>(g, s);
    else launch_gemm<false>(g, s);
    return vfn_check_launch("vfn_linear_rows");
}

// The forward product of a training-mode layer with the previous layer's BatchNorm + ReLU folded into its operand read (GemmArgs::fold_coef).
extern "C" int vfn_linear_rows_fold(int32_t arith, const float* z_prev, int32_t ldz, const float* coef_prev, int32_t n_prev, float post_prev,
                                    const float* w, int32_t ldw, const float* bias, int64_t m, int32_t n_out, int32_t k_in, float* c, int32_t ldc,
                                    float* stats_part, void* wplanes, void* stream) {
    // (the bf16-in-three-parts form keeps three operand planes per K-block: with the fold's coefficients beside them it spills — a layer on
    //  that arithmetic, the skip layer, keeps the row pass)
    VFN_REQUIRE(arith == 2, "vfn_linear_rows_fold: arith = %d (2: three f16 products)", arith);
    VFN_REQUIRE(z_prev && coef_prev && w && c && wplanes, "vfn_linear_rows_fold: NULL argument (the scratch for W's planes is required)");
    VFN_REQUIRE(n_out > 128 && n_out <= 256 && k_in >= 1 && n_prev >= 1 && n_prev <= k_in,
                "vfn_linear_rows_fold: n_out = %d (129 .. 256), n_prev = %d (1 .. k_in = %d)", n_out, n_prev, k_in);
    const int k_pad = (k_in + 7) & ~7;
    VFN_REQUIRE((ldz & 3) == 0 && ldz >= k_pad && ((uintptr_t)z_prev & 15) == 0,
                "vfn_linear_rows_fold: z needs 16-byte aligned rows with ldz (%d) >= %d (k rounded up to 8; pad columns zero)", ldz, k_pad);
    VFN_REQUIRE(ldc >= n_out && ldw >= k_in, "vfn_linear_rows_fold: ldc=%d ldw=%d too small", ldc, ldw);
    if (m <= 0) return VFN_OK;
    GemmArgs g = {};
    g.a = z_prev; g.w = w; g.bias = bias; g.c = c; g.stats_part = stats_part; g.m = m; g.lda = ldz; g.ldw = ldw; g.ldc = ldc;
    g.n_out = n_out; g.k_in = k_in; g.k_pad = k_pad; g.act = ACT_NONE; g.stats_ld = n_out;
    g.wp = static_cast<const unsigned short*>(wplanes);
    g.fold_coef = coef_prev; g.fold_n = n_prev; g.fold_post = post_prev;
    launch_gemm16<false, 0>(g, (hipStream_t)stream);
    return vfn_check_launch("vfn_linear_rows_fold");
}

extern "C" int vfn_linear_rows_dx_sums(const float* dz, int32_t lddz, const float* w, int32_t ldw, int64_t m, int32_t n_out, int32_t k_in, float* c,
                                       int32_t ldc, const float* z_prev, int32_t ldz_prev, const float* coef_prev, int32_t n_prev, float post_prev,
                                       float* sums_part, int32_t arith, void* wplanes, void* stream) {
    VFN_REQUIRE(arith == 4 || arith == 6, "vfn_linear_rows_dx_sums: arith = %d (4: three bf16 products, 6: bf16 in three parts)", arith);
    VFN_REQUIRE(dz && w && c && z_prev && coef_prev && sums_part, "vfn_linear_rows_dx_sums: NULL argument");
    VFN_REQUIRE(n_out >= 1 && k_in >= 1, "vfn_linear_rows_dx_sums: n_out=%d k_in=%d", n_out, k_in);
    const int k_pad = (k_in + 7) & ~7;
    VFN_REQUIRE((lddz & 3) == 0 && lddz >= k_pad && ((uintptr_t)dz & 15) == 0,
                "vfn_linear_rows_dx_sums: dz needs 16-byte aligned rows with lddz (%d) >= %d (k rounded up to 8; pad columns zero)", lddz, k_pad);
    VFN_REQUIRE(ldc >= n_out && ldw >= n_out, "vfn_linear_rows_dx_sums: ldc=%d ldw=%d too small", ldc, ldw);
    VFN_REQUIRE((long long)k_in * ldw * 4 < (1ll << 31), "vfn_linear_rows_dx_sums: W larger than 2 GiB");
    VFN_REQUIRE(n_prev >= 1 && n_prev <= n_out && n_prev <= 256 && ldz_prev >= n_prev,
                "vfn_linear_rows_dx_sums: the summed columns are the first n_prev (%d) <= min(n_out, 256) of C", n_prev);
    if (m <= 0) return VFN_OK;
    GemmArgs g = {};
    g.a = dz; g.w = w; g.c = c; g.m = m; g.lda = lddz; g.ldw = ldw; g.ldc = ldc;
    g.n_out = n_out; g.k_in = k_in; g.k_pad = k_pad; g.act = ACT_NONE;
    g.stats_part = sums_part; g.stats_ld = n_prev; g.zp = z_prev; g.ldzp = ldz_prev; g.coef_p = coef_prev; g.post_p = post_prev;
    g.wp = static_cast<const unsigned short*>(wplanes);
    if (arith == 4) launch_gemm16<true, 1, true>(g, (hipStream_t)stream);
    else launch_gemm16<true, 2, true>(g, (hipStream_t)stream);
    return vfn_check_launch("vfn_linear_rows_dx_sums");
}

extern "C" int64_t vfn_linear_rows_stat_parts(int64_t m) { return m <= 0 ? 0 : (m + GM_ROWS - 1) / GM_ROWS; }
extern "C" int64_t vfn_bstat_row_parts(int64_t m) { return m <= 0 ? 0 : (m + EW_ROWS - 1) / EW_ROWS; }

extern "C" int vfn_colsum_finish(const float* part, int64_t n_parts, int32_t width, double* sums, void* stream) {
    VFN_REQUIRE(part && sums && width >= 1 && n_parts >= 0, "vfn_colsum_finish: bad argument");
    hipLaunchKernelGGL(vfn_colsum_finish_kernel, dim3(width), dim3(256), 0, (hipStream_t)stream, part, (int)n_parts, width, sums);
    return vfn_check_launch("vfn_colsum_finish");
}

extern "C" int vfn_bstat_finalize(const double* sums, int64_t m, int32_t n, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* coef, void* stream) {
    VFN_REQUIRE(sums && gamma && beta && coef && n >= 1 && m >= 1, "vfn_bstat_finalize: bad argument");
    hipLaunchKernelGGL(vfn_bstat_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, (long long)m, n,
                       gamma, beta, eps, momentum, running_mean, running_var, coef);
    return vfn_check_launch("vfn_bstat_finalize");
}

extern "C" int vfn_bstat_relu_rows(const float* z, int32_t ldz, const float* coef, int64_t m, int32_t n, float post_scale, float* h,
                                   int32_t ldh, void* stream) {
    VFN_REQUIRE(z && coef && h && n >= 1 && ldz >= n && ldh >= n, "vfn_bstat_relu_rows: bad argument");
    if (m <= 0) return VFN_OK;
    if (rows16(z, ldz) && rows16(h, ldh) && rows16(coef, n) && n <= 1024)
        hipLaunchKernelGGL(vfn_bstat_relu_rows4_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, z, ldz, coef, (long long)m, n,
                           post_scale, h, ldh);
    else
        hipLaunchKernelGGL(vfn_bstat_relu_rows_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, z, ldz, coef, (long long)m, n,
                           post_scale, h, ldh);
    return vfn_check_launch("vfn_bstat_relu_rows");
}

extern "C" int vfn_bstat_relu_bwd_sums(const float* g, int32_t ldg, const float* z, int32_t ldz, const float* coef, int64_t m, int32_t n,
                                       float post_scale, float* part, void* stream) {
    VFN_REQUIRE(g && z && coef && part && n >= 1, "vfn_bstat_relu_bwd_sums: bad argument");
    if (m <= 0) return VFN_OK;
    hipLaunchKernelGGL(vfn_bstat_relu_bwd_sums_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, g, ldg, z, ldz,
                       coef, (long long)m, n, post_scale, part);
    return vfn_check_launch("vfn_bstat_relu_bwd_sums");
}

extern "C" int vfn_bstat_relu_bwd_rows(const float* g, int32_t ldg, const float* z, int32_t ldz, const float* coef, const double* sums,
                                       int64_t m, int32_t n, float post_scale, float* dz, int32_t lddz, void* stream) {
    VFN_REQUIRE(g && z && coef && sums && dz && n >= 1, "vfn_bstat_relu_bwd_rows: bad argument");
    if (m <= 0) return VFN_OK;
    if (rows16(g, ldg) && rows16(z, ldz) && rows16(dz, lddz) && rows16(coef, n) && n <= 1024)
        hipLaunchKernelGGL(vfn_bstat_relu_bwd_rows4_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, g, ldg, z, ldz,
                           coef, sums, (long long)m, n, post_scale, dz, lddz);
    else
        hipLaunchKernelGGL(vfn_bstat_relu_bwd_rows_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, g, ldg, z, ldz,
                           coef, sums, (long long)m, n, post_scale, dz, lddz);
    return vfn_check_launch("vfn_bstat_relu_bwd_rows");
}

extern "C" int vfn_act_bwd_rows(int32_t act, const float* dy, int32_t lddy, const float* y, int32_t ldy, int64_t m, int32_t n,
                                int32_t onehot_col, float* dz, int32_t lddz, void* stream) {
    VFN_REQUIRE(y && dz && n >= 1 && act >= 0 && act <= 2, "vfn_act_bwd_rows: bad argument");
    VFN_REQUIRE(dy || (onehot_col >= 0 && onehot_col < n), "vfn_act_bwd_rows: dy is NULL and onehot_col=%d is not a column", onehot_col);
    if (m <= 0) return VFN_OK;
    hipLaunchKernelGGL(vfn_act_bwd_rows_kernel, dim3(ew_blocks(m)), dim3(256), 0, (hipStream_t)stream, act, dy, lddy, y, ldy,
                       (long long)m, n, onehot_col, dz, lddz);
    return vfn_check_launch("vfn_act_bwd_rows");
}

extern "C" int vfn_embed_rows(const float* src3, int32_t ld_src, int32_t rows_per_src, int64_t m, int32_t multires, float scale,
                              float* dst, int32_t ld_dst, int32_t col0, void* stream) {
    VFN_REQUIRE(src3 && dst && multires >= 0 && multires <= 16 && ld_src >= 3 && rows_per_src >= 1, "vfn_embed_rows: bad argument");
    VFN_REQUIRE(col0 >= 0 && col0 + 3 + 6 * multires <= ld_dst, "vfn_embed_rows: columns %d..%d exceed ld_dst=%d", col0,
                col0 + 3 + 6 * multires, ld_dst);
    if (m <= 0) return VFN_OK;
    hipLaunchKernelGGL(vfn_embed_rows_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src3, ld_src,
                       rows_per_src, (long long)m, multires, scale, dst, ld_dst, col0);
    return vfn_check_launch("vfn_embed_rows");
}

extern "C" int vfn_embed_rows_bwd(const float* src3, int64_t m, int32_t multires, const float* d_a, int32_t ld_a, int32_t col_a,
                                  float scale_a, const float* d_b, int32_t ld_b, int32_t col_b, float scale_b, float* d_src3,
                                  int32_t accumulate, void* stream) {
    VFN_REQUIRE(src3 && d_a && d_src3 && multires >= 0 && multires <= 16, "vfn_embed_rows_bwd: bad argument");
    if (m <= 0) return VFN_OK;
    const EmbedBwdPiece p0 = {d_a, ld_a, col_a, scale_a}, p1 = {d_b, ld_b, col_b, scale_b};
    hipLaunchKernelGGL(vfn_embed_rows_bwd_kernel, dim3((unsigned)((3 * m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src3,
                       (long long)m, multires, p0, p1, d_src3, accumulate);
    return vfn_check_launch("vfn_embed_rows_bwd");
}
