// vfn_bwd16.hip — the dX chain of the backward pass on the bf16 matrix cores.
//
// Same function as vfn_mlp_bwd_kernel (vfn_mlp_bwd.hip): autograd of models/vector_field/vector_field_network.py:177-208
// and rendering_network.py:62-108 in the shipped training regime (eval-mode BatchNorm,
// train/vector_field_nerf_train.py:140-141,252).  Starting from the gradients of the two 3-channel heads it walks the
// layers backwards, dX_l = dY_l W'_l, dY_{l-1} = dX_l * f'(saved_{l-1}), writes every dY slot for the weight-gradient
// kernels and the two head pre-activation gradients dz_rgb / dz_vec.
//
// It is the forward f16x3 kernel (vfn_mlp16.hip) run on the TRANSPOSED weights: X^T[k][m] = W'^T[k][n] dY^T[n][m], the
// A operand is a 32-row tile of W'^T from a three-slot LDS ring fed by LDS-DMA, the B operand is the gradient tile of
// the wave's 32 points, and a finished 32x32 accumulator is, register for register, two K-blocks of the next step's B
// operand — one wave per SIMD, both operand sets in the AGPRs, compile-time chunk tables, the same hand-pipelined
// chunk loop.  Differences:
//  * operands are split into two bf16 halves (hi = truncated, lo = rounded remainder: 16 significant bits) instead of two
//    f16 halves: gradients span fp32's exponent range and bf16 keeps it without any scaling; three products per
//    K-block, ~2^-16 relative error per product — the gradient tests hold the result to 1e-3 and observe ~1e-5;
//  * the epilogue of a tile multiplies by the activation derivative of the layer below, read from the training
//    forward's workspace (ReLU mask / 1 - tanh^2), adds the rank-3 update of a 3-channel head where one branches off,
//    stores the tile to its dY slot and splits it into the next operand.  Those workspace loads return into VGPRs, and
//    hipcc drains every outstanding LDS-DMA before the first use of such a load; so the epilogue of the previous tile
//    runs AFTER the ring hand-over of a chunk (whose vmcnt(0) the mask loads, issued half a chunk earlier, ride on),
//    and the DMA pieces of chunk c+2 are issued behind it.
#include <string.h>
#include <utility>
#include "vfn_common.h"
#include "vfn_plan.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

enum : int { BM_VEC = 0, BM_FULL = 1, BM_FUSED = 3, BM_F16S = 16, BM_P1 = 32 };     // bit 0: the feature block's step is present; bit 1: rendering net; bit 4: scaled f16 gradients
// bit 5 (BM_P1): ONE bf16 product per K-block, round-to-nearest hi halves only (8 significant bits per operand) — the chain of the
// opt-in 16-bit-native training mode (BASELINE.json configs[2], "bf16 MFMA MLPs"); only the hi planes of the pack are fetched.
enum : int { MASK_RELU = 0, MASK_TANH = 1 };

// ------------------------------------------------------------------------------------------------
// The 12 matrix steps of the shipped geometry, in chain order.  Step s multiplies the gradient tile by W'_layer^T:
// it reduces over the layer's outputs n (NB blocks of 16) and produces the gradient of its "act" inputs k (TILES tiles
// of 32), which are the outputs of the layer whose workspace slot is SLOT.
// ------------------------------------------------------------------------------------------------
constexpr int ST_NB[12] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 14, 16, 16};
constexpr int ST_TILES[12] = {8, 8, 8, 8, 8, 8, 8, 8, 7, 8, 8, 8};
// (workspace slot written by step s: 11, 10, 9 rendering hidden 2, 1, 0 | 8 features | 7 .. 0 VF hidden 7 .. 0)
constexpr int ST_NET[12] = {1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
constexpr int ST_LAYER[12] = {3, 2, 1, 0, 8, 7, 6, 5, 4, 3, 2, 1};        // reference Linear index inside its net
constexpr int chunk_kb(int nb) { return 2 * nb; }
constexpr int step_off_kb(int s) {       // offset inside the step's own pack (rendering steps 0..3, VF steps 4..11)
    int o = 0;
    for (int i = (s < 4 ? 0 : 4); i < s; ++i) o += ST_TILES[i] * chunk_kb(ST_NB[i]);
    return o;
}
constexpr int first_step(int mode) { return (mode & 2) ? 0 : ((mode & 1) ? 4 : 5); }
struct ChunkD { int net, off_kb, kb; };
constexpr ChunkD chunk_of(int mode, int c) {
    for (int s = first_step(mode); s < 12; ++s) {
        if (c < ST_TILES[s]) return {ST_NET[s], step_off_kb(s) + c * chunk_kb(ST_NB[s]), chunk_kb(ST_NB[s])};
        c -= ST_TILES[s];
    }
    return {0, 0, 0};
}
constexpr int first_chunk(int mode, int s) { int c = 0; for (int i = first_step(mode); i < s; ++i) c += ST_TILES[i]; return c; }
constexpr int rn_pack_kb() { return step_off_kb(3) + ST_TILES[3] * chunk_kb(ST_NB[3]); }
constexpr int vf_pack_kb() { return step_off_kb(11) + ST_TILES[11] * chunk_kb(ST_NB[11]); }

// ------------------------------------------------------------------------------------------------
// pack: W'^T tiles, bf16 split, fragment order
//   chunk (k tile ck) = [nb][plane hi|lo][lane][8 bf16]; lane (i = lane & 31, g = lane >> 5), element j:
//     k = 32 ck + i,   n = 16 nb + 8 (j >> 2) + 4 g + (j & 3)      (accumulator-as-operand order)
// ------------------------------------------------------------------------------------------------
struct PackTEntry {
    const float* w; const float* bn_w; const float* bn_var;
    uint32_t off_kb, nb, tiles;
    int32_t in_dim, row_off, n_valid, col_off, k_valid;
    float scale;
};
struct PackTArgs {
    PackTEntry e[8];
    int32_t n_entries;
    uint32_t total_words;
    uint32_t* out;
    int32_t rne;          // hi plane rounded to nearest even (single-product chain) instead of truncated (three-product chain: w - hi exact)
};

__global__ void vfn_pack_bwd16_kernel(PackTArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total_words) return;
    int ei = 0;
    for (int i = 1; i < a.n_entries; ++i)
        if (idx >= a.e[i].off_kb * 256u) ei = i;
    const PackTEntry& e = a.e[ei];
    const uint32_t chunk_words = e.nb * 2u * 256u;
    const uint32_t local = idx - e.off_kb * 256u;
    const uint32_t ck = local / chunk_words, cw = local % chunk_words;
    const uint32_t blk = cw >> 8, lane = (cw >> 2) & 63u, jp = cw & 3u;
    const uint32_t part = blk & 1u, nb = blk >> 1;
    const int g = (int)(lane >> 5);
    const int k = (int)(32u * ck + (lane & 31u));
    unsigned short halves[2];
    for (int q = 0; q < 2; ++q) {
        const int j = (int)(2u * jp) + q;
        const int n = 16 * (int)nb + 8 * (j >> 2) + 4 * g + (j & 3);
        float w = 0.f;
        if (n < e.n_valid && k < e.k_valid) {
            const int row = e.row_off + n;
            w = e.w[(size_t)row * e.in_dim + e.col_off + k] * e.scale;
            if (e.bn_w) w *= e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f);
        }
        unsigned u = __builtin_bit_cast(unsigned, w);
        if (a.rne) u = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)w) << 16;
        const float hi = __builtin_bit_cast(float, u & 0xffff0000u);
        const __bf16 lo = (__bf16)(w - hi);
        halves[q] = part ? __builtin_bit_cast(unsigned short, lo) : (unsigned short)(u >> 16);
    }
    a.out[idx] = (uint32_t)halves[0] | ((uint32_t)halves[1] << 16);
}

int check_shipped(int net_kind, const vfn_net_geom* g, const char* what) {
    bool ok;
    if (net_kind == VFN_NET_VF) {
        static const int out[9] = {256, 256, 256, 217, 256, 256, 256, 256, 259};
        ok = g->n_layers == 9 && g->skip_layer == 4 && g->feature_dims == 256 && g->multires == 6;
        for (int i = 0; ok && i < 9; ++i) ok = g->out_dims[i] == out[i] && (i == 8 || g->has_bn[i]);
        ok = ok && g->in_dims[0] == 39 && g->in_dims[4] == 256;
    } else {
        ok = g->n_layers == 5 && g->feature_dims == 256 && g->multires == 4 && g->in_dims[0] == 289 && g->out_dims[4] == 3;
        for (int i = 0; ok && i < 4; ++i) ok = g->out_dims[i] == 256 && g->has_bn[i];
    }
    if (!ok) { vfn_set_error("%s: the bf16 backward kernels are specialised for the shipped layer shapes", what); return VFN_ERR_UNSUPPORTED; }
    return VFN_OK;
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
#ifndef BW16_EARLY_EPI
#define BW16_EARLY_EPI 1
#endif
#ifndef BW16_LATE_STORES
#define BW16_LATE_STORES 1
#endif
#ifndef BW16_STORE_AUX
#define BW16_STORE_AUX 2          // cache policy bits of the dY stores: 2 = nt (streaming).  With fragment-ordered slots every store is
                                  // eight whole 128-byte lines; nt keeps 7 GB of write-once data from washing through the L2s:
                                  // training step 12.5 -> 11.5 ms (0 = default write-back)
#endif
#ifndef BW_RING
#define BW_RING 4                 // LDS ring slots of 32 KiB: a chunk's pieces are issued BW_RING - 1 chunks ahead
#endif
#define BW_SLOT_KB 32
#define BW_SLOT (BW_SLOT_KB * 64)       // uint4 elements per ring slot
#define BW_WAVES 4
#define BW_PTS 128
#define BW_PMAX ((BW_SLOT_KB + BW_WAVES - 1) / BW_WAVES)

typedef __attribute__((address_space(3))) void lds_void;

struct Bwd16Args {
    const uint4* vf_wt;       // transposed bf16 packs
    const uint4* rn_wt;
    const float* vf_head;     // [3][256] rows 0..2 of the VF net's last Linear
    const float* rn_head;     // [3][256] the rendering net's last Linear
    const float* feats;       // [M][256] the tanh'ed features (workspace slot 8, row-major fp32): the one slot read as values
    int dy_flags;             // bit 1: dY in FRAGMENT ORDER ([group of 32 points][tile, register quad][lane][16 B], see
                              // csrc/vfn_dwf.hip) instead of row-major [M][256]; bit 2 (fragment order only): stored as bf16 (8 B)
    const uint32_t* masks;    // [13][M][2][4] u32: sign bits of the saved ReLU outputs (written by the training forwards)
    float* dy;                // [13][M][256]
    const float* d_colors; const float* colors;      // [M,3]
    const float* d_vec; const float* vec;            // row stride vec_stride
    const float* d_feats;                            // BM_FULL: gradient wrt the tanh'ed features, row stride vec_stride
    float* dz_rgb; float* dz_vec;                    // [M,4]
    long long n_points;
    const int* n_dev;          // optional device-side count: the launch covers min(n_points, *n_dev) points (csrc/vfn_train.hip)
    int vec_stride;
    long long ws_first, ws_points;   // this launch's points are points ws_first .. of a workspace (feats, masks, dy, dz_*) sized for ws_points
};

struct X16 { bf8 hi[16]; bf8 lo[16]; };

struct Pipe {
    uint4* lds;
    const float* heads;              // LDS: [2][3][256] head weights (0 = vector head, 1 = rgb head)
    __amdgpu_buffer_rsrc_t vf_w, rn_w;
    const float* feats; float* dy;
    long long slot_floats; uint32_t slot_bytes;    // one dY slot
    uint32_t feat_bytes;             // M * 1024
    uint32_t voff;                   // feature loads (row-major): m * 1024 + 16 * (lane >> 5); out of range for m >= M
    uint32_t dvoff;                  // dY stores: row-major = voff; fragment order (m >> 5) * 32768 + 16 (8 as bf16) * lane
    uint32_t st_tile, st_q;          // byte strides of (tile, register quad) of a dY slot for fp32 stores (bf16: half)
    int dy16;                        // dY leaves as 1: bf16, 2: scaled f16 (csrc/vfn_dwf.hip, "dY form 3")
    uint32_t evoff;                  // scaled f16: this lane's exponent byte of tile 0, (m >> 5) * 32768 + 16384 + lane; out of range
                                     // past the last group (lanes of padding points in a live group write 255 = "all zero")
    int live;                        // this lane's point exists
    u32x4 mw[13];                    // this lane's sign-bit words, one per slot: tile t -> half t & 1 of dword t >> 1, bit r <-> register r
};

struct Carry {
    f32x16 pend;                     // accumulators of the previous tile
    f32x4v mask[4];                  // saved activations under the pending tile (register groups 4q..4q+3)
    bf8 fh0, fl0;                    // first fragments of the next chunk
};

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int NET, int OFF_KB, int SLOT>
__device__ __forceinline__ void dma_piece(const Pipe& p, int blk, int lane) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(NET ? p.rn_w : p.vf_w, (lds_void*)(p.lds + SLOT * BW_SLOT + blk * 64), 16, lane * 16,
                                             (OFF_KB + blk) * 1024, 0, 0);
}
// single-product launches fetch the hi planes (even blocks) only.  Every wave issues the SAME number of pieces (the hand-over's
// vmcnt immediate counts them): a chunk whose hi planes do not divide by the wave count repeats its first blocks (same bytes, same place).
constexpr int p1_pieces(int kb) { return ((kb / 2 + BW_WAVES - 1) / BW_WAVES) * BW_WAVES; }
__device__ __forceinline__ int p1_block(int idx, int kb) { return 2 * (idx < kb / 2 ? idx : idx - kb / 2); }
template <int MODE, int C, bool P1 = false>
__device__ __forceinline__ void dma_chunk(const Pipe& p, int wave, int lane) {
    constexpr ChunkD d = chunk_of(MODE, C);
    if constexpr (P1) {
#pragma unroll
        for (int i = 0; i * BW_WAVES < p1_pieces(d.kb); ++i) dma_piece<d.net, d.off_kb, C % BW_RING>(p, p1_block(wave + BW_WAVES * i, d.kb), lane);
    } else {
#pragma unroll
    for (int i = 0; i * BW_WAVES < d.kb; ++i)
        if (wave + BW_WAVES * i < d.kb) dma_piece<d.net, d.off_kb, C % BW_RING>(p, wave + BW_WAVES * i, lane);
    }
}
template <int MODE, int C, bool P1 = false>
__device__ __forceinline__ void prefetch_chunk(Carry& cy, const Pipe& p, int lane) {
    const uint4* cb = p.lds + (C % BW_RING) * BW_SLOT;
    cy.fh0 = __builtin_bit_cast(bf8, cb[0 * 64 + lane]);
    if constexpr (!P1) cy.fl0 = __builtin_bit_cast(bf8, cb[1 * 64 + lane]);
}

// workspace access: registers 4q..4q+3 of a tile = columns 32 TILE + 8 q + 4 (lane >> 5) .. +3 of this lane's point
template <int SLOT, int TILE>
__device__ __forceinline__ f32x4v load_group(const Pipe& p, int q) {
    static_assert(SLOT == 8, "only the tanh'ed feature block is read back as values");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.feats), 0, (int)p.feat_bytes, 0x00020000);
    return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)p.voff, (32 * TILE + 8 * q) * 4, 0));
}
// what the epilogue multiplies by: for ReLU layers 1 / 0 from the sign bits (no memory access), for the tanh'ed feature
// block the saved values themselves
template <int SLOT, int TILE, int MASK>
__device__ __forceinline__ f32x4v mask_group(const Pipe& p, int q) {
    if constexpr (MASK == MASK_RELU) {
        // all-ones / zero WORDS (one v_bfe_i32 per value), ANDed onto the gradient in epi_pair: two VALU instructions per value and no SGPR
        // pair in between — as 1.f / 0.f selected by compares it was five, with the wait states of the VALU-writes-SGPR hazard (round 5)
        const int w = (int)p.mw[SLOT][TILE >> 1];
        constexpr int b0 = 16 * (TILE & 1);
        int m0 = __builtin_amdgcn_sbfe(w, b0 + 4 * q, 1), m1 = __builtin_amdgcn_sbfe(w, b0 + 4 * q + 1, 1),
            m2 = __builtin_amdgcn_sbfe(w, b0 + 4 * q + 2, 1), m3 = __builtin_amdgcn_sbfe(w, b0 + 4 * q + 3, 1);
        asm("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));       // (opaque: or the AND below is folded back into compare + select)
        return f32x4v{__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1), __builtin_bit_cast(float, m2), __builtin_bit_cast(float, m3)};
    } else {
        return load_group<SLOT, TILE>(p, q);
    }
}
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
// scaled f16: the exponent byte (k + 113) and the scale 2^k of this lane's 16 values of a tile, max |v| 2^k in [2^14, 2^15)
__device__ __forceinline__ float tile_scale(const f32x16& v, int& b, bool live = true) {
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int r = 2; r < 16; r += 2) m = fmaxf(m, fmaxf(fabsf(v[r]), fabsf(v[r + 1])));
    // A non-finite gradient of a LIVE point (a diverging step) must not be masked.  NaNs beside finite values need nothing: fmaxf drops
    // them from the maximum, they are stored as f16 NaNs and stay NaN under whatever factor the consumer rescales them by (0 included).
    // What used to be lost is a lane whose maximum is Inf (or whose 16 values are all NaN): biased exponent 255 has no byte and the
    // lane left as "all zero".  Its values now leave as NaN under byte 239 (which does not move the slab's common scale), so the weight
    // gradients they enter come out non-finite as the reference's fp32 products would.  (Lanes past the last point may hold anything;
    // their byte is 255 and the consumer zeroes them.)
    // Bit arithmetic, no selects: as `if (biased == 255) return ...` — and as ternaries alike — hipcc made divergent (EXEC-masked) blocks of
    // it in the middle of every chunk's K loop.  zm / im: all ones when the lane's maximum is zero (or subnormal) / non-finite.
    const int biased = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
    const int bn = min(254 - biased, 239);                 // k + 113: every normal fp32 magnitude has its byte (below 2^-112 the scale saturates)
    const int zm = (biased - 1) >> 31, im = (254 - biased) >> 31;
    const int b_fin = bn | (zm & 255);                                      // zero -> 255 ("all zero")
    const int s_fin = (int)((unsigned)(bn + 14) << 23) & ~zm;               // zero -> scale 0
    const int b_inf = live ? 239 : 255, s_inf = live ? 0x7fc00000 : 0;      // (uniform per lane for the whole launch)
    b = (b_fin & ~im) | (b_inf & im);
    return __builtin_bit_cast(float, (s_fin & ~im) | (s_inf & im));
}
template <int SLOT, int TILE>
__device__ __forceinline__ void store_group(const Pipe& p, const f32x16& v, int q, float scale = 1.0f) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dy + (long long)SLOT * p.slot_floats, 0, (int)p.slot_bytes, 0x00020000);
    const f32x4v g = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    if (p.dy16 == 2) {                // f16 of the scaled values: 11 significant bits in ONE factor of dW = dY^T X
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const half4 h = __builtin_convertvector(g * scale, half4);
        // (scaled f16 exists in fragment order only: tile stride 4096 B, quad stride 1024 B as fp32, half that here — a CONSTANT scalar offset;
        //  through the runtime strides of `p` hipcc took it for a divergent value and wrapped every store in a waterfall loop, see below)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs, (int)p.dvoff, (TILE * 4096 + q * 1024) / 2, BW16_STORE_AUX);
    } else if (p.dy16) {              // bf16 (round to nearest even): half the bytes, 8 significant bits in ONE factor of dW = dY^T X
        typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const bf4 b = __builtin_convertvector(g, bf4);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b), rs, (int)p.dvoff, (int)((TILE * p.st_tile + q * p.st_q) >> 1), BW16_STORE_AUX);
    } else if (p.st_q == 1024u) {     // fragment order: whole lines per instruction, streaming stores
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)p.dvoff, (int)(TILE * p.st_tile + q * p.st_q), BW16_STORE_AUX);
    } else {                          // row-major: 32 lines x 32 bytes per instruction — the default write-back policy (nt costs +33 % there)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)p.dvoff, (int)(TILE * p.st_tile + q * p.st_q), 0);
    }
}
// scaled f16, spread over the K steps of the next tile: one register quad -> four f16 of the scaled values
__device__ __forceinline__ u32x2s pack_quad_f16s(const f32x16& v, int q, float scale) {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const f32x4v g = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    return __builtin_bit_cast(u32x2s, __builtin_convertvector(g * scale, half4));
}
// ... and the stores of a tile encoded that way: exponent byte(s) first, the four pieces last
template <int SLOT, int TILE>
__device__ __forceinline__ void store_tile_f16s(const Pipe& p, int b, const u32x2s (&eq)[4]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dy + (long long)SLOT * p.slot_floats, 0, (int)p.slot_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(p.live ? b : 255), rs, (int)p.evoff, TILE * 64, 0);
    if constexpr (SLOT == 3 && TILE == 6) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)255, rs, (int)p.evoff, 7 * 64, 0);   // see store_tile
#ifdef BW16_ABL_NOSTORE      // timing only: the encode stays, the pieces do not leave
    asm volatile("" :: "v"(eq[0]), "v"(eq[1]), "v"(eq[2]), "v"(eq[3]));
#else
    // Round 5: the scalar offsets are compile-time constants (scaled f16 = fragment order: 2048 B per tile, 512 B per register quad).  They
    // used to come from the Pipe's runtime strides, which hipcc did not know to be uniform: each of the four stores of every tile sat in a
    // waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec / store / s_cbranch_execnz: 245 loops in the vector-only kernel), i.e. four
    // basic-block boundaries at the end of every chunk with their conservative s_waitcnt's and no scheduling across them.
#pragma unroll
    for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_raw_buffer_store_b64(eq[q], rs, (int)p.dvoff, (TILE * 4096 + q * 1024) / 2, BW16_STORE_AUX);
#endif
}
// a whole finished tile: (scaled f16: the lane's exponent byte first, so that the four piece stores stay the youngest
// vector-memory operations of the chunk), then its four register quads
template <int SLOT, int TILE>
__device__ __forceinline__ void store_tile(const Pipe& p, const f32x16& v) {
    float scale = 1.0f;
    if (p.dy16 == 2) {
        int b;
        scale = tile_scale(v, b, p.live);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dy + (long long)SLOT * p.slot_floats, 0, (int)p.slot_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(p.live ? b : 255), rs, (int)p.evoff, TILE * 64, 0);
        // slot 3 (the 217-wide layer in front of the skip) has seven tiles: its eighth is never produced, and the weight-gradient
        // kernel takes the smallest exponent byte of a whole slab as its common scale — that tile must read as "all zero"
        if constexpr (SLOT == 3 && TILE == 6) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)255, rs, (int)p.evoff, 7 * 64, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) store_group<SLOT, TILE>(p, v, q, scale);
}

// One register pair of a finished tile: (+ head rank-3 update) * activation derivative -> pend (for the store) and the
// (hi, lo) bf16 halves of element pair (j, j+1) of an operand block.
//   HEAD: -1 none, 0 vector head, 1 rgb head; dz: this point's 3 head pre-activation gradients
template <int MASK, int HEAD, int TILE, bool P1 = false>
__device__ __forceinline__ void epi_pair(f32x16& pend, const f32x4v (&mask)[4], int pr, const Pipe& p, const float (&dz)[3], int g,
                                         bf8& hi, bf8& lo, int j) {
    float v0 = pend[2 * pr], v1 = pend[2 * pr + 1];
    if (HEAD >= 0) {
        const int r = 2 * pr, k = 32 * TILE + (r & 3) + 8 * (r >> 2) + 4 * g;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x2v w = *reinterpret_cast<const f32x2v*>(p.heads + (HEAD * 3 + c) * 256 + k);
            v0 = fmaf(dz[c], w[0], v0); v1 = fmaf(dz[c], w[1], v1);
        }
    }
    const float s0 = mask[pr >> 1][(2 * pr) & 3], s1 = mask[pr >> 1][(2 * pr + 1) & 3];
    if (MASK == MASK_RELU) {
        v0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & __builtin_bit_cast(unsigned, s0));
        v1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & __builtin_bit_cast(unsigned, s1));
    }
    else { v0 *= 1.0f - s0 * s0; v1 *= 1.0f - s1 * s1; }
    pend[2 * pr] = v0; pend[2 * pr + 1] = v1;
    if constexpr (P1) {          // single product: the bf16 rounding (to nearest even) is the operand
        typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
        const f32x2v v = {v0, v1};
        u32x4v hv = __builtin_bit_cast(u32x4v, hi);
        hv[j >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
        hi = __builtin_bit_cast(bf8, hv);
        return;
    }
    const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
    const unsigned hp = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f32x2v r = {v0 - __builtin_bit_cast(float, u0 & 0xffff0000u), v1 - __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const unsigned lp = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2));
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    u32x4v hv = __builtin_bit_cast(u32x4v, hi), lv = __builtin_bit_cast(u32x4v, lo);
    hv[j >> 1] = hp; lv[j >> 1] = lp;
    hi = __builtin_bit_cast(bf8, hv); lo = __builtin_bit_cast(bf8, lv);
}

// A whole tile outside the pipelined loop (the tiles a chain starts from, and the very last one).
template <int SLOT, int TILE, int MASK, int HEAD, bool SPLIT, bool P1 = false>
__device__ __forceinline__ void finish_tile_with(f32x16& v, const f32x4v (&mask)[4], const Pipe& p, const float (&dz)[3], int g, X16& xout) {
    bf8 hi[2], lo[2];
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) epi_pair<MASK, HEAD, TILE, P1>(v, mask, pr, p, dz, g, hi[pr >> 2], lo[pr >> 2], (pr & 3) * 2);
    store_tile<SLOT, TILE>(p, v);
    if (SPLIT) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            asm volatile("" : "+a"(hi[s]));
            xout.hi[2 * TILE + s] = hi[s];
            if constexpr (!P1) { asm volatile("" : "+a"(lo[s])); xout.lo[2 * TILE + s] = lo[s]; }
        }
    }
}
template <int SLOT, int TILE, int MASK, int HEAD, bool SPLIT, bool P1 = false>
__device__ __forceinline__ void finish_tile(f32x16& v, const Pipe& p, const float (&dz)[3], int g, X16& xout) {
    f32x4v mask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mask[q] = mask_group<SLOT, TILE, MASK>(p, q);
    finish_tile_with<SLOT, TILE, MASK, HEAD, SPLIT, P1>(v, mask, p, dz, g, xout);
}
// The eight tiles a chain starts from: ALL their saved activations are requested first (32 loads in one batch); with the
// loads inside the per-tile code the register pins of finish_tile (volatile asm) keep hipcc from hoisting them and every
// tile pays its own round trip to HBM before any matrix work exists to hide it.
template <int SLOT, int MASK>
__device__ __forceinline__ void load_start_masks(const Pipe& p, f32x4v (&mk)[8][4]) {
    static_for<8>([&](auto it) {
        constexpr int t = decltype(it)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) mk[t][q] = mask_group<SLOT, t, MASK>(p, q);
    });
}

// One matrix step: xout <- f'(saved) * (W'^T xin [+ head]) — NCH chunks (C0 .. of the launch), one 32-row tile each.
//   OSLOT / MASK / HEAD describe the tiles this step produces; P* the pending tile handed over by the previous step
//   (PT = its tile index: it becomes K-blocks 2 PT, 2 PT + 1 of `xpend` = this step's own input).
// MX = launch mode (bits 0-1: BM_*) | BM_F16S (the gradients leave as scaled f16: compile-time, so that the encode sits in the
// hand-scheduled steps without branches)
template <int MX, int C0, int NB, int NCH, int OSLOT, int MASK, int HEAD, int PSLOT, int PT, int PMASK, int PHEAD>
__device__ __forceinline__ void step16(const X16& xin, X16& xout, X16& xpend, Carry& cy, const Pipe& p, const float (&dzv)[3],
                                       const float (&dzc)[3], int wave, int lane) {
    constexpr int MODE = MX & 3;
    constexpr bool F16S = (MX & BM_F16S) != 0;
    constexpr bool P1 = (MX & BM_P1) != 0;
    constexpr int H = NB / 2;                         // hand-over after step H-1
#if BW16_EARLY_EPI
    // The pending tile's epilogue runs in the FIRST half of the chunk (its masks are register-resident sign bits; only the
    // tanh'ed feature tiles load values, and wait for them) and its dY stores go out right after the hand-over: they then have
    // a whole chunk to retire before the next vmcnt(0) instead of half of one.
    constexpr int E = H;                              // steps 0 .. E-1 carry the pending tile's epilogue
    constexpr int EPI0 = 0;                           // first epilogue step
    [[maybe_unused]] constexpr int ST0 = H;           // first store step (builds without BW16_LATE_STORES)
#else
    constexpr int E = NB - 2 - H;                     // steps H .. H+E-1 carry the pending tile's epilogue
    constexpr int EPI0 = H;
    [[maybe_unused]] constexpr int ST0 = H + E;
#endif
    constexpr int DSTEPS = NB - H;
    static_assert(2 * PT >= EPI0 + E || PT < 0, "the pending tile must be complete before it is read");
    const int g = lane >> 5;
    static_for<NCH>([&](auto ich) {
        constexpr int ch = decltype(ich)::value;
        constexpr int C = C0 + ch;
        constexpr ChunkD dcur = chunk_of(MODE, C), dnext = chunk_of(MODE, C + 1), ddma = chunk_of(MODE, C + BW_RING - 1);
        static_assert(dcur.kb == 2 * NB, "step shape and chunk table disagree");
        const uint4* cb = p.lds + (C % BW_RING) * BW_SLOT;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        bf8 fh[2], fl[2];
        fh[0] = cy.fh0;
        if constexpr (!P1) fl[0] = cy.fl0;
        bf8 ehi[2], elo[2];
        // scaled f16 gradients: the pending tile is encoded in the shadow of this tile's second half (scale at step H, one
        // register quad per step after it), so that the last step only issues the stores
        float esc = 1.0f;
        int eb = 255;
        u32x2s eq[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
        constexpr bool ENC_SPREAD = BW16_EARLY_EPI && BW16_LATE_STORES && H + 5 < NB;
        f32x4v mnext[4];
#pragma unroll
        for (int st = 0; st < NB; ++st) {
            if (st + 1 < NB) {
                fh[(st + 1) & 1] = __builtin_bit_cast(bf8, cb[(2 * (st + 1)) * 64 + lane]);
                if constexpr (!P1) fl[(st + 1) & 1] = __builtin_bit_cast(bf8, cb[(2 * (st + 1) + 1) * 64 + lane]);
            }
            const bf8 a_hi = fh[st & 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, xin.hi[st], acc, 0, 0, 0);
#ifndef ABL_P1
            if constexpr (!P1) {
                const bf8 a_lo = fl[st & 1];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, xin.lo[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, xin.hi[st], acc, 0, 0, 0);
            }
#endif
            // -- middle: ring hand-over.  The wave must see ITS pieces of chunk c+1 landed (issued in the previous chunk's second
            // half); the barrier then extends that to everybody's pieces and frees the slot of chunk c-1.  Vector-memory
            // operations retire in issue order (MI355X_MICROARCH.md, s_waitcnt), and the only operations issued AFTER those DMA
            // pieces are the previous chunk's dY stores (the very last instructions of its last step, below): the wait leaves
            // exactly those in flight, so a tile's stores have a chunk and a half to retire instead of stalling every hand-over —
            // the chain was bound by that store round trip, not by store bandwidth (3.8 ms -> see DESIGN.md).
            if (st == H - 1 && dnext.kb > 0) {
#if BW16_LATE_STORES
                // operations younger than this wave's pieces of chunk c+1, which may stay in flight: the four dY stores at the end
                // of a chunk (every chunk but the launch's first has a pending tile), and with a four-slot ring the pieces of
                // chunk c+2 between them.  (The tanh tiles' value loads sit among them too: uncounted, so the wait is merely
                // stricter there.)
                constexpr int ahead_kb = chunk_of(MODE, C - 1 + BW_RING - 1 > 0 ? C - 1 + BW_RING - 1 : 0).kb;
                constexpr int young = (C >= 1 ? 4 * (C - 1 > 0) : 0) +
                                      (BW_RING > 3 ? (C >= 1 ? (P1 ? p1_pieces(ahead_kb) : ahead_kb) / BW_WAVES : 0) + (C >= 2 ? 4 * (C - 2 > 0) : 0) : 0);
                static_assert(young < 64, "vmcnt is a 6-bit field");
                if constexpr (young == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(young) : "memory");
#else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                __builtin_amdgcn_s_barrier();
            }
            // -- second half: the pending tile's epilogue ...
            if (st >= EPI0 && st < EPI0 + E && (ch > 0 || PSLOT >= 0)) {
#pragma unroll
                for (int pr = (st - EPI0) * 8 / E; pr < (st - EPI0 + 1) * 8 / E; ++pr) {
                    const int sblk = pr >> 2, j = (pr & 3) * 2;
                    if (ch > 0) epi_pair<MASK, HEAD, (ch > 0 ? ch - 1 : 0), P1>(cy.pend, cy.mask, pr, p, HEAD == 1 ? dzc : dzv, g, ehi[sblk], elo[sblk], j);
                    else epi_pair<PMASK, PHEAD, (PT >= 0 ? PT : 0), P1>(cy.pend, cy.mask, pr, p, PHEAD == 1 ? dzc : dzv, g, ehi[sblk], elo[sblk], j);
                    if ((pr & 3) == 3) {
                        asm volatile("" : "+a"(ehi[sblk]));
                        if constexpr (!P1) asm volatile("" : "+a"(elo[sblk]));
                        if (ch > 0) { xout.hi[2 * (ch > 0 ? ch - 1 : 0) + sblk] = ehi[sblk]; if constexpr (!P1) xout.lo[2 * (ch > 0 ? ch - 1 : 0) + sblk] = elo[sblk]; }
                        else { xpend.hi[2 * (PT >= 0 ? PT : 0) + sblk] = ehi[sblk]; if constexpr (!P1) xpend.lo[2 * (PT >= 0 ? PT : 0) + sblk] = elo[sblk]; }
                    }
                }
            }
            // ... its dY stores, the mask loads of the tile being computed, and the DMA pieces of chunk c+2
#if !BW16_LATE_STORES
            if (st >= ST0 && st < ST0 + 2 && (ch > 0 || PSLOT >= 0)) {
#pragma unroll
                for (int q = (st - ST0) * 2; q < (st - ST0 + 1) * 2; ++q) {
                    if (ch > 0) store_group<OSLOT, (ch > 0 ? ch - 1 : 0)>(p, cy.pend, q);
                    else store_group<(PSLOT >= 0 ? PSLOT : 0), (PT >= 0 ? PT : 0)>(p, cy.pend, q);
                }
            }
#endif
            if (ENC_SPREAD && F16S && (ch > 0 || PSLOT >= 0)) {
                if (st == H) esc = tile_scale(cy.pend, eb, p.live);
                if (st > H && st <= H + 4) eq[st - H - 1] = pack_quad_f16s(cy.pend, st - H - 1, esc);
            }
            if (st >= H && st < H + 4) mnext[st - H] = mask_group<OSLOT, ch, MASK>(p, st - H);
            if (P1 && st >= H && ddma.kb > 0) {
                constexpr int NP = p1_pieces(ddma.kb) / BW_WAVES;        // pieces per wave
#pragma unroll
                for (int i = (st - H) * NP / DSTEPS; i < (st - H + 1) * NP / DSTEPS; ++i)
                    dma_piece<ddma.net, ddma.off_kb, (C + BW_RING - 1) % BW_RING>(p, p1_block(wave + BW_WAVES * i, ddma.kb), lane);
            }
            if (!P1 && st >= H && ddma.kb > 0) {
#pragma unroll
                for (int i = (st - H) * BW_PMAX / DSTEPS; i < (st - H + 1) * BW_PMAX / DSTEPS; ++i) {
                    if (BW_WAVES * i + BW_WAVES <= ddma.kb) dma_piece<ddma.net, ddma.off_kb, (C + BW_RING - 1) % BW_RING>(p, wave + BW_WAVES * i, lane);
                    else if (BW_WAVES * i < ddma.kb) { if (wave + BW_WAVES * i < ddma.kb) dma_piece<ddma.net, ddma.off_kb, (C + BW_RING - 1) % BW_RING>(p, wave + BW_WAVES * i, lane); }
                }
            }
            if (st == NB - 1 && dnext.kb > 0) prefetch_chunk<MODE, (dnext.kb > 0 ? C + 1 : C), P1>(cy, p, lane);
#if BW16_LATE_STORES
            // the pending tile's dY: the LAST vector-memory instructions of the chunk (see the hand-over)
            if (st == NB - 1 && (ch > 0 || PSLOT >= 0)) {
                __builtin_amdgcn_sched_barrier(0);
                if (ENC_SPREAD && F16S) {
                    if (ch > 0) store_tile_f16s<OSLOT, (ch > 0 ? ch - 1 : 0)>(p, eb, eq);
                    else store_tile_f16s<(PSLOT >= 0 ? PSLOT : 0), (PT >= 0 ? PT : 0)>(p, eb, eq);
                } else {
                    if (ch > 0) store_tile<OSLOT, (ch > 0 ? ch - 1 : 0)>(p, cy.pend);
                    else store_tile<(PSLOT >= 0 ? PSLOT : 0), (PT >= 0 ? PT : 0)>(p, cy.pend);
                }
            }
#endif
            // (Explicit instruction groups here — one MFMA, then n VALU instructions in its shadow, as the forward kernel's K steps have them —
            //  were measured in round 5 with n = 6, 8, 12: the vector-only chain within 0.5 % of the compiler's own order, the fused one 1-3 %
            //  slower.  Not the lever.)
            __builtin_amdgcn_sched_barrier(0);
        }
        cy.pend = acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) cy.mask[q] = mnext[q];
    });
}

template <int MX>
__global__ __launch_bounds__(256, 1) void vfn_bwd16_kernel(const Bwd16Args a) {
    constexpr int MODE = MX & 3;
    constexpr bool P1 = (MX & BM_P1) != 0;
    __shared__ __attribute__((aligned(16))) uint4 s_ring[BW_RING * BW_SLOT + 2 * 3 * 256 / 4];
    float* s_heads = reinterpret_cast<float*>(s_ring + BW_RING * BW_SLOT);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5;
    const long long m = (long long)blockIdx.x * BW_PTS + wave * 32 + (lane & 31);
    long long n_live = a.n_points;
    if (a.n_dev) {             // (uniform; before any barrier)
        const long long nd = (long long)*a.n_dev;
        n_live = nd < n_live ? nd : n_live;
        if ((long long)blockIdx.x * BW_PTS >= n_live) return;
    }
    const bool in = m < n_live;

    // head weights -> LDS; this point's head gradients
    for (int i = tid; i < 768; i += 256) {
        s_heads[i] = a.vf_head[i];
        s_heads[768 + i] = (MODE & 2) ? a.rn_head[i] : 0.f;
    }
    float dzv[3] = {0.f, 0.f, 0.f}, dzc[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = a.vec[m * a.vec_stride + c];
            dzv[c] = a.d_vec[m * a.vec_stride + c] * (1.0f - t * t);
            if (MODE & 2) {
                const float cc = a.colors[m * 3 + c];
                dzc[c] = a.d_colors[m * 3 + c] * cc * (1.0f - cc);
            }
        }
        if (g == 0) {
            *reinterpret_cast<f32x4v*>(a.dz_vec + (m + a.ws_first) * 4) = f32x4v{dzv[0], dzv[1], dzv[2], 0.f};
            if (MODE & 2) *reinterpret_cast<f32x4v*>(a.dz_rgb + (m + a.ws_first) * 4) = f32x4v{dzc[0], dzc[1], dzc[2], 0.f};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    Pipe p;
    p.lds = s_ring; p.heads = s_heads;
    p.vf_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.vf_wt), 0, vf_pack_kb() * 1024, 0x00020000);
    p.rn_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>((MODE & 2) ? a.rn_wt : a.vf_wt), 0,
                                               ((MODE & 2) ? rn_pack_kb() : vf_pack_kb()) * 1024, 0x00020000);
    const bool frag = (a.dy_flags & 2) != 0;
    p.dy16 = (MX & BM_F16S) ? 2 : ((frag && (a.dy_flags & 4)) ? 1 : 0);
    p.live = in ? 1 : 0;
    const long long mw = m + a.ws_first;          // this point's place in the workspace
    // (descriptors of the dY slots and of the feature rows start at THIS LAUNCH's first point — 1 KiB per point in either order — and the
    //  32-bit offsets are launch-relative: a launch covers < 2^21 points, the workspace may hold more; csrc/vfn_mlp16.hip does the same)
    p.evoff = (m & ~31ll) < n_live ? (uint32_t)((m >> 5) * 32768 + 16384 + lane) : 0xc0000000u;
    p.feats = a.feats + a.ws_first * 256; p.dy = a.dy + a.ws_first * 256;
    {
        const long long left = (a.ws_points - a.ws_first) * 1024;
        p.feat_bytes = (uint32_t)(left < 0x7fffffffll ? left : 0x7fffffffll);
    }
    p.slot_floats = frag ? ((a.ws_points + 31) >> 5) * 8192 : a.ws_points * 256;
    {
        const long long left = p.slot_floats * 4 - a.ws_first * 1024;
        p.slot_bytes = (uint32_t)(left < 0x7fffffffll ? left : 0x7fffffffll);
    }
    p.voff = in ? (uint32_t)(m * 1024 + g * 16) : 0xfffffff0u;
    // (fragment order adds scalar offsets of up to 32 KiB: the out-of-range value must not wrap; a launch is limited to 2 GiB per slot)
    p.dvoff = !frag ? p.voff : (in ? (uint32_t)((m >> 5) * 32768 + lane * (p.dy16 ? 8 : 16)) : 0xc0000000u);
    // (readfirstlane: the strides become scalar offsets of the dY stores — as values hipcc knows to be uniform, or every store is a waterfall loop)
    p.st_tile = (uint32_t)__builtin_amdgcn_readfirstlane(frag ? 4096 : 128); p.st_q = (uint32_t)__builtin_amdgcn_readfirstlane(frag ? 1024 : 32);
    {   // sign bits of every ReLU layer this launch walks through: 16 bytes per slot, all requested now, in registers for good
        const unsigned mbytes = (unsigned)(a.ws_points * 32);
        const unsigned mvoff = in ? (unsigned)((2 * mw + g) * 16) : 0xfffffff0u;
        static_for<13>([&](auto is) {
            constexpr int sl = decltype(is)::value;
            constexpr bool used = sl != 8 && (sl < 8 || MODE == BM_FUSED);
            if constexpr (used) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<unsigned char*>(const_cast<uint32_t*>(a.masks)) + (size_t)sl * mbytes, 0, (int)mbytes, 0x00020000);
                p.mw[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs, mvoff, 0, 0);
            } else {
                p.mw[sl] = u32x4{0, 0, 0, 0};
            }
        });
    }
    dma_chunk<MODE, 0, P1>(p, wave, lane);
    dma_chunk<MODE, 1, P1>(p, wave, lane);
    if (BW_RING > 3) dma_chunk<MODE, 2, P1>(p, wave, lane);
    __syncthreads();                       // head tables visible (this also waits for the two chunks: once, harmless)

    // ---- the tiles the chain starts from: no matrix product, just the head update / the caller's gradient ----
    X16 xa, xb;
    if constexpr (MODE == BM_FUSED) {
        // gradient wrt the last hidden output of the rendering net = rank-3 update from the rgb head, ReLU-masked (slot 12)
        f32x4v mk[8][4];
        load_start_masks<12, MASK_RELU>(p, mk);
        static_for<8>([&](auto it) {
            constexpr int t = decltype(it)::value;
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
            finish_tile_with<12, t, MASK_RELU, 1, true, P1>(v, mk[t], p, dzc, g, xa);
        });
    } else if constexpr (MODE == BM_FULL) {
        // dZ_f = dF * (1 - F^2) straight from the caller's gradient (slot 8)
        f32x4v mk[8][4];
        load_start_masks<8, MASK_TANH>(p, mk);
        static_for<8>([&](auto it) {
            constexpr int t = decltype(it)::value;
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
                v[r] = in ? a.d_feats[m * a.vec_stride + k] : 0.f;
            }
            finish_tile_with<8, t, MASK_TANH, -1, true, P1>(v, mk[t], p, dzv, g, xa);
        });
    } else {
        // vector-only: gradient wrt the last plain hidden output = rank-3 update from the vector head (slot 7)
        f32x4v mk[8][4];
        load_start_masks<7, MASK_RELU>(p, mk);
        static_for<8>([&](auto it) {
            constexpr int t = decltype(it)::value;
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
            finish_tile_with<7, t, MASK_RELU, 0, true, P1>(v, mk[t], p, dzv, g, xa);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunks 0 and 1 landed, start tiles stored
    __builtin_amdgcn_s_barrier();
    Carry cy;
#pragma unroll
    for (int r = 0; r < 16; ++r) cy.pend[r] = 0.f;
    prefetch_chunk<MODE, 0, P1>(cy, p, lane);

    constexpr int R = MASK_RELU, T = MASK_TANH;
    // step16<MODE, C0, NB, NCH, OSLOT, MASK, HEAD, PSLOT, PT, PMASK, PHEAD>(xin, xout, xpend, ...)
    if constexpr (MODE == BM_FUSED) {
        step16<MX, first_chunk(MODE, 0), 16, 8, 11, R, -1, -1, -1, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);   // through R3
        step16<MX, first_chunk(MODE, 1), 16, 8, 10, R, -1, 11, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);    // R2
        step16<MX, first_chunk(MODE, 2), 16, 8, 9, R, -1, 10, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);     // R1
        step16<MX, first_chunk(MODE, 3), 16, 8, 8, T, -1, 9, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);      // R0 -> features
        step16<MX, first_chunk(MODE, 4), 16, 8, 7, R, 0, 8, 7, T, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);       // feature block + vector head
    } else if constexpr (MODE == BM_FULL) {
        step16<MX, first_chunk(MODE, 4), 16, 8, 7, R, 0, -1, -1, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);
    }
    if constexpr (MODE != BM_VEC) {
        // gradient of VF hidden 7 is in xb
        step16<MX, first_chunk(MODE, 5), 16, 8, 6, R, -1, 7, 7, R, 0>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);       // L7
        step16<MX, first_chunk(MODE, 6), 16, 8, 5, R, -1, 6, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);      // L6
        step16<MX, first_chunk(MODE, 7), 16, 8, 4, R, -1, 5, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);      // L5
        step16<MX, first_chunk(MODE, 8), 16, 7, 3, R, -1, 4, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);      // L4 (skip): 217 inputs
        step16<MX, first_chunk(MODE, 9), 14, 8, 2, R, -1, 3, 6, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);      // L3
        step16<MX, first_chunk(MODE, 10), 16, 8, 1, R, -1, 2, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);     // L2
        step16<MX, first_chunk(MODE, 11), 16, 8, 0, R, -1, 1, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);     // L1
        finish_tile<0, 7, R, -1, false>(cy.pend, p, dzv, g, xa);
    } else {
        // vector-only: the start tiles (gradient of VF hidden 7) are in xa
        step16<MX, first_chunk(MODE, 5), 16, 8, 6, R, -1, -1, -1, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 6), 16, 8, 5, R, -1, 6, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 7), 16, 8, 4, R, -1, 5, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 8), 16, 7, 3, R, -1, 4, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 9), 14, 8, 2, R, -1, 3, 6, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 10), 16, 8, 1, R, -1, 2, 7, R, -1>(xb, xa, xb, cy, p, dzv, dzc, wave, lane);
        step16<MX, first_chunk(MODE, 11), 16, 8, 0, R, -1, 1, 7, R, -1>(xa, xb, xa, cy, p, dzv, dzc, wave, lane);
        finish_tile<0, 7, R, -1, false>(cy.pend, p, dzv, g, xa);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int64_t vfn_packed_bwd16_size(int32_t net_kind, const vfn_net_geom* geom) {
    VFN_REQUIRE(geom, "vfn_packed_bwd16_size: NULL argument");
    int rc = check_shipped(net_kind, geom, "vfn_packed_bwd16_size");
    if (rc != VFN_OK) return rc;
    return (int64_t)(net_kind == VFN_NET_VF ? vf_pack_kb() : rn_pack_kb()) * 1024;
}

extern "C" int vfn_pack_weights_bwd16(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                                      void* packed, void* stream) {
    return vfn_pack_weights_bwd16_mode(net_kind, geom, layers, 0, packed, stream);
}

extern "C" int vfn_pack_weights_bwd16_mode(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers, int32_t round_hi,
                                           void* packed, void* stream) {
    VFN_REQUIRE(geom && layers && packed, "vfn_pack_weights_bwd16: NULL argument");
    int rc = check_shipped(net_kind, geom, "vfn_pack_weights_bwd16");
    if (rc != VFN_OK) return rc;
    PackTArgs a;
    memset(&a, 0, sizeof(a));
    const int s_lo = net_kind == VFN_NET_VF ? 4 : 0, s_hi = net_kind == VFN_NET_VF ? 12 : 4;
    for (int s = s_lo; s < s_hi; ++s) {
        const int li = ST_LAYER[s];
        const vfn_layer_params& q = layers[li];
        VFN_REQUIRE(q.weight, "vfn_pack_weights_bwd16: layer %d has NULL weight", li);
        PackTEntry& e = a.e[a.n_entries++];
        e.w = q.weight;
        if (geom->has_bn[li]) {
            VFN_REQUIRE(q.bn_weight && q.bn_var, "vfn_pack_weights_bwd16: layer %d BatchNorm pointer NULL", li);
            e.bn_w = q.bn_weight; e.bn_var = q.bn_var;
        }
        e.off_kb = (uint32_t)step_off_kb(s); e.nb = (uint32_t)ST_NB[s]; e.tiles = (uint32_t)ST_TILES[s];
        e.in_dim = geom->in_dims[li]; e.scale = 1.0f; e.row_off = 0; e.col_off = 0;
        e.n_valid = geom->out_dims[li]; e.k_valid = 256;
        if (net_kind == VFN_NET_VF) {
            if (li == geom->n_layers - 1) { e.row_off = 3; e.n_valid = geom->feature_dims; }          // feature rows of the last Linear
            if (li == geom->skip_layer) { e.k_valid = geom->out_dims[li - 1]; e.scale = 0.70710678118654752440f; }
        } else if (li == 0) {
            e.col_off = geom->in_dims[0] - geom->feature_dims;                                         // [p, PE(d), n | features]
        }
    }
    a.total_words = (uint32_t)(net_kind == VFN_NET_VF ? vf_pack_kb() : rn_pack_kb()) * 256u;
    a.out = (uint32_t*)packed;
    a.rne = round_hi ? 1 : 0;
    hipLaunchKernelGGL(vfn_pack_bwd16_kernel, dim3((a.total_words + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_pack_weights_bwd16");
}

extern "C" int vfn_mlp_bwd_chain_bf16(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                      const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                      const float* saved, const uint32_t* masks, float* dy, const float* d_colors, const float* colors,
                                      const float* d_vec, const float* vec, const float* d_feats, int32_t vec_stride,
                                      int64_t n_points, float* dz_rgb, float* dz_vec, void* stream) {
    // row-major workspace: the features are slot 8 of saved[13][M][256]
    return vfn_mlp_bwd_chain_bf16_ws(vf_geom, vf_packed_bwd16, vf_head_w, rn_geom, rn_packed_bwd16, rn_head_w,
                                     saved ? saved + (size_t)8 * (size_t)(n_points > 0 ? n_points : 0) * 256 : nullptr, masks, dy, 0, d_colors, colors,
                                     d_vec, vec, d_feats, vec_stride, n_points, dz_rgb, dz_vec, stream);
}

extern "C" int vfn_mlp_bwd_chain_bf16_ws(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                         const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                         const float* saved, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                                         const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                                         int32_t vec_stride, int64_t n_points, float* dz_rgb, float* dz_vec, void* stream) {
    return vfn_mlp_bwd_chain_bf16_ws_at(vf_geom, vf_packed_bwd16, vf_head_w, rn_geom, rn_packed_bwd16, rn_head_w, saved, masks, dy, dy_flags, d_colors,
                                        colors, d_vec, vec, d_feats, vec_stride, n_points, dz_rgb, dz_vec, 0, n_points, stream);
}

extern "C" int vfn_mlp_bwd_chain_bf16_ws_at(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                            const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                            const float* saved, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                                            const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                                            int32_t vec_stride, int64_t n_points, float* dz_rgb, float* dz_vec, int64_t ws_first,
                                            int64_t ws_points, void* stream) {
    return vfn_internal_bwd_chain_bf16_ws_at(vf_geom, vf_packed_bwd16, vf_head_w, rn_geom, rn_packed_bwd16, rn_head_w, saved, masks, dy, dy_flags, d_colors,
                                             colors, d_vec, vec, d_feats, vec_stride, n_points, nullptr, dz_rgb, dz_vec, ws_first, ws_points, stream);
}

int vfn_internal_bwd_chain_bf16_ws_at(const vfn_net_geom* vf_geom, const void* vf_packed_bwd16, const float* vf_head_w,
                                      const vfn_net_geom* rn_geom, const void* rn_packed_bwd16, const float* rn_head_w,
                                      const float* saved, const uint32_t* masks, void* dy, int32_t dy_flags, const float* d_colors,
                                      const float* colors, const float* d_vec, const float* vec, const float* d_feats,
                                      int32_t vec_stride, int64_t n_points, const int32_t* n_dev, float* dz_rgb, float* dz_vec,
                                      int64_t ws_first, int64_t ws_points, void* stream) {
    VFN_REQUIRE(vf_geom, "vfn_mlp_bwd_chain_bf16: NULL argument");
    int rc = check_shipped(VFN_NET_VF, vf_geom, "vfn_mlp_bwd_chain_bf16");
    if (rc != VFN_OK) return rc;
    const bool fused = rn_geom != nullptr;
    if (fused) {
        rc = check_shipped(VFN_NET_RENDER, rn_geom, "vfn_mlp_bwd_chain_bf16");
        if (rc != VFN_OK) return rc;
        VFN_REQUIRE(rn_packed_bwd16 && rn_head_w && d_colors && colors && dz_rgb, "vfn_mlp_bwd_chain_bf16: NULL rendering-net argument");
    }
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed_bwd16 && vf_head_w && saved && masks && dy && d_vec && vec && dz_vec, "vfn_mlp_bwd_chain_bf16: NULL argument");
    VFN_REQUIRE(vec_stride >= 3, "vfn_mlp_bwd_chain_bf16: vec_stride must be >= 3");
    VFN_REQUIRE(ws_first >= 0 && ws_first + n_points <= ws_points, "vfn_mlp_bwd_chain_bf16: points %lld .. %lld outside a workspace of %lld",
                (long long)ws_first, (long long)(ws_first + n_points), (long long)ws_points);
    VFN_REQUIRE(!(dy_flags & 2) || ws_first % 32 == 0, "vfn_mlp_bwd_chain_bf16: ws_first must be a multiple of 32 in fragment order");
    VFN_REQUIRE(n_points < (1ll << 21) && ws_points < (1ll << 26), "vfn_mlp_bwd_chain_bf16: at most 2097151 points per launch (32-bit slot offsets) and "
                "67108863 per workspace (got %lld in %lld)", (long long)n_points, (long long)ws_points);
    VFN_REQUIRE(!(dy_flags & 12) || (dy_flags & 2), "vfn_mlp_bwd_chain_bf16: 16-bit gradients need the fragment-ordered layout");
    VFN_REQUIRE((dy_flags & 12) != 12, "vfn_mlp_bwd_chain_bf16: dy_flags asks for bf16 AND scaled f16 gradients");
#if !BW16_LATE_STORES
    VFN_REQUIRE(!(dy_flags & 8), "vfn_mlp_bwd_chain_bf16: this build stores a tile's quads in separate steps; scaled f16 gradients need BW16_LATE_STORES");
#endif
    Bwd16Args a = {};
    a.vf_wt = (const uint4*)vf_packed_bwd16; a.rn_wt = (const uint4*)rn_packed_bwd16; a.vf_head = vf_head_w; a.rn_head = rn_head_w;
    a.feats = saved; a.dy_flags = dy_flags & 14; a.masks = masks; a.dy = (float*)dy; a.d_colors = d_colors; a.colors = colors; a.d_vec = d_vec; a.vec = vec; a.d_feats = d_feats;
    a.dz_rgb = dz_rgb; a.dz_vec = dz_vec; a.n_points = n_points; a.n_dev = n_dev; a.vec_stride = vec_stride;
    a.ws_first = ws_first; a.ws_points = ws_points;
    const unsigned blocks = (unsigned)((n_points + BW_PTS - 1) / BW_PTS);
    hipStream_t s = (hipStream_t)stream;
    if (dy_flags & 16) {       // single-product chain (opt-in 16-bit-native training): scaled f16 gradients, fused or vector-only
        VFN_REQUIRE((dy_flags & 8) && !(d_feats && !fused), "vfn_mlp_bwd_chain_bf16: the single-product chain takes scaled f16 gradients and no feature gradient");
        if (fused) hipLaunchKernelGGL(vfn_bwd16_kernel<BM_FUSED | BM_F16S | BM_P1>, dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(vfn_bwd16_kernel<BM_VEC | BM_F16S | BM_P1>, dim3(blocks), dim3(256), 0, s, a);
    } else if (dy_flags & 8) {
        if (fused) hipLaunchKernelGGL(vfn_bwd16_kernel<BM_FUSED | BM_F16S>, dim3(blocks), dim3(256), 0, s, a);
        else if (d_feats) hipLaunchKernelGGL(vfn_bwd16_kernel<BM_FULL | BM_F16S>, dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(vfn_bwd16_kernel<BM_VEC | BM_F16S>, dim3(blocks), dim3(256), 0, s, a);
    } else if (fused) hipLaunchKernelGGL(vfn_bwd16_kernel<BM_FUSED>, dim3(blocks), dim3(256), 0, s, a);
    else if (d_feats) hipLaunchKernelGGL(vfn_bwd16_kernel<BM_FULL>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(vfn_bwd16_kernel<BM_VEC>, dim3(blocks), dim3(256), 0, s, a);
    return vfn_check_launch("vfn_mlp_bwd_chain_bf16");
}
