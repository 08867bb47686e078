// vfn_mlp16.hip — "f16x3" forward kernels: the two MLPs on the f16 matrix cores with fp32-equivalent accuracy.
//
// Same function as vfn_mlp.hip (models/vector_field/vector_field_network.py:177-208, rendering_network.py:62-108,
// embedder.py:11-37) but 16x the matrix rate of v_mfma_f32_32x32x2_f32 is bought back with a split
// representation: every fp32 value v is carried as two halves (hi = f16(v), lo = f16(v - hi), |v - hi - lo| <=
// 2^-22 |v|) and each product a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on
// v_mfma_f32_32x32x16_f16 (products of halves are exact in fp32, accumulation is fp32), i.e. three f16 MFMAs per
// K=16 block instead of eight fp32 MFMAs: 5.3x fewer matrix-pipe cycles at ~2^-22 relative product error.
// Measured against the reference on the golden fixtures it is as close as the exact-fp32 kernel.
//
// Structure (different from the fp32 kernel — activations never touch LDS):
//  * the GEMM is transposed, Y^T[n][m] = W'[n][k] X^T[k][m]: the MFMA's A operand is the weight tile, B is the
//    activation; a wave owns 16 points (the lane's column m = lane & 15) end to end, on v_mfma_f32_16x16x32_f16.
//    The 16x16 accumulators of output tiles 2s and 2s+1 ARE, register for register, the B operand of K-block s of
//    the next layer (accumulator row 4*(lane>>4) + reg <-> fragment element order, absorbed into the weight
//    packing), so a layer's output is split to (hi, lo) halves in registers and consumed in place: no LDS round
//    trip, no cross-lane traffic, no barrier on the activation path.  16 points per wave keep the two activation
//    sets at 128 VGPRs, so 8 waves (two per SIMD) fit and one wave's epilogue / LDS waits hide under its
//    partner's MFMAs;
//  * weights (A fragments, hi and lo planes, lane-linear 1 KiB blocks) are shared by the workgroup's 8 waves
//    through a three-slot LDS ring of chunks (32 output rows x all K = 33 KiB) filled by LDS-DMA two chunks ahead,
//    tracked with counted vmcnt waits and one raw s_barrier per chunk;
//  * the weights are pre-scaled by 2^6 at pack time (exact), so the low halves stay normal f16 numbers; the
//    epilogue multiplies by 2^-6.
#include <string.h>
#include "vfn_common.h"
#include "vfn_plan.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VFN16_WSCALE 64.0f
#define VFN16_INV_WSCALE 0.015625f
#define VFN16_MAX_CHUNK_KB 42     // 2 tiles x 10 K-blocks x 2 planes + 1 bias block

// ------------------------------------------------------------------------------------------------
// plan: where each hidden entry's chunks live in the f16 pack
// ------------------------------------------------------------------------------------------------
struct Plan16 {
    int32_t n_hidden;
    int32_t feat_layer;
    int32_t multires;
    uint32_t total_kb;
    uint32_t head_off_kb;
    uint32_t off_kb[VFN_MAX_LAYERS];   // first chunk of hidden entry h (KiB from the pack base)
    uint8_t act16[VFN_MAX_LAYERS];     // K blocks of 32 taken from the activation registers
    uint8_t aux16[VFN_MAX_LAYERS];     // K blocks of 32 taken from the auxiliary (encoding) registers
    uint8_t n_tiles[VFN_MAX_LAYERS];   // chunks = pairs of 16-row output tiles
};

static int make_plan16(int kind, const vfn_net_geom* g, VfnNetPlan* p32, Plan16* p, const char* what) {
    char err[256] = {0};
    int rc = vfn_make_plan(kind, g, p32, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("%s: %s", what, err); return rc; }
    memset(p, 0, sizeof(*p));
    p->n_hidden = p32->n_hidden; p->feat_layer = p32->feat_layer; p->multires = p32->multires;
    uint32_t off = 0;
    for (int h = 0; h < p32->n_hidden; ++h) {
        const VfnLayerPlan& lp = p32->hidden[h];
        p->act16[h] = (uint8_t)(lp.nkb_act / 4);
        p->aux16[h] = lp.nkb_aux ? 2 : 0;     // 39 / 33 encoding columns -> 64
        p->n_tiles[h] = (uint8_t)lp.n_tiles;
        if (lp.nkb_act % 4) { vfn_set_error("%s: act width not a multiple of 32", what); return VFN_ERR_UNSUPPORTED; }
        p->off_kb[h] = off;
        off += lp.n_tiles * (4u * (p->act16[h] + p->aux16[h]) + 1u);
    }
    p->head_off_kb = off;
    off += 2u * 8u + 1u;                      // head: ONE 16-row tile
    p->total_kb = off;
    return VFN_OK;
}

// ------------------------------------------------------------------------------------------------
// pack: fold BatchNorm / skip scale, scale by 2^6, split to halves, fragment order
//   chunk = [tile t16 of the pair][kb][plane hi|lo][lane][8 halves] ++ bias block [t16][g][4 floats] (1 KiB)
//   element j of lane (r = lane & 15, g = lane >> 4) of K-block kb (32 wide), output row n = 32*chunk + 16*t16 + r:
//     act blocks:  k = 32*kb + 16*(j>>2) + 4*g + (j&3)           (accumulator-pair-as-operand order)
//     aux blocks:  k_aux = 32*(kb - act) + 8*g + j
//   the head is a single tile (n_tiles16 == 1).
// ------------------------------------------------------------------------------------------------
struct Pack16Entry {
    const float* w; const float* b; const float* bn_w; const float* bn_b; const float* bn_mean; const float* bn_var;
    uint32_t off_kb, n_chunks, tiles16, act16, aux16;
    int32_t in_dim, row_off, n_rows, act_col_off, act_valid, aux_col_off, aux_valid;
    float scale;
};
struct Pack16Args {
    Pack16Entry e[VFN_MAX_LAYERS + 2];
    int32_t n_entries;
    uint32_t total_words;   // 32-bit words
    uint32_t* out;
};

__device__ __forceinline__ float folded_weight(const Pack16Entry& e, int n, int col) {
    if (n >= e.n_rows || col < 0) return 0.f;
    const int row = e.row_off + n;
    float w = e.w[(size_t)row * e.in_dim + col];
    if (e.bn_w) w *= e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f);
    return w * e.scale * VFN16_WSCALE;
}

__global__ void vfn_pack16_kernel(Pack16Args a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;   // one 32-bit word
    if (idx >= a.total_words) return;
    int ei = 0;
    for (int i = 1; i < a.n_entries; ++i)
        if (idx >= a.e[i].off_kb * 256u) ei = i;
    const Pack16Entry& e = a.e[ei];
    const uint32_t nkb = e.act16 + e.aux16;
    const uint32_t wblocks = e.tiles16 * nkb * 2u;             // 1 KiB weight blocks per chunk
    const uint32_t chunk_words = (wblocks + 1u) * 256u;
    const uint32_t local = idx - e.off_kb * 256u;
    const uint32_t ck = local / chunk_words, cw = local % chunk_words;
    uint32_t word = 0;
    if (cw < wblocks * 256u) {
        const uint32_t blk = cw >> 8, lane = (cw >> 2) & 63u, jp = cw & 3u;   // word jp holds elements 2jp, 2jp+1
        const uint32_t part = blk & 1u, kb = (blk >> 1) % nkb, t16 = (blk >> 1) / nkb;
        const int g = (int)(lane >> 4);
        const int n = (int)(32u * ck + 16u * t16 + (lane & 15u));
        _Float16 halves[2];
        for (int q = 0; q < 2; ++q) {
            const int j = (int)(2u * jp) + q;
            int col = -1;
            if (kb < e.act16) {
                const int k = 32 * (int)kb + 16 * (j >> 2) + 4 * g + (j & 3);
                if (k < e.act_valid) col = e.act_col_off + k;
            } else {
                const int k = 32 * (int)(kb - e.act16) + 8 * g + j;
                if (k < e.aux_valid) col = e.aux_col_off + k;
            }
            const float w = folded_weight(e, n, col);
            const _Float16 hi = (_Float16)w;
            halves[q] = part ? (_Float16)(w - (float)hi) : hi;
        }
        word = (uint32_t)__builtin_bit_cast(unsigned short, halves[0]) |
               ((uint32_t)__builtin_bit_cast(unsigned short, halves[1]) << 16);
    } else {
        const uint32_t bi = cw - wblocks * 256u;   // bias block: [t16][g][4] floats in accumulator-row order
        if (bi < 32u) {
            const int t16 = (int)(bi >> 4), g = (int)((bi >> 2) & 3u), r = (int)(bi & 3u);
            const int n = (int)(32u * ck) + 16 * t16 + 4 * g + r;
            float b = 0.f;
            if (n < e.n_rows && (uint32_t)t16 < e.tiles16) {
                const int row = e.row_off + n;
                b = e.b[row];
                if (e.bn_w) b = (b - e.bn_mean[row]) * (e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f)) + e.bn_b[row];
            }
            word = __builtin_bit_cast(uint32_t, b * VFN16_WSCALE);
        }
    }
    a.out[idx] = word;
}

extern "C" int64_t vfn_pack16_size(int32_t net_kind, const vfn_net_geom* geom) {
    VfnNetPlan p32; Plan16 p;
    int rc = make_plan16(net_kind, geom, &p32, &p, "vfn_pack16_size");
    if (rc != VFN_OK) return rc;
    return (int64_t)p.total_kb * 1024;
}

extern "C" int vfn_pack16_weights(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                                  void* packed16, void* stream) {
    VfnNetPlan p32; Plan16 p;
    int rc = make_plan16(net_kind, geom, &p32, &p, "vfn_pack16_weights");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(layers && packed16, "vfn_pack16_weights: NULL argument");
    Pack16Args a;
    memset(&a, 0, sizeof(a));
    const int L = geom->n_layers, pe_dim = p32.pe_dim, F = geom->feature_dims;
    auto fill = [&](Pack16Entry& e, int i) -> int {
        const vfn_layer_params& q = layers[i];
        VFN_REQUIRE(q.weight && q.bias, "vfn_pack16_weights: layer %d has NULL weight/bias", i);
        e.w = q.weight; e.b = q.bias;
        if (geom->has_bn[i]) {
            VFN_REQUIRE(q.bn_weight && q.bn_bias && q.bn_mean && q.bn_var, "vfn_pack16_weights: layer %d BatchNorm pointer NULL", i);
            e.bn_w = q.bn_weight; e.bn_b = q.bn_bias; e.bn_mean = q.bn_mean; e.bn_var = q.bn_var;
        }
        e.in_dim = geom->in_dims[i]; e.scale = 1.0f;
        return VFN_OK;
    };
    for (int h = 0; h < p.n_hidden; ++h) {
        const int i = p32.hidden[h].ref_layer;
        Pack16Entry& e = a.e[a.n_entries++];
        rc = fill(e, i);
        if (rc != VFN_OK) return rc;
        e.off_kb = p.off_kb[h]; e.n_chunks = p.n_tiles[h]; e.tiles16 = 2; e.act16 = p.act16[h]; e.aux16 = p.aux16[h];
        const bool feat = p.feat_layer && h == p.n_hidden - 1;
        e.row_off = feat ? 3 : 0;
        e.n_rows = feat ? F : geom->out_dims[i];
        if (net_kind == VFN_NET_VF) {
            if (i == 0) { e.act_valid = 0; e.aux_col_off = 0; e.aux_valid = pe_dim; }
            else if (i == geom->skip_layer) {
                e.act_col_off = 0; e.act_valid = geom->out_dims[i - 1];
                e.aux_col_off = geom->out_dims[i - 1]; e.aux_valid = pe_dim;
                e.scale = 0.70710678118654752440f;
            } else { e.act_col_off = 0; e.act_valid = VFN_HIDDEN; }
        } else {
            if (i == 0) { e.act_col_off = 6 + pe_dim; e.act_valid = F; e.aux_col_off = 0; e.aux_valid = 6 + pe_dim; }
            else { e.act_col_off = 0; e.act_valid = VFN_HIDDEN; }
        }
    }
    {
        Pack16Entry& e = a.e[a.n_entries++];
        rc = fill(e, L - 1);
        if (rc != VFN_OK) return rc;
        e.off_kb = p.head_off_kb; e.n_chunks = 1; e.tiles16 = 1; e.act16 = 8; e.aux16 = 0;
        e.row_off = 0; e.n_rows = 3; e.act_col_off = 0; e.act_valid = VFN_HIDDEN;
    }
    a.total_words = p.total_kb * 256u;
    a.out = (uint32_t*)packed16;
    hipLaunchKernelGGL(vfn_pack16_kernel, dim3((a.total_words + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_pack16_weights");
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
namespace {

enum : int { EPI_RELU = 0, EPI_TANH = 1, EPI_HEAD_TANH = 2, EPI_HEAD_SIGMOID = 3 };
enum : int { M16_VF_VEC = 0, M16_FUSED = 1 };

#define VFN16_MAX_CHUNKS 160
#define VFN16_SLOT (VFN16_MAX_CHUNK_KB * 64)   // uint4 elements per LDS ring slot
#define VFN16_WAVES 8
#define VFN16_PTS 128                          // points per workgroup (16 per wave)

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct Mlp16Args {
    Plan16 vf, rn;
    const uint4* vf_w;    // f16 pack
    const uint4* rn_w;
    const float* points;
    const float* ray_dirs;
    float* out_vec;       // [M,3]
    float* out_colors;    // [M,3]
    long long n_points;
    int dirs_div;
    int n_chunks;
    // the weight chunks in the order the kernel consumes them: bit 31 = rendering net, bits 30..8 = KiB offset in that
    // net's pack, bits 7..0 = KiB size.  Copied to LDS at kernel start (dynamic indexing of a by-value argument would
    // be lowered to a private-memory copy, and scratch traffic would break the counted vmcnt waits).
    uint32_t chunk[VFN16_MAX_CHUNKS];
};

struct X16 { half8 hi[8]; half8 lo[8]; };      // 256 activation columns of this lane's point, split (8 K-blocks of 32)
struct A16 { half8 hi[2]; half8 lo[2]; };      // 64 auxiliary (encoding) columns

// Three-slot LDS ring fed by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): while chunk c feeds the
// MFMAs, chunk c+1 is landing and chunk c+2 is being issued.  Completion is tracked with COUNTED vmcnt waits and a
// raw s_barrier (a __syncthreads() would drain the DMA in flight); nothing else in the loop touches vector memory.
struct Pipe {
    uint4* lds;            // ring
    const uint32_t* tab;   // chunk list (LDS)
    const uint4* vf_w;
    const uint4* rn_w;
    int n_chunks;
    int slot;              // ring slot of the chunk being consumed
    int c;                 // index of that chunk in the chunk list
};

__device__ __forceinline__ void split2(float a, float b, _Float16& h0, _Float16& h1, _Float16& l0, _Float16& l1) {
    const float2v v = {a, b};
    const half2v hi = __builtin_convertvector(v, half2v);
    const float2v back = __builtin_convertvector(hi, float2v);
    const half2v lo = __builtin_convertvector(v - back, half2v);
    h0 = hi[0]; h1 = hi[1]; l0 = lo[0]; l1 = lo[1];
}

// tanh through one exp: 1 - 2 / (1 + e^{2v}).  Absolute error ~1e-7 (the subtraction rounds at 1 ulp of 1), which is
// what matters for values that feed the next layer in fp32-equivalent arithmetic; saturates correctly to +-1.
// The ocml tanhf expands to a long two-branch sequence whose temporaries spilled to scratch in the unrolled epilogue.
__device__ __forceinline__ float tanh_exp(float v) { return 1.0f - 2.0f / (1.0f + expf(2.0f * v)); }

__device__ __forceinline__ uint32_t chunk_entry(const Pipe& p, int idx) {
    return __builtin_amdgcn_readfirstlane(p.tab[idx]);
}
// The DMA is issued by the first VFN16_DMA_WAVES waves only, and AFTER their MFMAs: those are the older waves, which
// win matrix-pipe arbitration, finish each chunk's MFMAs first and would otherwise idle at the barrier; issuing an
// LDS-DMA piece blocks a wave for ~70 cycles, so the younger (critical) waves issue none.  Interleaved A/B on one
// GPU: -5 % kernel time vs. all eight waves issuing at the start of the chunk.
#ifndef VFN16_DMA_WAVES
#define VFN16_DMA_WAVES 4
#endif
#define EXP_DMA_WAVES VFN16_DMA_WAVES
__device__ __forceinline__ int dma_count(uint32_t entry, int wave) {   // DMA instructions this wave issues for a chunk
    const int kb = (int)(entry & 0xffu);
    if (EXP_DMA_WAVES != VFN16_WAVES) return (wave < EXP_DMA_WAVES && kb > wave) ? (kb - wave + EXP_DMA_WAVES - 1) / EXP_DMA_WAVES : 0;
    return kb > wave ? (kb - wave + VFN16_WAVES - 1) / VFN16_WAVES : 0;
}

__device__ __forceinline__ void dma_issue(const Pipe& p, int chunk_idx, int slot, int wave, int lane) {
    const uint32_t e = chunk_entry(p, chunk_idx);
    const uint4* src = ((e >> 31) ? p.rn_w : p.vf_w) + (size_t)((e >> 8) & 0x7fffffu) * 64;
    const int kb = (int)(e & 0xffu);
    uint4* dst = p.lds + slot * VFN16_SLOT;
    for (int b = wave; b < kb; b += VFN16_WAVES)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + b * 64 + lane), (lds_void*)(dst + b * 64), 16, 0, 0);
}

__device__ __forceinline__ void wait_all_but(int n) {   // s_waitcnt vmcnt(n), n wave-uniform in 0..6
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

// End of a chunk: chunk c+1 must have landed (ours: counted wait leaving chunk c+2's DMA in flight; everyone's: barrier).
__device__ __forceinline__ void pipe_next(Pipe& p, int wave) {
#ifndef ABL_NOSYNC
    if (p.c + 1 < p.n_chunks) {
#ifndef ABL_NOWAIT
        wait_all_but(p.c + 2 < p.n_chunks ? dma_count(chunk_entry(p, p.c + 2), wave) : 0);
#endif
#ifndef ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
    }
#endif
    p.c += 1;
    p.slot = p.slot == 2 ? 0 : p.slot + 1;
}

// One layer: xout <- f(W' [xin ; aux] + b') — NCH chunks of TPC 16-row output tiles each.
// D layout of v_mfma_f32_16x16x32_f16: column = lane & 15 (the point), row = 4 * (lane >> 4) + reg.
template <int ACT, int AUX, int NCH, int TPC, int EPI>
__device__ __forceinline__ void layer16(const X16& xin, const A16& aux, X16& xout, float (&head)[3], Pipe& p, int wave,
                                        int lane) {
    constexpr int NKB = ACT + AUX;
    const int g = lane >> 4;
    const bool late_dma = wave < VFN16_DMA_WAVES;   // wave-uniform; see VFN16_DMA_WAVES
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        // (1) keep the ring two chunks ahead (issued after the MFMAs, by the older waves only: see VFN16_DMA_WAVES)
#ifndef ABL_NODMA
        if (!late_dma && p.c + 2 < p.n_chunks) dma_issue(p, p.c + 2, p.slot == 0 ? 2 : p.slot - 1, wave, lane);
#endif
        // (2) the chunk's tiles as one flat sequence of TPC * NKB steps (one K-block of one tile each).  The A
        // fragments of step s+2 are read while the MFMAs of step s run; the order is pinned with
        // sched_group_barrier because hipcc otherwise sinks every ds_read to just before its use and the LDS
        // latency is paid on every step.
        const uint4* cb = p.lds + p.slot * VFN16_SLOT;
        constexpr int STEPS = TPC * NKB;
        f32x4v acc[TPC];
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = reinterpret_cast<const f32x4v*>(cb + STEPS * 2 * 64)[t * 4 + g];
        half8 fh[3], fl[3];
        fh[0] = __builtin_bit_cast(half8, cb[0 * 64 + lane]);
        fl[0] = __builtin_bit_cast(half8, cb[1 * 64 + lane]);
        if (STEPS > 1) {
            fh[1] = __builtin_bit_cast(half8, cb[2 * 64 + lane]);
            fl[1] = __builtin_bit_cast(half8, cb[3 * 64 + lane]);
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 2 < STEPS) {
                fh[(st + 2) % 3] = __builtin_bit_cast(half8, cb[(2 * (st + 2)) * 64 + lane]);
                fl[(st + 2) % 3] = __builtin_bit_cast(half8, cb[(2 * (st + 2) + 1) * 64 + lane]);
            }
            const int t = st / NKB, kb = st % NKB;
            const half8 a_hi = fh[st % 3], a_lo = fl[st % 3];
            const half8 x_hi = kb < ACT ? xin.hi[kb < ACT ? kb : 0] : aux.hi[kb >= ACT ? kb - ACT : 0];
            const half8 x_lo = kb < ACT ? xin.lo[kb < ACT ? kb : 0] : aux.lo[kb >= ACT ? kb - ACT : 0];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, x_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, x_lo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, x_hi, acc[t], 0, 0, 0);
        }
        // schedule: bias + the first two steps' reads, then per step [2 reads for step s+2 | 3 MFMAs of step s]
        __builtin_amdgcn_sched_group_barrier(0x100, TPC + (STEPS > 1 ? 4 : 2), 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 2 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        // (3) epilogue: the tile pair becomes K-block `ch` of the next layer's operand
#ifdef ABL_NOEPI
        if (EPI == EPI_RELU || EPI == EPI_TANH) {
            asm volatile("" :: "v"(acc[0]), "v"(acc[TPC - 1]));
            xout.hi[ch] = xin.hi[0]; xout.lo[ch] = xin.lo[0];
        } else
#endif
        if (EPI == EPI_RELU || EPI == EPI_TANH) {
            half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                float v0 = acc[j >> 2][j & 3] * VFN16_INV_WSCALE, v1 = acc[j >> 2][(j & 3) + 1] * VFN16_INV_WSCALE;
                // ReLU, saturated below the f16 range so that an out-of-family activation degrades instead of turning
                // into inf - inf = NaN in the split (activations of BatchNorm'ed layers are O(1..100))
                if (EPI == EPI_RELU) { v0 = fminf(fmaxf(v0, 0.f), 60000.f); v1 = fminf(fmaxf(v1, 0.f), 60000.f); }
                else { v0 = tanh_exp(v0); v1 = tanh_exp(v1); }
                _Float16 h0, h1, l0, l1;
                split2(v0, v1, h0, h1, l0, l1);
                hi[j] = h0; hi[j + 1] = h1; lo[j] = l0; lo[j + 1] = l1;
            }
            xout.hi[ch] = hi; xout.lo[ch] = lo;
        } else {
            // 3-channel head: rows 0..2 of the tile are registers 0..2 of the g == 0 lanes
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = acc[0][c] * VFN16_INV_WSCALE;
                head[c] = (EPI == EPI_HEAD_TANH) ? tanh_exp(v) : 1.0f / (1.0f + expf(-v));
            }
        }
#ifndef ABL_NODMA
        if (late_dma && p.c + 2 < p.n_chunks) dma_issue(p, p.c + 2, p.slot == 0 ? 2 : p.slot - 1, wave, lane);
#endif
        // (4) hand over to the next chunk
        pipe_next(p, wave);
    }
}

// encoding columns [x(3), sin/cos(2^k x)...] of one 3-vector: column k of the aux operand
__device__ __forceinline__ float enc_value(const float (&x)[3], const float (&sn)[18], const float (&cs)[18], int multires, int k) {
    if (k < 3) return x[k];
    const int idx = k - 3, oct = idx / 6, rem = idx - 6 * oct;
    if (oct >= multires) return 0.f;
    return rem < 3 ? sn[3 * oct + rem] : cs[3 * oct + rem - 3];
}

// aux operand from a column generator: element j of lane group g of K-block s <-> column 32 s + 8 g + j
template <typename F>
__device__ __forceinline__ void build_aux(A16& aux, int g, F col) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float v0, v1;
            if (g == 0) { v0 = col(32 * s + j); v1 = col(32 * s + j + 1); }
            else if (g == 1) { v0 = col(32 * s + 8 + j); v1 = col(32 * s + 8 + j + 1); }
            else if (g == 2) { v0 = col(32 * s + 16 + j); v1 = col(32 * s + 16 + j + 1); }
            else { v0 = col(32 * s + 24 + j); v1 = col(32 * s + 24 + j + 1); }
            _Float16 h0, h1, l0, l1;
            split2(v0, v1, h0, h1, l0, l1);
            hi[j] = h0; hi[j + 1] = h1; lo[j] = l0; lo[j + 1] = l1;
        }
        aux.hi[s] = hi; aux.lo[s] = lo;
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void vfn_mlp16_kernel(const Mlp16Args a) {
    // ONE __shared__ object: a second one beside an LDS-DMA destination makes hipcc drain vmcnt(0) before every
    // first ds_read after a DMA issue (cdna_hip_programming.md, "three .s-level traps")
    __shared__ __attribute__((aligned(16))) uint4 s_ring[3 * VFN16_SLOT + VFN16_MAX_CHUNKS / 4 + 512 * 2];
    uint32_t* s_tab = reinterpret_cast<uint32_t*>(s_ring + 3 * VFN16_SLOT);
    // per-thread parking space for the point and its view direction (8 floats): they are needed again only when
    // the rendering net's aux operand is built, ~100 chunks later, and would otherwise be spilled to scratch —
    // private-memory traffic that shows up as ~0.7 GB of HBM reads+writes per launch and perturbs the vmcnt waits
    float* s_park = reinterpret_cast<float*>(s_ring + 3 * VFN16_SLOT + VFN16_MAX_CHUNKS / 4) + threadIdx.x * 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const long long m = (long long)blockIdx.x * VFN16_PTS + wave * 16 + (lane & 15);
    const bool in = m < a.n_points;

    if (tid < VFN16_MAX_CHUNKS) s_tab[tid] = a.chunk[tid];
    // this lane's point (the four lane groups of a wave share the 16 points); loaded BEFORE any DMA so that the
    // counted vmcnt waits of the ring only ever see DMA instructions
    float x[3] = {0.f, 0.f, 0.f};
    if (in) { x[0] = a.points[m * 3 + 0]; x[1] = a.points[m * 3 + 1]; x[2] = a.points[m * 3 + 2]; }
    float d[3] = {0.f, 0.f, 0.f};
    if (MODE == M16_FUSED && in) {
        const long long di = m / a.dirs_div;
        d[0] = a.ray_dirs[di * 3 + 0]; d[1] = a.ray_dirs[di * 3 + 1]; d[2] = a.ray_dirs[di * 3 + 2];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == M16_FUSED) {
        s_park[0] = x[0]; s_park[1] = x[1]; s_park[2] = x[2];
        s_park[4] = d[0]; s_park[5] = d[1]; s_park[6] = d[2];
    }
    __syncthreads();   // chunk list visible

    Pipe p;
    p.lds = s_ring; p.tab = s_tab; p.vf_w = a.vf_w; p.rn_w = a.rn_w; p.n_chunks = a.n_chunks; p.slot = 0; p.c = 0;
    dma_issue(p, 0, 0, wave, lane);
    if (p.n_chunks > 1) dma_issue(p, 1, 1, wave, lane);

    // ---- positional encoding of the point -> aux operand ------------------------------------------------
    const int vf_multires = a.vf.multires, rn_multires = a.rn.multires;
    A16 aux;
    {
        float sn[18], cs[18];
#pragma unroll
        for (int o = 0; o < 6; ++o)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (o < vf_multires) sincosf(x[c] * (float)(1 << o), &sn[3 * o + c], &cs[3 * o + c]);
                else { sn[3 * o + c] = 0.f; cs[3 * o + c] = 0.f; }
            }
        build_aux(aux, g, [&](int k) { return enc_value(x, sn, cs, vf_multires, k); });
    }
    // chunk 0 landed (chunk 1 may still be in flight)
    wait_all_but(p.n_chunks > 1 ? dma_count(chunk_entry(p, 1), wave) : 0);
    __builtin_amdgcn_s_barrier();

    // ---- VF net (shipped family: L0 | plain... | narrow | skip | plain... | [features] + head) ------------
    X16 xa, xb;
    float vec[3] = {0.f, 0.f, 0.f};
    const int n_plain = a.vf.n_hidden - a.vf.feat_layer;
    layer16<0, 2, 8, 2, EPI_RELU>(xa, aux, xb, vec, p, wave, lane);   // layer 0: encoding only
    bool in_b = true;                                                 // current activations live in xb
    for (int hh = 1; hh < n_plain; ++hh) {
        const int shape = (a.vf.aux16[hh] ? 2 : 0) | (a.vf.n_tiles[hh] == 7 ? 1 : 0);
        if (in_b) {
            if (shape == 0) layer16<8, 0, 8, 2, EPI_RELU>(xb, aux, xa, vec, p, wave, lane);
            else if (shape == 1) layer16<8, 0, 7, 2, EPI_RELU>(xb, aux, xa, vec, p, wave, lane);
            else layer16<7, 2, 8, 2, EPI_RELU>(xb, aux, xa, vec, p, wave, lane);
        } else {
            if (shape == 0) layer16<8, 0, 8, 2, EPI_RELU>(xa, aux, xb, vec, p, wave, lane);
            else if (shape == 1) layer16<8, 0, 7, 2, EPI_RELU>(xa, aux, xb, vec, p, wave, lane);
            else layer16<7, 2, 8, 2, EPI_RELU>(xa, aux, xb, vec, p, wave, lane);
        }
        in_b = !in_b;
    }
    if (MODE == M16_VF_VEC) {
        if (in_b) layer16<8, 0, 1, 1, EPI_HEAD_TANH>(xb, aux, xa, vec, p, wave, lane);
        else layer16<8, 0, 1, 1, EPI_HEAD_TANH>(xa, aux, xb, vec, p, wave, lane);
        if (in && g == 0) { a.out_vec[m * 3 + 0] = vec[0]; a.out_vec[m * 3 + 1] = vec[1]; a.out_vec[m * 3 + 2] = vec[2]; }
        return;
    }
    // fused: feature block (tanh) then the vector head, both from the same input
    if (in_b) {
        layer16<8, 0, 8, 2, EPI_TANH>(xb, aux, xa, vec, p, wave, lane);
        layer16<8, 0, 1, 1, EPI_HEAD_TANH>(xb, aux, xb, vec, p, wave, lane);
    } else {
        layer16<8, 0, 8, 2, EPI_TANH>(xa, aux, xb, vec, p, wave, lane);
        layer16<8, 0, 1, 1, EPI_HEAD_TANH>(xa, aux, xa, vec, p, wave, lane);
    }
    in_b = !in_b;   // the features are in the other set now
    // the head's outputs sit in the g == 0 lanes; the other lane groups of the same point need them for the aux operand
    float nrm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = __shfl(vec[c], lane & 15, 64);

    // ---- rendering net: aux = [p(3), d(3), sin/cos(2^k d)(6L), n(3)] -------------------------------------
    {
        float xr[3] = {s_park[0], s_park[1], s_park[2]};
        float dr[3] = {s_park[4], s_park[5], s_park[6]};
        float sn[18], cs[18];
#pragma unroll
        for (int o = 0; o < 6; ++o)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (o < rn_multires) sincosf(dr[c] * (float)(1 << o), &sn[3 * o + c], &cs[3 * o + c]);
                else { sn[3 * o + c] = 0.f; cs[3 * o + c] = 0.f; }
            }
        const int ncol = 6 + 6 * rn_multires;   // first normal column
        build_aux(aux, g, [&](int k) -> float {
            if (k < 3) return xr[k];
            if (k >= ncol) return k < ncol + 3 ? nrm[k - ncol] : 0.f;
            return enc_value(dr, sn, cs, rn_multires, k - 3);
        });
    }
    float rgb[3] = {0.f, 0.f, 0.f};
    const int rn_hidden = a.rn.n_hidden;
    for (int hh = 0; hh < rn_hidden; ++hh) {
        if (in_b) {
            if (hh == 0) layer16<8, 2, 8, 2, EPI_RELU>(xb, aux, xa, rgb, p, wave, lane);
            else layer16<8, 0, 8, 2, EPI_RELU>(xb, aux, xa, rgb, p, wave, lane);
        } else {
            if (hh == 0) layer16<8, 2, 8, 2, EPI_RELU>(xa, aux, xb, rgb, p, wave, lane);
            else layer16<8, 0, 8, 2, EPI_RELU>(xa, aux, xb, rgb, p, wave, lane);
        }
        in_b = !in_b;
    }
    if (in_b) layer16<8, 0, 1, 1, EPI_HEAD_SIGMOID>(xb, aux, xa, rgb, p, wave, lane);
    else layer16<8, 0, 1, 1, EPI_HEAD_SIGMOID>(xa, aux, xb, rgb, p, wave, lane);
    // outputs last: the only vector-memory stores of the kernel come after the last counted wait
    if (in && g == 0) {
        a.out_vec[m * 3 + 0] = nrm[0]; a.out_vec[m * 3 + 1] = nrm[1]; a.out_vec[m * 3 + 2] = nrm[2];
        a.out_colors[m * 3 + 0] = rgb[0]; a.out_colors[m * 3 + 1] = rgb[1]; a.out_colors[m * 3 + 2] = rgb[2];
    }
}

// chunk list in consumption order
int push_chunks(Mlp16Args& a, const Plan16& pl, int net, int h) {
    const uint32_t kb = 4u * (pl.act16[h] + pl.aux16[h]) + 1u;
    for (uint32_t ck = 0; ck < pl.n_tiles[h]; ++ck) {
        if (a.n_chunks >= VFN16_MAX_CHUNKS) return VFN_ERR_UNSUPPORTED;
        a.chunk[a.n_chunks++] = ((uint32_t)net << 31) | ((pl.off_kb[h] + ck * kb) << 8) | kb;
    }
    return VFN_OK;
}
int push_head(Mlp16Args& a, const Plan16& pl, int net) {
    if (a.n_chunks >= VFN16_MAX_CHUNKS) return VFN_ERR_UNSUPPORTED;
    a.chunk[a.n_chunks++] = ((uint32_t)net << 31) | (pl.head_off_kb << 8) | 17u;
    return VFN_OK;
}

int check_family(const Plan16& vf, const char* what) {
    // the f16x3 kernel is specialised for the shipped layer shapes: first layer encoding-only, optional narrow layer
    // (7 chunks) before a skip layer (7 + 2 K-blocks), everything else 256 x 256
    if (vf.act16[0] != 0 || vf.aux16[0] != 2 || vf.n_tiles[0] != 8) { vfn_set_error("%s: unsupported first layer", what); return VFN_ERR_UNSUPPORTED; }
    const int n_plain = vf.n_hidden - vf.feat_layer;
    for (int h = 1; h < n_plain; ++h) {
        const bool skip = vf.aux16[h] != 0;
        if ((skip && vf.act16[h] != 7) || (!skip && vf.act16[h] != 8) || (vf.n_tiles[h] != 8 && vf.n_tiles[h] != 7) ||
            (skip && vf.n_tiles[h] != 8)) {
            vfn_set_error("%s: hidden layer %d has a shape the f16x3 kernel is not specialised for", what, h);
            return VFN_ERR_UNSUPPORTED;
        }
    }
    return VFN_OK;
}

}  // namespace

extern "C" int vfn_vf_mlp16_fwd(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                                float* out_vec, void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32;
    int rc = make_plan16(VFN_NET_VF, geom, &p32, &a.vf, "vfn_vf_mlp16_fwd");
    if (rc != VFN_OK) return rc;
    rc = check_family(a.vf, "vfn_vf_mlp16_fwd");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(packed16 && points && out_vec, "vfn_vf_mlp16_fwd: NULL argument");
    a.vf_w = (const uint4*)packed16; a.points = points; a.out_vec = out_vec; a.n_points = n_points; a.dirs_div = 1;
    const int n_plain = a.vf.n_hidden - a.vf.feat_layer;
    for (int h = 0; h < n_plain; ++h)
        if ((rc = push_chunks(a, a.vf, 0, h)) != VFN_OK) { vfn_set_error("vfn_vf_mlp16_fwd: too many weight chunks"); return rc; }
    if ((rc = push_head(a, a.vf, 0)) != VFN_OK) { vfn_set_error("vfn_vf_mlp16_fwd: too many weight chunks"); return rc; }
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_VEC>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_mlp16_fwd");
}

extern "C" int vfn_vf_render_fused16_fwd(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                         const void* rn_packed16, const float* points, const float* ray_dirs,
                                         int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                         void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32;
    int rc = make_plan16(VFN_NET_VF, vf_geom, &p32, &a.vf, "vfn_vf_render_fused16_fwd");
    if (rc != VFN_OK) return rc;
    rc = check_family(a.vf, "vfn_vf_render_fused16_fwd");
    if (rc != VFN_OK) return rc;
    rc = make_plan16(VFN_NET_RENDER, rn_geom, &p32, &a.rn, "vfn_vf_render_fused16_fwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(a.vf.feat_layer && vf_geom->feature_dims == VFN_HIDDEN && rn_geom->feature_dims == VFN_HIDDEN,
                "vfn_vf_render_fused16_fwd: both nets need feature_dims == %d", VFN_HIDDEN);
    VFN_REQUIRE(a.rn.act16[0] == 8 && a.rn.aux16[0] == 2, "vfn_vf_render_fused16_fwd: unsupported rendering layer 0");
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed16 && rn_packed16 && points && ray_dirs && normals && colors, "vfn_vf_render_fused16_fwd: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_vf_render_fused16_fwd: samples_per_ray must be > 0");
    a.vf_w = (const uint4*)vf_packed16; a.rn_w = (const uint4*)rn_packed16; a.points = points; a.ray_dirs = ray_dirs;
    a.out_vec = normals; a.out_colors = colors; a.n_points = n_points; a.dirs_div = samples_per_ray;
    for (int h = 0; h < a.vf.n_hidden && rc == VFN_OK; ++h) rc = push_chunks(a, a.vf, 0, h);
    if (rc == VFN_OK) rc = push_head(a, a.vf, 0);
    for (int h = 0; h < a.rn.n_hidden && rc == VFN_OK; ++h) rc = push_chunks(a, a.rn, 1, h);
    if (rc == VFN_OK) rc = push_head(a, a.rn, 1);
    if (rc != VFN_OK) { vfn_set_error("vfn_vf_render_fused16_fwd: too many weight chunks"); return rc; }
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_render_fused16_fwd");
}
