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
//  * the GEMM is transposed, Y^T[n][m] = W'[n][k] X^T[k][m]: the MFMA's A operand is a 32-row weight tile, B is the
//    activation; a wave owns 32 points (the lane's column m = lane & 31) end to end.  The 32x32 accumulator of output
//    tile t IS, register for register, the B operand of K-blocks 2t and 2t+1 of the next layer (accumulator row
//    (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) <-> fragment element order, absorbed into the weight packing), so a
//    layer's output is split to (hi, lo) halves in registers and consumed in place: no LDS round trip, no cross-lane
//    traffic on the activation path;
//  * ONE wave per SIMD, 4 waves = 128 points per workgroup, the whole 512-entry register file per wave: the two
//    activation sets (2 x 128 registers) live in the AGPRs, where the MFMA reads its B operand directly; accumulators,
//    weight fragments and the epilogue use the arch VGPRs (hipcc -mllvm -amdgpu-mfma-vgpr-form).  32 points per wave
//    halve the LDS fragment traffic per FLOP of a 16-point tiling, and v_mfma_f32_32x32x16 leaves 24 of its 32 cycles
//    free for other issue (8 of 16 for the 16x16x32 form);
//  * weights (A fragments, hi and lo planes, lane-linear 1 KiB blocks) are shared by the workgroup's 4 waves through
//    a three-slot LDS ring of chunks (one 32-row tile x all K, <= 39 KiB) filled by LDS-DMA (buffer_load ... lds);
//  * with one wave per SIMD nothing else hides the epilogue, the DMA issue or the ring hand-over, so the chunk loop is
//    software-pipelined by hand over the K steps of a tile (see layer16): epilogue of the previous tile in the first
//    half, vmcnt(0) + s_barrier in the middle, DMA pieces of chunk c+2 in the second half, bias and first fragments of
//    chunk c+1 in the last step;
//  * the layer sequence is the shipped one (confs/vf_nerf.conf:13-37) and compiled in: straight-line code, every chunk's
//    place in the pack, its size and its ring slot are immediates; other geometries return VFN_ERR_UNSUPPORTED and
//    callers use the fp32 kernels;
//  * accumulators hold 2^6 x the true pre-activation and the factor rides on the activations (operand blocks hold
//    2^6 x, weights are packed unscaled, the ReLU epilogue needs no multiply).  f16 denormals are honoured by the
//    MFMA, which the low halves of the weights rely on.
//
// Build flags (build.sh): -mllvm -amdgpu-mfma-vgpr-form (accumulators in arch VGPRs, leaving all 256 AGPRs to the
// activation sets) and -mllvm -pragma-unroll-threshold=10000000 (the K loops must unroll fully: operand blocks are
// register arrays).
#include <string.h>
#include <utility>
#include "vfn_common.h"
#include "vfn_plan.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Scaling: accumulators hold 2^6 x the true pre-activation.  VFN16_ASCALE = 1 carries the factor on the ACTIVATIONS
// (operand blocks hold 2^6 x, weights are packed unscaled, the ReLU epilogue needs no multiply); 0 carries it on the
// weights (activations unscaled, one multiply per value in the epilogue).
#ifndef VFN16_ASCALE
#define VFN16_ASCALE 1
#endif
#define VFN16_WSCALE 64.0f
#define VFN16_INV_WSCALE 0.015625f
#define VFN16_PACK_WSCALE (VFN16_ASCALE ? 1.0f : VFN16_WSCALE)
#define VFN16_XSCALE (VFN16_ASCALE ? VFN16_WSCALE : 1.0f)
#ifndef VFN16_SAVE_AUX
#define VFN16_SAVE_AUX 2         // cache policy bits of the training-mode workspace stores: 2 = nt (streaming; whole-line stores of the
                                 // fragment-ordered workspace: -0.1 ms on the step; with the row-major 32-byte pieces of round 1 nt cost
                                 // +33 %), 0 = default write-back
#endif
#ifndef VFN16_HANDOVER_NUM
#define VFN16_HANDOVER_NUM 8     // ring hand-over after K step NKB * n / 16
#endif
#ifndef VFN16_DMA_STEPS
#define VFN16_DMA_STEPS 16       // the DMA pieces of chunk c+2 are spread over at most this many K steps after it
#endif
#ifndef VFN16_EPI_PER_MFMA
#define VFN16_EPI_PER_MFMA 6     // VALU instructions of the pending epilogue scheduled behind each MFMA
#endif
#ifndef VFN16_EPI_PER_MFMA2
#define VFN16_EPI_PER_MFMA2 9    // the same for the two-product tiles of the colour branch (two MFMAs per K step)
#endif
#ifndef VFN16_LATE_STORES
#define VFN16_LATE_STORES 1      // training modes: the stores of a finished tile are the LAST vector-memory instructions of a chunk (after
                                 // its DMA pieces), so the next ring hand-over waits with vmcnt(4) — operations retire in issue order,
                                 // the four youngest are those stores — instead of draining them
#endif
#ifndef VFN16_MASK_STEP
#define VFN16_MASK_STEP(H, NKB) (H)     // K step of a tile that carries the sign-bit collection of the pending tile (training)
#endif
#define VFN16_MAX_CHUNK_KB 39     // (16 act + 3 aux) K-blocks x 2 planes + 1 bias block
#define VFN16_STATS_WORDS 64      // tail of a pack (256 bytes): per-entry weight statistics, see Pack16Args::stats

struct Plan16 {
    int32_t n_hidden;
    int32_t feat_layer;
    int32_t multires;
    uint32_t total_kb;
    uint32_t head_off_kb;
    uint32_t off_kb[VFN_MAX_LAYERS];
    uint8_t act16[VFN_MAX_LAYERS];     // K blocks of 16 taken from the activation registers
    uint8_t aux16[VFN_MAX_LAYERS];     // K blocks of 16 taken from the auxiliary (encoding) registers
    uint8_t n_tiles[VFN_MAX_LAYERS];   // chunks = 32-row output tiles
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
        if (lp.nkb_act % 2) { vfn_set_error("%s: act width not a multiple of 16", what); return VFN_ERR_UNSUPPORTED; }
        p->act16[h] = (uint8_t)(lp.nkb_act / 2);
        p->aux16[h] = lp.nkb_aux ? 3 : 0;     // 39 / 33 encoding columns -> 48
        p->n_tiles[h] = (uint8_t)lp.n_tiles;
        p->off_kb[h] = off;
        off += lp.n_tiles * (2u * (p->act16[h] + p->aux16[h]) + 1u);
    }
    p->head_off_kb = off;
    off += 2u * 16u + 1u;
    p->total_kb = off;
    return VFN_OK;
}

// chunk = [kb][plane hi|lo][lane][8 halves] ++ bias block [hl][16 floats] (1 KiB)
//   lane (i = lane & 31, g = lane >> 5), element j of K-block kb (16 wide), output row n = 32*chunk + i:
//     act blocks:  k = 16*kb + 8*(j>>2) + 4*g + (j&3)          (accumulator-as-operand order)
//     aux blocks:  k_aux = 16*(kb - act) + 8*g + j
struct Pack16Entry {
    const float* w; const float* b; const float* bn_w; const float* bn_b; const float* bn_mean; const float* bn_var;
    uint32_t off_kb, n_chunks, act16, aux16;
    int32_t in_dim, row_off, n_rows, act_col_off, act_valid, aux_col_off, aux_valid;
    float scale;
};
struct Pack16Args {
    Pack16Entry e[VFN_MAX_LAYERS + 2];
    int32_t n_entries;
    uint32_t total_words;
    uint32_t* out;
    uint32_t* stats;     // [VFN16_STATS_WORDS] after the pack: word e = bits of max |folded weight| of pack entry e (layers in plan
                         // order, then the head) — the host checks them against the range the (hi, lo) f16 halves represent well
};

__device__ __forceinline__ float folded_weight(const Pack16Entry& e, int n, int col) {
    if (n >= e.n_rows || col < 0) return 0.f;
    const int row = e.row_off + n;
    float w = e.w[(size_t)row * e.in_dim + col];
    if (e.bn_w) w *= e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f);
    return w * e.scale * VFN16_PACK_WSCALE;
}

// Range statistics: max |folded weight| of every pack entry (non-negative floats order like their bit patterns).  VFN16_STAT_WGS extra
// workgroups per entry at the end of the grid sweep its weights coalesced (workgroup s takes rows s, s + 32, ...) and each ends with
// ONE atomicMax.  (The pack blocks used to do it — a wave-level maximum and an atomicMax per wave, 4 368 operations on ten words:
// operations on one address serialise at the memory side, ~10 ns each, and were 43 of this kernel's 53 us for the vector-field net's
// 2.2 MB; 13 us now.  It runs twice per optimizer step.)
#define VFN16_STAT_WGS 32
__device__ __forceinline__ void pack16_entry_stats(const Pack16Args& a, int ei, int share) {
    __shared__ float s_max[4];
    const Pack16Entry& e = a.e[ei];
    const int ncol = e.act_valid + e.aux_valid;
    float m = 0.f;
    for (int n = share; n < e.n_rows; n += VFN16_STAT_WGS)
        for (int c = (int)threadIdx.x; c < ncol; c += (int)blockDim.x)
            m = fmaxf(m, fabsf(folded_weight(e, n, c < e.act_valid ? e.act_col_off + c : e.aux_col_off + (c - e.act_valid))));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float wm = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (wm > 0.f) atomicMax(a.stats + ei, __builtin_bit_cast(uint32_t, wm));
    }
}

__global__ __launch_bounds__(256) void vfn_pack16_kernel(Pack16Args a) {
    if (blockIdx.x * blockDim.x >= a.total_words) {      // the statistics workgroups behind the pack's blocks
        const int sidx = (int)(blockIdx.x - a.total_words / blockDim.x);
        if (a.stats) pack16_entry_stats(a, sidx / VFN16_STAT_WGS, sidx % VFN16_STAT_WGS);
        return;
    }
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    int ei = 0;
    for (int i = 1; i < a.n_entries; ++i)
        if (idx >= a.e[i].off_kb * 256u) ei = i;
    const Pack16Entry& e = a.e[ei];
    const uint32_t nkb = e.act16 + e.aux16;
    const uint32_t wblocks = nkb * 2u;
    const uint32_t chunk_words = (wblocks + 1u) * 256u;
    const uint32_t local = idx - e.off_kb * 256u;
    const uint32_t ck = local / chunk_words, cw = local % chunk_words;
    uint32_t word = 0;
    if (cw < wblocks * 256u) {
        const uint32_t blk = cw >> 8, lane = (cw >> 2) & 63u, jp = cw & 3u;
        const uint32_t part = blk & 1u, kb = blk >> 1;
        const int g = (int)(lane >> 5);
        const int n = (int)(32u * ck + (lane & 31u));
        _Float16 halves[2];
        for (int q = 0; q < 2; ++q) {
            const int j = (int)(2u * jp) + q;
            int col = -1;
            if (kb < e.act16) {
                const int k = 16 * (int)kb + 8 * (j >> 2) + 4 * g + (j & 3);
                if (k < e.act_valid) col = e.act_col_off + k;
            } else {
                const int k = 16 * (int)(kb - e.act16) + 8 * g + j;
                if (k < e.aux_valid) col = e.aux_col_off + k;
            }
            const float w = folded_weight(e, n, col);
            const _Float16 hi = (_Float16)w;
            halves[q] = part ? (_Float16)(w - (float)hi) : hi;
        }
        word = (uint32_t)__builtin_bit_cast(unsigned short, halves[0]) |
               ((uint32_t)__builtin_bit_cast(unsigned short, halves[1]) << 16);
    } else {
        const uint32_t bi = cw - wblocks * 256u;   // bias block: [hl][16] floats in accumulator-register order
        if (bi < 32u) {
            const int hl = (int)(bi >> 4), r = (int)(bi & 15u);
            const int n = (int)(32u * ck) + (r & 3) + 8 * (r >> 2) + 4 * hl;
            float b = 0.f;
            if (n < e.n_rows) {
                const int row = e.row_off + n;
                b = e.b[row];
                if (e.bn_w) b = (b - e.bn_mean[row]) * (e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f)) + e.bn_b[row];
            }
            word = __builtin_bit_cast(uint32_t, b * VFN16_WSCALE);
        }
    }
    a.out[idx] = word;
}

static int check_kind16(int net_kind, const Plan16& p, const char* what);   // plan against the compiled-in layer tables

extern "C" int64_t vfn_pack16_size(int32_t net_kind, const vfn_net_geom* geom) {
    VfnNetPlan p32; Plan16 p;
    int rc = make_plan16(net_kind, geom, &p32, &p, "vfn_pack16_size");
    if (rc != VFN_OK) return rc;
    rc = check_kind16(net_kind, p, "vfn_pack16_size");
    if (rc != VFN_OK) return rc;
    return (int64_t)p.total_kb * 1024 + VFN16_STATS_WORDS * 4;
}

extern "C" int vfn_pack16_weights(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                                   void* packed16, void* stream) {
    VfnNetPlan p32; Plan16 p;
    int rc = make_plan16(net_kind, geom, &p32, &p, "vfn_pack16_weights");
    if (rc != VFN_OK) return rc;
    rc = check_kind16(net_kind, p, "vfn_pack16_weights");
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
        e.off_kb = p.off_kb[h]; e.n_chunks = p.n_tiles[h]; e.act16 = p.act16[h]; e.aux16 = p.aux16[h];
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
        e.off_kb = p.head_off_kb; e.n_chunks = 1; e.act16 = 16; e.aux16 = 0;
        e.row_off = 0; e.n_rows = 3; e.act_col_off = 0; e.act_valid = VFN_HIDDEN;
    }
    a.total_words = p.total_kb * 256u;
    a.out = (uint32_t*)packed16;
    a.stats = a.out + a.total_words;
    if (hipMemsetAsync(a.stats, 0, VFN16_STATS_WORDS * 4, (hipStream_t)stream) != hipSuccess) {
        vfn_set_error("vfn_pack16_weights: hipMemsetAsync failed");
        return VFN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(vfn_pack16_kernel, dim3(a.total_words / 256 + VFN16_STAT_WGS * a.n_entries), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_pack16_weights");
}

namespace {

enum : int { EPI_RELU = 0, EPI_TANH = 1, EPI_HEAD_TANH = 2, EPI_HEAD_SIGMOID = 3 };
// launch modes: bit 0 = the VF feature block is evaluated, bit 1 = the rendering net follows, bit 2 = training (every
// hidden layer's post-activation output and the encoding tiles are written out for the backward kernels)
//   bit 3 = the feature block leaves the kernel as operand blocks (for a later rendering-net launch), bit 4 = rendering net
//   only, its feature operand gathered from such blocks
//   bit 5 = the COLOUR BRANCH (feature block of the VF net + the whole rendering net) runs on TWO products per fp32-equivalent
//   product, w_hi x_hi + w_hi x_lo: its weights are used as their f16 roundings (11 significant bits), its activations stay
//   split.  The vector head and everything before it keep three products, so normals, density, weights and depth are
//   unchanged to the bit; colours move by <= 2e-5 on the reference's golden outputs (profiles/r02/ab_colour_products.txt),
//   inside the 1e-4 contract.  The same pack serves both: a two-product chunk DMAs and reads only its hi planes.
//   bit 6 = ONE product per K-block everywhere, w_hi x_hi: plain f16 arithmetic with fp32 accumulation (11 significant bits per
//   operand) — the opt-in 16-bit-native TRAINING forward (BASELINE.json configs[2], "bf16 MFMA MLPs" as written; f16 has three
//   more bits at the same matrix rate and the range guard covers its exponent range).  Same pack: only the hi planes and the
//   bias block of a chunk are fetched.  Outside the 1e-4 contract by construction; never the default.
// M16_SF16 (training modes): the ReLU slots are f16 in FRAGMENT ORDER — the default storage — as a compile-time fact.  The stores of a tile
// then are four straight-line instructions with constant scalar offsets; with the storage form read from the Pipe at run time every register
// quad of every tile went through two or three uniform branches (f16? fragment order?): ~1 500 branches in the vector-only kernel, four
// basic-block boundaries per K step that no instruction could be scheduled across (round 5; the saving forward's "cost of the stores").
enum : int { M16_FEAT = 1, M16_RENDER = 2, M16_TRAIN = 4, M16_BLKOUT = 8, M16_BLKIN = 16, M16_C2 = 32, M16_P1 = 64, M16_SF16 = 128,
             M16_VF_VEC = 0, M16_FUSED = M16_FEAT | M16_RENDER, M16_VF_BLK = M16_FEAT | M16_BLKOUT, M16_RN_BLK = M16_RENDER | M16_BLKIN,
             M16_VF_VEC_TRAIN = M16_TRAIN, M16_VF_FULL_TRAIN = M16_FEAT | M16_TRAIN, M16_FUSED_TRAIN = M16_FUSED | M16_TRAIN };

#ifndef VFN16_FDEPTH
#define VFN16_FDEPTH 2            // A-fragment ring: K steps in registers (1 ahead)
#endif
#define VFN16_SLOT (VFN16_MAX_CHUNK_KB * 64)   // uint4 elements per LDS ring slot
#define VFN16_WAVES 4
#define VFN16_PTS 128                          // points per workgroup (32 per wave)

typedef __attribute__((address_space(3))) void lds_void;

// ------------------------------------------------------------------------------------------------
// The shipped layer sequence (confs/vf_nerf.conf:13-37) as compile-time tables: the kernel is straight-line code, every
// chunk's place in the pack, its size and its ring slot are immediates.  check_vf16() / check_rn16() verify a network against
// these tables; other geometries run on the fp32 kernels.
// ------------------------------------------------------------------------------------------------
constexpr int VF_ACT[9] = {0, 16, 16, 16, 14, 16, 16, 16, 16};    // K-blocks of 16 from the activations
constexpr int VF_AUX[9] = {3, 0, 0, 0, 3, 0, 0, 0, 0};            // K-blocks of 16 from the encoding
constexpr int VF_TILES[9] = {8, 8, 8, 7, 8, 8, 8, 8, 8};          // [8] = the feature block of the last Linear
constexpr int RN_ACT[4] = {16, 16, 16, 16};
constexpr int RN_AUX[4] = {3, 0, 0, 0};
constexpr int RN_TILES[4] = {8, 8, 8, 8};
constexpr int HEAD_KB = 2 * 16 + 1;
constexpr int chunk_kb(int act, int aux) { return 2 * (act + aux) + 1; }
constexpr int vf_off_kb(int h) { int o = 0; for (int i = 0; i < h; ++i) o += VF_TILES[i] * chunk_kb(VF_ACT[i], VF_AUX[i]); return o; }
constexpr int rn_off_kb(int h) { int o = 0; for (int i = 0; i < h; ++i) o += RN_TILES[i] * chunk_kb(RN_ACT[i], RN_AUX[i]); return o; }

struct ChunkD { int net, off_kb, kb, aux; };      // net 0 = VF pack, 1 = rendering pack; kb = 0: past the end; aux = encoding K-blocks
// chunk c of the launch in consumption order (fused: VF hidden + features, VF head, rendering hidden, rendering head;
// vector-only: the 8 plain VF layers, VF head)
constexpr ChunkD chunk_of(int mode, int c) {
    if (!(mode & M16_BLKIN)) {
        const int vf_layers = (mode & M16_FEAT) ? 9 : 8;
        for (int h = 0; h < vf_layers; ++h) {
            if (c < VF_TILES[h]) return {0, vf_off_kb(h) + c * chunk_kb(VF_ACT[h], VF_AUX[h]), chunk_kb(VF_ACT[h], VF_AUX[h]), VF_AUX[h]};
            c -= VF_TILES[h];
        }
        if (c == 0) return {0, vf_off_kb(9), HEAD_KB, 0};
        c -= 1;
    }
    if (mode & M16_RENDER) {
        for (int h = 0; h < 4; ++h) {
            if (c < RN_TILES[h]) return {1, rn_off_kb(h) + c * chunk_kb(RN_ACT[h], RN_AUX[h]), chunk_kb(RN_ACT[h], RN_AUX[h]), RN_AUX[h]};
            c -= RN_TILES[h];
        }
        if (c == 0) return {1, rn_off_kb(4), HEAD_KB, 0};
    }
    return {0, 0, 0, 0};
}

constexpr int rn_first_chunk(int mode) { return (mode & M16_BLKIN) ? 0 : 72; }
// chunks of the colour branch in a fused launch: the feature block (63..70) and the rendering net (72..104); 71 is the vector head
// (the ENCODING K-blocks of such a chunk — the rendering net's first layer reads the point, PE(view direction) and the normal
// there — keep three products: a weight's rounding error is multiplied by its input, and a point coordinate is not bounded)
constexpr bool chunk_two(int mode, int c) { return (mode & M16_C2) && !(mode & M16_BLKIN) && c >= 63 && c != 71; }
constexpr bool mode_one(int mode) { return (mode & M16_P1) != 0; }
// The weight ring.  Split modes: three slots of one whole chunk (<= 39 KiB) each, a chunk's pieces requested two chunks ahead.  The
// single-product modes fetch the hi planes and the bias block only and store them COMPACTLY (K-block s at block s, bias behind them:
// <= 20 KiB), so five slots fit where three did and a chunk is requested FOUR ahead: with a third of the matrix work per chunk the
// stores of a finished tile need more than the chunk and a half a three-slot ring gives them to retire (the hand-over's vmcnt leaves
// the stores of the last three chunks in flight instead of one's; 942 -> see DESIGN.md section 3, Backward 5).
#ifndef VFN16_P1_RING
#define VFN16_P1_RING 5
#endif
#define VFN16_P1_SLOT_KB 20
constexpr int ring_of(int mode) { return mode_one(mode) ? VFN16_P1_RING : 3; }
constexpr int slot_u4(int mode) { return mode_one(mode) ? VFN16_P1_SLOT_KB * 64 : VFN16_MAX_CHUNK_KB * 64; }      // uint4 elements per slot
constexpr int slot_base(int mode, int c) { return (c % ring_of(mode)) * slot_u4(mode); }
static_assert(VFN16_P1_RING * VFN16_P1_SLOT_KB <= 3 * VFN16_MAX_CHUNK_KB, "the single-product ring must fit into the split modes' ring");
// single-product training launches: vector-memory operations a wave has issued AFTER its pieces of chunk c+1 when it reaches the
// hand-over of chunk c.  Chunk k issues, in this order, the pieces of chunk k+RING-1 and then the four stores of its pending tile
// (none in the launch's first chunk and in the rendering net's first: no pending tile there); the first RING-1 chunks are requested
// in the prologue.  Pieces per wave: at least floor(pieces / waves).
constexpr int stores_of(int mode, int c) { return (c == 0 || ((mode & M16_RENDER) && c == rn_first_chunk(mode))) ? 0 : 4; }
constexpr int young_ops(int mode, int c) {
    const int ring = ring_of(mode);
    int n = 0;
    for (int k = (c - ring + 2 > 0 ? c - ring + 2 : 0); k < c; ++k) n += stores_of(mode, k);
    for (int j = 2; j <= ring - 2; ++j)
        if (c + j >= ring - 1 && chunk_of(mode, c + j).kb > 0) n += (((chunk_of(mode, c + j).kb - 1) / 2 + 1) / VFN16_WAVES);
    return n;
}

struct Mlp16Args {
    const uint4* vf_w;
    const uint4* rn_w;
    const float* points;
    const float* ray_dirs;
    float* out_vec;
    float* out_colors;
    long long n_points;
    int dirs_div;
    int vf_multires, rn_multires;
    uint32_t vf_bytes, rn_bytes;
    // training modes: saved[slot][M][256] (slots: VF hidden 0..8 incl. the feature block, rendering hidden 9..12),
    // save_aux_vf / save_aux_rn [M][40]
    float* saved;
    float* save_aux_vf;
    float* save_aux_rn;
    // training modes: ReLU sign bits of every saved activation, [13][M][2][4] u32 (see save_mask)
    uint32_t* save_masks;
    int save_f16;             // training modes, flags: bit 0 = store the ReLU slots as f16; bit 1 = FRAGMENT ORDER: a slot is
                              // [group of 32 points][tile t, register quad q][lane][16 B (8 B as f16)], i.e. every store
                              // instruction writes 1 KiB (512 B) of consecutive bytes (csrc/vfn_dwf.hip reads it back);
                              // otherwise row-major [M][256] (f16: the first 512 bytes of every 1 KiB row).  The tanh'ed
                              // feature slot (8) is row-major fp32 either way.
    // split launches: feature operand blocks, 1 KiB per point: [tile t][lane half g][hi 2t | lo 2t | hi 2t+1 | lo 2t+1] x 16 B
    uint4* blk_out;           // M16_BLKOUT: written for rows 0 .. n_points-1
    const uint4* blk_in;      // M16_BLKIN: block buffer, groups of 32 rows (store_blocks)
    const float* vec_in;      // M16_BLKIN: [rows,3] vector outputs of the feature launches
    const int* src;           // M16_BLKIN: [rows] position of every row among the sorted samples (< 0: padding row)
    const int* out_index;     // fused launches, optional: outputs of point m go to row out_index[m] (< 0: dropped) instead of m
    long long ws_first, ws_points;   // training modes: this launch's points are points ws_first .. of a workspace sized for ws_points
                                     // (one workspace filled by several launches: the proposal samples, then the new fine samples)
    uint32_t* status;         // optional (vfn_f16x3_set_status): bit 0 is OR-ed in when a hidden activation reached the f16 clamp,
                              // bit 1 when an input (point coordinate / encoding operand) did — the result of such a launch is not
                              // fp32-equivalent and the caller should repeat it on the exact-fp32 kernels
    const int* n_dev;            // optional: the launch covers min(n_points, *n_dev) points — a count that only the device knows (the
                                 // samples with non-zero weight, csrc/vfn_train.hip); workgroups past it leave at once
    unsigned long long* clock;   // optional (vfn_f16x3_set_clock_probe): workgroup b < clock_slots of a fused VF + rendering launch leaves
    long long clock_slots;       // its shader-clock cycles (s_memtime) in clock[2b] and its 100 MHz ticks (s_memrealtime) in clock[2b+1]
};

// Range of the split-f16 operands: activations ride at 2^6 x their value (VFN16_XSCALE) and are clamped here, i.e. true
// activations above ~937 saturate.  The clamp keeps inf - inf = NaN out of the split; the status word reports that it acted.
#define VFN16_CLAMP 60000.0f
#ifndef VFN16_RANGE_TRACK
#define VFN16_RANGE_TRACK 2       // where the kernels look for values at the clamp: 0 nowhere, 1 per epilogue pair (+3.2 % on the fused
                                  // launch), 2 per finished tile (+1.3 %; placing it in the shadow of the next tile's first MFMAs measured the same)
#endif

struct X16 { half8 hi[16]; half8 lo[16]; };     // 256 activation columns x this lane's point, split (16 K-blocks of 16)
struct A16 { half8 hi[3]; half8 lo[3]; };       // 48 auxiliary (encoding) columns

struct Pipe16 {
    uint4* lds;                    // ring base
    __amdgpu_buffer_rsrc_t vf_w;   // the two f16 packs as buffer resources: the DMA addresses stay in SGPRs
    __amdgpu_buffer_rsrc_t rn_w;
    // training modes
    float* saved;
    long long slot_floats;         // M * 256
    uint32_t slot_bytes;           // M * 1024 (the host checks that it fits)
    uint32_t save_voff;            // fp32 stores of the ReLU slots: row-major m * 1024 + 16 * (lane >> 5); fragment order
                                   // (m >> 5) * 32768 + 16 * lane; out of range for m >= M
    uint32_t save_voff16;          // f16 stores: m * 1024 + 8 * (lane >> 5) / (m >> 5) * 32768 + 8 * lane
    uint32_t feat_voff;            // the feature slot (always row-major fp32): m * 1024 + 16 * (lane >> 5)
    uint32_t st_tile, st_q;        // byte strides of (tile, register quad) in a slot for fp32 stores (f16: half): 128, 32 row-major;
                                   // 4096, 1024 fragment order
    int save16;                    // training: ReLU slots hold f16 values (the tanh'ed feature slot stays fp32)
    uint32_t* masks;               // ReLU sign bits, 32 bytes per point and slot: [slot][m][lane half][4 dwords]
    uint32_t mask_bytes;           // M * 32
    uint32_t mask_voff;            // (2 m + (lane >> 5)) * 16, out of range for m >= M
    // feature operand blocks
    uint4* blk_out;
    uint32_t blk_bytes, blk_voff;  // rows * 1024; m * 1024 + 64 * (lane >> 5), out of range for m >= M
};

// State carried from chunk to chunk (and from layer call to layer call): the previous tile's accumulators, whose
// epilogue runs in the shadow of the next tile's MFMAs, and the next chunk's bias and first fragments, which are read
// before the current chunk ends so that no tile starts with an exposed LDS latency.
struct Carry16 {
    f32x16 pend;
    f32x16 bias;
    half8 fh0, fl0;
    uint32_t lm[4];      // training: sign bits of the layer being produced, 16 per finished tile (tile t -> half t & 1 of dword t >> 1)
    unsigned long long sat;   // lanes (as a wave mask, kept in scalar registers) whose ReLU output reached VFN16_CLAMP: the clamp acted
};

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// (hi, lo) halves of two values: hi = f16(v), lo = f16(v - hi).  The residual is one v_fma_mix_f32 per value (the f16
// half is converted inside the instruction; hipcc otherwise emits v_cvt_f32_f16 + v_sub_f32), and deliberately not a
// packed fp32 op: v_pk_add_f32 beside MFMAs costs ~13 extra cycles per instruction (MI355X_MICROARCH.md, "price of one
// filler beside MFMAs").
__device__ __forceinline__ void split2(float a, float b, _Float16& h0, _Float16& h1, _Float16& l0, _Float16& l1) {
    const float2v v = {a, b};
    const half2v hi = __builtin_convertvector(v, half2v);
    h0 = hi[0]; h1 = hi[1];
    const uint32_t hp = __builtin_bit_cast(uint32_t, hi);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(b));
    const float2v r = {r0, r1};
    const half2v lo = __builtin_convertvector(r, half2v);
    l0 = lo[0]; l1 = lo[1];
}

// tanh through one exp: 1 - 2 / (1 + e^{2v}).  Absolute error ~1e-7; saturates correctly to +-1.
__device__ __forceinline__ float tanh_exp(float v) { return 1.0f - 2.0f / (1.0f + expf(2.0f * v)); }

// One 1-KiB LDS-DMA piece: block `blk` of chunk descriptor D into ring slot SLOT.
template <int NET, int OFF_KB, int LDS_U4>
__device__ __forceinline__ void dma_piece(const Pipe16& p, int blk, int lane, int dst_blk = -1) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(NET ? p.rn_w : p.vf_w, (lds_void*)(p.lds + LDS_U4 + (dst_blk < 0 ? blk : dst_blk) * 64), 16,
                                             lane * 16, (OFF_KB + blk) * 1024, 0, 0);
}
template <int MODE, int C>
__device__ __forceinline__ void dma_chunk(const Pipe16& p, int wave, int lane) {     // whole chunk at once (prologue only)
    constexpr ChunkD d = chunk_of(MODE, C);
    if constexpr (mode_one(MODE)) {      // hi planes (even blocks) and the bias block (the last one)
        constexpr int NKB = (d.kb - 1) / 2;
#pragma unroll
        for (int i = 0; i * VFN16_WAVES < NKB + 1; ++i)
            if (wave + VFN16_WAVES * i < NKB + 1) dma_piece<d.net, d.off_kb, slot_base(MODE, C)>(p, 2 * (wave + VFN16_WAVES * i), lane, wave + VFN16_WAVES * i);
    } else {
#pragma unroll
    for (int i = 0; i * VFN16_WAVES < d.kb; ++i)
        if (wave + VFN16_WAVES * i < d.kb) dma_piece<d.net, d.off_kb, slot_base(MODE, C)>(p, wave + VFN16_WAVES * i, lane);
    }
}

template <int MODE, int C>
__device__ __forceinline__ void prefetch_chunk(Carry16& cy, const Pipe16& p, int lane) {
    constexpr ChunkD d = chunk_of(MODE, C);
    const uint4* cb = p.lds + slot_base(MODE, C);
    const f32x4v* bb = reinterpret_cast<const f32x4v*>(cb + (mode_one(MODE) ? (d.kb - 1) / 2 : d.kb - 1) * 64) + (lane >> 5) * 4;
    const f32x4v b0 = bb[0], b1 = bb[1], b2 = bb[2], b3 = bb[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) { cy.bias[q] = b0[q]; cy.bias[4 + q] = b1[q]; cy.bias[8 + q] = b2[q]; cy.bias[12 + q] = b3[q]; }
    cy.fh0 = __builtin_bit_cast(half8, cb[0 * 64 + lane]);
    if (!chunk_two(MODE, C) && !mode_one(MODE)) cy.fl0 = __builtin_bit_cast(half8, cb[1 * 64 + lane]);     // (no two-product chunk starts with an encoding block)
}

// Two accumulator values -> (hi, lo) halves of element pair (j, j+1) of an operand block.
template <int EPI, bool KEEP, bool LO = true>
__device__ __forceinline__ void epi_pair(f32x16& pend, unsigned long long& sat, int pr, half8& hi, half8& lo, int j) {
    float v0 = pend[2 * pr], v1 = pend[2 * pr + 1];
    // one compare per pair into a SCALAR accumulator (a vector accumulator carried through the pipelined loop made hipcc spill)
#if VFN16_RANGE_TRACK == 1
    if (EPI == EPI_RELU) sat |= __builtin_amdgcn_ballot_w64(fmaxf(v0, v1) >= VFN16_CLAMP);
#endif
    if (!VFN16_ASCALE || EPI != EPI_RELU) { v0 *= VFN16_INV_WSCALE; v1 *= VFN16_INV_WSCALE; }
    // ReLU, saturated below the f16 range so that an out-of-family activation degrades instead of turning into
    // inf - inf = NaN in the split (activations of BatchNorm'ed layers are O(1..100)); one v_max3 per pair remembers
    // whether the clamp ever acted (reported through the status word at the end of the kernel)
    if (EPI == EPI_RELU) {
        v0 = fminf(fmaxf(v0, 0.f), VFN16_CLAMP); v1 = fminf(fmaxf(v1, 0.f), VFN16_CLAMP);
    }
    else { v0 = tanh_exp(v0) * VFN16_XSCALE; v1 = tanh_exp(v1) * VFN16_XSCALE; }
    if constexpr (LO) {
        _Float16 h0, h1, l0, l1;
        split2(v0, v1, h0, h1, l0, l1);
        hi[j] = h0; hi[j + 1] = h1; lo[j] = l0; lo[j + 1] = l1;
    } else {                              // single-product mode: the f16 rounding is the operand
        const float2v v = {v0, v1};
        const half2v h = __builtin_convertvector(v, half2v);
        hi[j] = h[0]; hi[j + 1] = h[1];
    }
    if (KEEP) { pend[2 * pr] = v0 * (1.0f / VFN16_XSCALE); pend[2 * pr + 1] = v1 * (1.0f / VFN16_XSCALE); }   // training: the value the backward reads
}

// Training: registers 4q..4q+3 of a finished tile = 4 consecutive columns (32 TILE + 8 q + 4 (lane >> 5)) of this lane's
// point -> one 16-byte buffer store into slot SLOT (rows beyond M are dropped by the range check).
template <int SLOT, int TILE, bool SF16 = false>
__device__ __forceinline__ void save_group(const Pipe16& p, const f32x16& v, int q) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.saved + (long long)SLOT * p.slot_floats, 0,
                                                                        (int)p.slot_bytes, 0x00020000);
    const f32x4v g = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    if constexpr (SF16 && SLOT != 8) {   // f16 values, fragment order (2048 B per tile, 512 B per register quad): no run-time form
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const half4 h = __builtin_convertvector(g, half4);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs, (int)p.save_voff16, (TILE * 4096 + q * 1024) / 2, VFN16_SAVE_AUX);
        return;
    }
#if defined(ABL_SAVE_COALESCED)
    // timing only (WRONG layout): the same bytes as one 1-KiB run per instruction, to price the row-major store pattern
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)(((p.save_voff >> 15) << 15) + (threadIdx.x & 63) * 16),
                                           (4 * TILE + q) * 1024, VFN16_SAVE_AUX);
#elif !defined(ABL_NOSAVE)
    if (SLOT == 8) {                    // the tanh'ed features: row-major fp32 (returned to callers, read by the chain as values)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)p.feat_voff, (32 * TILE + 8 * q) * 4, 0);
    } else if (p.save16) {              // 11-bit operands for the weight gradients, half the workspace traffic
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const half4 h = __builtin_convertvector(g, half4);
        // (streaming stores only where an instruction writes whole lines, i.e. in fragment order)
        if (p.st_q == 1024u) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs, (int)p.save_voff16, (int)((TILE * p.st_tile + q * p.st_q) >> 1), VFN16_SAVE_AUX);
        else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs, (int)p.save_voff16, (int)((TILE * p.st_tile + q * p.st_q) >> 1), 0);
    } else {
        if (p.st_q == 1024u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)p.save_voff, (int)(TILE * p.st_tile + q * p.st_q), VFN16_SAVE_AUX);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), rs, (int)p.save_voff, (int)(TILE * p.st_tile + q * p.st_q), 0);
    }
#else
    asm volatile("" :: "v"(g));
#endif
}

// Training: the sign bits of a finished (activated) tile.  The dX chain needs of every ReLU output only whether it is
// positive: 16 bits per lane and tile (bit r <-> accumulator register r, the same layout the chain's tiles have) instead of
// 64 bytes of fp32 — the chain then reads 32 bytes per point and layer where it read 1 KiB.  The bits of a layer's tiles are
// collected in cy.lm and leave as ONE 16-byte store when its last tile is done.
template <int TILE, int EPI>
__device__ __forceinline__ void collect_mask(Carry16& cy) {
    if constexpr (EPI == EPI_RELU) {      // values are >= +0 here: positive <=> a non-zero bit pattern (two VALU ops per value)
        typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
        const u32x16 u = __builtin_bit_cast(u32x16, cy.pend);
        // (two VALU instructions per value; from `min(u, 1) << r` hipcc made v_cmp_class + v_cndmask + v_or3 through an SGPR pair per value,
        //  with the wait states that hazard costs — 55 issue slots per tile instead of 32)
        // u + 0x7fffffff has its top bit set exactly for u != 0 (u < 0x7f800000 here); a funnel shift moves it into the word: b = b << 1 | top
        unsigned b = 0;
#pragma unroll
        for (int r = 15; r >= 0; --r) b = __builtin_amdgcn_alignbit(b, u[r] + 0x7fffffffu, 31);
        cy.lm[TILE >> 1] |= b << (16 * (TILE & 1));
    }                                     // the tanh'ed feature block has no mask: the chain reads its values
}
// ... in PARTS pieces over as many K steps (16 / PARTS values each, highest registers first; `b` is carried by the caller): the whole
// collection in ONE step was 32 VALU instructions between three matrix instructions that cover 24
template <int TILE, int EPI, int PART, int PARTS>
__device__ __forceinline__ void collect_mask_part(Carry16& cy, unsigned& b) {
    if constexpr (EPI == EPI_RELU) {
        typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
        const u32x16 u = __builtin_bit_cast(u32x16, cy.pend);
        constexpr int N = 16 / PARTS;
        if constexpr (PART == 0) b = 0;
#pragma unroll
        for (int r = 15 - PART * N; r > 15 - (PART + 1) * N; --r) b = __builtin_amdgcn_alignbit(b, u[r] + 0x7fffffffu, 31);
        if constexpr (PART == PARTS - 1) cy.lm[TILE >> 1] |= b << (16 * (TILE & 1));
    }
}
template <int SLOT>
__device__ __forceinline__ void store_mask(const Pipe16& p, Carry16& cy) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(p.masks) + (size_t)SLOT * p.mask_bytes, 0, (int)p.mask_bytes, 0x00020000);
    const u32x4 w = {cy.lm[0], cy.lm[1], cy.lm[2], cy.lm[3]};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)p.mask_voff, 0, 0);
    cy.lm[0] = 0; cy.lm[1] = 0; cy.lm[2] = 0; cy.lm[3] = 0;
}

// One layer: xout <- f(W' [xin ; aux] + b') — NCH chunks (C0 .. C0+NCH-1 of the launch) of one 32-row output tile each.
// D layout of v_mfma_f32_32x32x16_f16: column = lane & 31 (the point), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
// registers 0..7 / 8..15 of tile t ARE operand K-blocks 2t / 2t+1 of the next layer (order absorbed into the pack).
//
// With one wave per SIMD nothing else hides the epilogue, the DMA issue or the ring hand-over, so the chunk loop is
// software-pipelined over the K steps of a tile (three MFMAs = 96 matrix-pipe cycles each, fenced by sched_barrier):
//   first half   the epilogue of the PREVIOUS tile (8 register pairs);
//   middle       s_waitcnt vmcnt(0) + s_barrier: chunk c+1 (issued half a chunk ago) has landed for everybody and
//                everybody has left chunk c-1, whose slot is now free;
//   second half  the LDS-DMA pieces of chunk c+2 into that slot;
//   last step    the bias and first fragments of chunk c+1 (cy).
// The last tile of a layer is handed to the next layer call in cy.pend (PEPI = its epilogue, PKB = the K-block pair of
// `xpend` it becomes); it is needed only by K steps PKB, PKB+1 of that layer's first tile.
// Operand blocks 2 TILE, 2 TILE + 1 of `x` (one finished feature tile) -> the block buffer, in the registers' own order:
// the 32 points of a wave form a 32 KiB group [operand block 0..15][hi | lo][lane 0..63][16 B], so every store (and every
// load of the rendering launch, whose waves own the same 32 rows) moves 1 KiB of consecutive bytes.
template <int TILE>
__device__ __forceinline__ void store_blocks(const Pipe16& p, const X16& x) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.blk_out, 0, (int)p.blk_bytes, 0x00020000);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x.hi[2 * TILE + s]), rs, (int)p.blk_voff, (2 * TILE + s) * 2048, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x.lo[2 * TILE + s]), rs, (int)p.blk_voff, (2 * TILE + s) * 2048 + 1024, 0);
    }
}

// BLK: the tiles this layer produces (and a pending tile it finishes, PBLK) are feature tiles that also go to the block buffer
template <int MODE, int C0, int ACT, int AUX, int NCH, int EPI, int PEPI, int PKB, int SLOT = -1, int PSLOT = -1, bool BLK = false,
          bool PBLK = false>
__device__ __forceinline__ void layer16(const X16& xin, const A16& aux, X16& xout, X16& xpend, Carry16& cy, float (&head)[3],
                                       const Pipe16& p, int wave, int lane) {
    constexpr int NKB = ACT + AUX;
    constexpr int E = NKB / 2 > 0 ? NKB / 2 : 1;        // steps that carry the pending tile's epilogue
    constexpr int H = NKB * VFN16_HANDOVER_NUM / 16 > 0 ? NKB * VFN16_HANDOVER_NUM / 16 : 1;   // steps before the hand-over
    constexpr int PMAX = (VFN16_MAX_CHUNK_KB + VFN16_WAVES - 1) / VFN16_WAVES;   // DMA pieces per wave and chunk
    constexpr int DSPAN = NKB - H > 0 ? NKB - H : 1;
    constexpr int DSTEPS = DSPAN < VFN16_DMA_STEPS ? DSPAN : VFN16_DMA_STEPS;          // steps that carry DMA pieces
    static_assert(PEPI < 0 || PKB >= E, "the pending tile must be complete before it is read");
    constexpr bool TRAIN = (MODE & M16_TRAIN) != 0;
    constexpr bool LATE = TRAIN && VFN16_LATE_STORES;
    // late stores: DMA pieces in steps [H, DEND), the (mask store and the) four stores of the pending tile in steps [DEND, NKB)
    constexpr int DEND = LATE ? (NKB - 4 > H + 1 ? NKB - 4 : (H + 1 < NKB ? H + 1 : NKB - 1)) : 0;
    constexpr int SBEG = LATE ? DEND : H;
    constexpr int SSTEPS = LATE ? (NKB - DEND > 0 ? NKB - DEND : 1) : (DSPAN < 4 ? DSPAN : 4);      // steps that carry the 4 stores of a tile
    constexpr int DST = LATE ? (DEND - H > 0 ? DEND - H : 1) : DSTEPS;                             // steps that carry DMA pieces
    // late stores: the sign-bit collection of the pending tile in MPARTS pieces over the steps [H, H + MPARTS) in front of its stores
    // (the epilogue pairs are done by step E <= H; 0: in one piece at the first store step)
    constexpr int MPARTS = (LATE && E <= H) ? (DEND - H >= 4 ? 4 : (DEND - H >= 2 ? 2 : (DEND - H >= 1 ? 1 : 0))) : 0;
    static_for<NCH>([&](auto ich) {
        constexpr int ch = decltype(ich)::value;
        constexpr int C = C0 + ch;
        constexpr int RING = ring_of(MODE);
        constexpr ChunkD dcur = chunk_of(MODE, C), dnext = chunk_of(MODE, C + 1), ddma = chunk_of(MODE, C + RING - 1);
        static_assert(dcur.kb == 2 * NKB + 1, "layer shape and chunk table disagree");
        constexpr bool P1 = mode_one(MODE);                // one product per K-block: hi planes only, no lo halves anywhere
        constexpr bool W2 = chunk_two(MODE, C);            // this tile: two products, hi planes of the weights only
        constexpr bool D2 = chunk_two(MODE, C + RING - 1) || P1;  // the chunk being fetched: hi planes (even blocks) + the bias block
        constexpr int DNKB = (ddma.kb - 1) / 2;             // pieces of a two-product chunk: hi planes, lo planes of the encoding blocks, bias
        constexpr int DPIECES = P1 ? DNKB + 1 : (D2 ? DNKB + ddma.aux + 1 : ddma.kb);
        auto piece_blk = [&](int idx) -> int {
            if (P1) return idx < DNKB ? 2 * idx : 2 * DNKB;
            if (!D2) return idx;
            if (ddma.aux == 0) return 2 * idx;
            return idx < DNKB ? 2 * idx : (idx < DNKB + ddma.aux ? 2 * (idx - ddma.aux) + 1 : 2 * DNKB);
        };
        constexpr int PM = D2 ? (DPIECES + VFN16_WAVES - 1) / VFN16_WAVES : PMAX;
        const uint4* cb = p.lds + slot_base(MODE, C);
        constexpr int FB = P1 ? 1 : 2;                      // blocks per K step in the slot (compact hi planes / hi + lo)
        f32x16 acc = cy.bias;
        half8 fh[VFN16_FDEPTH], fl[VFN16_FDEPTH];
        fh[0] = cy.fh0;
        if (!P1 && (!W2 || ACT == 0)) fl[0] = cy.fl0;
        if (VFN16_FDEPTH == 3 && NKB > 1) {
            fh[1] = __builtin_bit_cast(half8, cb[FB * 64 + lane]);
            if (!W2 && !P1) fl[1] = __builtin_bit_cast(half8, cb[3 * 64 + lane]);
        }
        half8 ehi[2], elo[2];
        [[maybe_unused]] unsigned mbits = 0;
#pragma unroll
        for (int st = 0; st < NKB; ++st) {
            constexpr int AHEAD = VFN16_FDEPTH - 1;
            if (st + AHEAD < NKB) {
                fh[(st + AHEAD) % VFN16_FDEPTH] = __builtin_bit_cast(half8, cb[(FB * (st + AHEAD)) * 64 + lane]);
                if (!P1 && (!W2 || st + AHEAD >= ACT)) fl[(st + AHEAD) % VFN16_FDEPTH] = __builtin_bit_cast(half8, cb[(2 * (st + AHEAD) + 1) * 64 + lane]);
            }
            const half8 a_hi = fh[st % VFN16_FDEPTH];
            const half8 x_hi = st < ACT ? xin.hi[st < ACT ? st : 0] : aux.hi[st >= ACT ? st - ACT : 0];
            const half8 x_lo = st < ACT ? xin.lo[st < ACT ? st : 0] : aux.lo[st >= ACT ? st - ACT : 0];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_hi, acc, 0, 0, 0);
#ifndef ABL_P1
            if constexpr (!P1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_lo, acc, 0, 0, 0);
                if (!W2 || st >= ACT) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[st % VFN16_FDEPTH], x_hi, acc, 0, 0, 0);
            }
#endif
            // -- first half: epilogue pairs of the pending tile
#ifdef ABL_NOEPI
            if (st == 0 && (ch > 0 || PEPI >= 0)) asm volatile("" :: "v"(cy.pend));
            if (false) {
#else
            if (st < E && (ch > 0 || PEPI >= 0)) {
#endif
#pragma unroll
                for (int pr = st * 8 / E; pr < (st + 1) * 8 / E; ++pr) {
                    const int sblk = pr >> 2, j = (pr & 3) * 2;
                    if (ch > 0) epi_pair<EPI, TRAIN, !P1>(cy.pend, cy.sat, pr, ehi[sblk], elo[sblk], j);
                    else epi_pair<(PEPI >= 0 ? PEPI : 0), TRAIN, !P1>(cy.pend, cy.sat, pr, ehi[sblk], elo[sblk], j);
                    if ((pr & 3) == 3) {
                        asm volatile("" : "+a"(ehi[sblk]));   // operands live in AGPRs (MFMA reads them there)
                        if constexpr (!P1) asm volatile("" : "+a"(elo[sblk]));
                        if (ch > 0) { xout.hi[2 * (ch > 0 ? ch - 1 : 0) + sblk] = ehi[sblk]; if constexpr (!P1) xout.lo[2 * (ch > 0 ? ch - 1 : 0) + sblk] = elo[sblk]; }
                        else { xpend.hi[PKB + sblk] = ehi[sblk]; if constexpr (!P1) xpend.lo[PKB + sblk] = elo[sblk]; }
                    }
                }
#ifndef VFN16_NOGROUPS
                if constexpr (!P1) {
                // one MFMA, then its share of the epilogue in that MFMA's shadow
                __builtin_amdgcn_sched_group_barrier(0x100, W2 ? 1 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, W2 ? VFN16_EPI_PER_MFMA2 : VFN16_EPI_PER_MFMA, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, W2 ? VFN16_EPI_PER_MFMA2 : VFN16_EPI_PER_MFMA, 0);
                if (!W2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VFN16_EPI_PER_MFMA, 0);
                }
                }
#endif
            }
            // -- middle: ring hand-over
            if (st == H - 1 && dnext.kb > 0) {
#ifndef ABL_NOSYNC
#ifdef ABL_LOOSE_WAIT      // timing-only: leave the four stores of a training tile in flight (not safe: see DESIGN.md)
                if ((MODE & M16_TRAIN) != 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else
#endif
                // late stores: the previous chunk ended with the four stores of its pending tile (every chunk but the launch's
                // first and the one after a layer that starts without a pending tile); everything older — the DMA pieces of
                // chunk c+1 among it — has landed once at most those four are outstanding
                if constexpr (P1 && LATE) {
                    // five-slot ring: younger than this wave's pieces of chunk c+1 are the stores of the last RING-2 chunks and the
                    // pieces of chunks c+2 .. c+RING-2 between them (lower bounds: a smaller immediate only waits for more)
                    constexpr int young = young_ops(MODE, C);
                    static_assert(young < 64, "vmcnt is a 6-bit field");
                    if constexpr (young == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(young) : "memory");
                } else
                if (LATE && C > 0 && (ch == 0 || ch > 1 || PEPI >= 0)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#endif
            }
            // -- second half: this step's share of the DMA pieces of chunk c+2
#ifndef ABL_NODMA
            if (st >= H && st < H + DST && ddma.kb > 0) {
#pragma unroll
                for (int i = (st - H) * PM / DST; i < (st - H + 1) * PM / DST; ++i) {
                    if (VFN16_WAVES * i + VFN16_WAVES <= DPIECES) dma_piece<ddma.net, ddma.off_kb, slot_base(MODE, C + RING - 1)>(p, piece_blk(wave + VFN16_WAVES * i), lane, P1 ? wave + VFN16_WAVES * i : -1);
                    else if (VFN16_WAVES * i < DPIECES) {
                        // the last, partial round of pieces (a 33-block chunk: its bias block): EVERY wave issues one — the waves past the end
                        // repeat the last piece (same bytes, same place) — instead of a uniform branch in the middle of this K step
                        const int idx = min(wave + VFN16_WAVES * i, DPIECES - 1);
                        dma_piece<ddma.net, ddma.off_kb, slot_base(MODE, C + RING - 1)>(p, piece_blk(idx), lane, P1 ? idx : -1);
                    }
                }
            }
#endif
            // -- training: the finished (activated) pending tile goes out after the hand-over, so that the stores have
            // half a chunk to retire before the next vmcnt(0)
#ifndef ABL_NOMASK
            // (late stores: the pending tile's sign bits are collected in the DMA steps, a quarter per step, ahead of its stores)
            if (TRAIN && MPARTS > 0 && st >= H && st < H + MPARTS && (ch > 0 || PEPI >= 0)) {
                static_for<MPARTS>([&](auto ip) {
                    constexpr int part = decltype(ip)::value;
                    if (st - H == part) {
                        if (ch > 0) collect_mask_part<(ch > 0 ? ch - 1 : 0), EPI, part, (MPARTS > 0 ? MPARTS : 1)>(cy, mbits);
                        else collect_mask_part<PKB / 2, (PEPI >= 0 ? PEPI : 0), part, (MPARTS > 0 ? MPARTS : 1)>(cy, mbits);
                    }
                });
            }
#endif
            if (TRAIN && st >= SBEG && st < SBEG + SSTEPS && (ch > 0 || PEPI >= 0)) {
#ifndef ABL_NOMASK
                if (st == (LATE ? SBEG : VFN16_MASK_STEP(H, NKB))) {       // its sign bits; the pending tile (ch == 0) is the last one of the previous layer
                    if constexpr (MPARTS > 0) {
                        if (ch == 0) store_mask<(PSLOT >= 0 ? PSLOT : 0)>(p, cy);
                    } else {
                        if (ch > 0) collect_mask<(ch > 0 ? ch - 1 : 0), EPI>(cy);
                        else { collect_mask<PKB / 2, (PEPI >= 0 ? PEPI : 0)>(cy); store_mask<(PSLOT >= 0 ? PSLOT : 0)>(p, cy); }
                    }
                }
#endif
#pragma unroll
                for (int q = (st - SBEG) * 4 / SSTEPS; q < (st - SBEG + 1) * 4 / SSTEPS; ++q) {
                    if (ch > 0) save_group<(SLOT >= 0 ? SLOT : 0), (ch > 0 ? ch - 1 : 0), (MODE & M16_SF16) != 0>(p, cy.pend, q);
                    else save_group<(PSLOT >= 0 ? PSLOT : 0), PKB / 2, (MODE & M16_SF16) != 0>(p, cy.pend, q);
                }
            }
            // -- split launches: a finished feature tile leaves as operand blocks, after the hand-over like the stores above
            if (st == H && ((BLK && ch > 0) || (PBLK && ch == 0))) {
                if (ch > 0) store_blocks<(ch > 0 ? ch - 1 : 0)>(p, xout);
                else store_blocks<PKB / 2>(p, xpend);
            }
            // -- last step: the next chunk's bias and first fragments
            if (st == NKB - 1 && dnext.kb > 0) prefetch_chunk<MODE, (dnext.kb > 0 ? C + 1 : C)>(cy, p, lane);
#ifndef VFN16_NOSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (EPI == EPI_RELU || EPI == EPI_TANH) {
            cy.pend = acc;
#if VFN16_RANGE_TRACK == 2
            if (EPI == EPI_RELU) {        // the finished tile's largest (2^6-scaled) pre-activation: 8 v_max3 + one compare per tile
                float mx = fmaxf(acc[0], acc[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(acc[r], acc[r + 1]));
                cy.sat |= __builtin_amdgcn_ballot_w64(mx >= VFN16_CLAMP);
            }
#endif
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = acc[c] * VFN16_INV_WSCALE;     // rows 0..2 = registers 0..2 of the lanes with lane < 32
                head[c] = (EPI == EPI_HEAD_TANH) ? tanh_exp(v) : 1.0f / (1.0f + expf(-v));
            }
        }
    });
}

// encoding columns [x(3), sin/cos(2^k x)...] of one 3-vector: column k of the aux operand
__device__ __forceinline__ float enc_value(const float (&x)[3], const float (&sn)[18], const float (&cs)[18], int multires, int k) {
    if (k < 3) return x[k];
    const int idx = k - 3, oct = idx / 6, rem = idx - 6 * oct;
    if (oct >= multires) return 0.f;
    return rem < 3 ? sn[3 * oct + rem] : cs[3 * oct + rem - 3];
}

// aux operand from a column generator: element j of lane half g of K-block s <-> column 16 s + 8 g + j
template <typename F>
__device__ __forceinline__ void build_aux(A16& aux, int g, F col, float& ain) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float v0, v1;
            if (g == 0) { v0 = col(16 * s + j); v1 = col(16 * s + j + 1); }
            else { v0 = col(16 * s + 8 + j); v1 = col(16 * s + 8 + j + 1); }
            v0 *= VFN16_XSCALE; v1 *= VFN16_XSCALE;
            // inputs beyond the f16 range (|coordinate| > ~937): clamp (no inf - inf in the split) and remember
            v0 = fminf(fmaxf(v0, -VFN16_CLAMP), VFN16_CLAMP); v1 = fminf(fmaxf(v1, -VFN16_CLAMP), VFN16_CLAMP);
            ain = fmaxf(ain, fmaxf(fabsf(v0), fabsf(v1)));
            _Float16 h0, h1, l0, l1;
            split2(v0, v1, h0, h1, l0, l1);
            hi[j] = h0; hi[j + 1] = h1; lo[j] = l0; lo[j + 1] = l1;
        }
        aux.hi[s] = hi; aux.lo[s] = lo;
    }
}

// training: this lane's columns of the encoding tile -> dst[m][40] (column 16 s + 8 g + j; 16-byte groups below 40)
template <typename F>
__device__ __forceinline__ void save_aux(float* dst, long long m, int g, F col) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = 16 * s + 8 * g + 4 * h;
            if (c0 + 4 <= VFN_AUX_K) {
                f32x4v v;
                if (g == 0) { v = f32x4v{col(16 * s + 4 * h), col(16 * s + 4 * h + 1), col(16 * s + 4 * h + 2), col(16 * s + 4 * h + 3)}; }
                else { v = f32x4v{col(16 * s + 8 + 4 * h), col(16 * s + 8 + 4 * h + 1), col(16 * s + 8 + 4 * h + 2), col(16 * s + 8 + 4 * h + 3)}; }
                *reinterpret_cast<f32x4v*>(dst + m * VFN_AUX_K + c0) = v;
            }
        }
}

__device__ __forceinline__ void load_aux(A16& ax, const float* park) {
    const half8* pk = reinterpret_cast<const half8*>(park + 8);
#pragma unroll
    for (int q = 0; q < 3; ++q) { ax.hi[q] = pk[2 * q]; ax.lo[q] = pk[2 * q + 1]; }
}

// The rendering net on a feature operand that is already in `xb` (fused launches: straight from the VF net's epilogue;
// M16_RN_BLK: gathered from the block buffer): aux = [p(3), d(3), sin/cos(2^k d)(6L), n(3)], four hidden layers, rgb head.
// sin / cos of 2^o x for the octaves o < multires of one 3-vector.  The two lanes that share a point (lane halves g = 0, 1)
// each evaluate ONE octave of every pair (2j + g: same instruction stream, different argument) and swap results, so a lane
// runs 3 * ceil(multires / 2) sincosf instead of 3 * multires — same inputs to the same function, so the values are
// bit-identical to evaluating all of them.
__device__ __forceinline__ void encode_sincos(const float (&x)[3], int multires, int g, float (&sn)[18], float (&cs)[18]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = 0.f, k = 0.f;
            if (2 * j < multires) {
                sincosf(x[c] * (g ? (float)(2 << (2 * j)) : (float)(1 << (2 * j))), &s, &k);
                if (2 * j + g >= multires) { s = 0.f; k = 0.f; }      // odd multires: the pair's upper octave does not exist
            }
            const float so = __shfl_xor(s, 32, 64), ko = __shfl_xor(k, 32, 64);
            sn[6 * j + c] = g ? so : s;          cs[6 * j + c] = g ? ko : k;              // octave 2j
            sn[6 * j + 3 + c] = g ? s : so;      cs[6 * j + 3 + c] = g ? k : ko;          // octave 2j + 1
        }
}

// aux operand of the rendering net for one sample: [p(3), d(3), sin/cos(2^k d)(6L), n(3)]
template <int MODE>
__device__ __forceinline__ void render_aux(const Mlp16Args& a, A16& aux, const float (&xr)[3], const float (&dr)[3], const float (&nrm)[3],
                                           long long m, bool in, int g, float& ain) {
    const int rn_multires = a.rn_multires;
    float sn[18], cs[18];
    encode_sincos(dr, rn_multires, g, sn, cs);
    const int ncol = 6 + 6 * rn_multires;   // first normal column
    auto rn_col = [&](int k) -> float {
        if (k < 3) return xr[k];
        if (k >= ncol) return k < ncol + 3 ? nrm[k - ncol] : 0.f;
        return enc_value(dr, sn, cs, rn_multires, k - 3);
    };
    build_aux(aux, g, rn_col, ain);
    if ((MODE & M16_TRAIN) && in) save_aux(a.save_aux_rn, m, g, rn_col);
}

template <int MODE>
__device__ __forceinline__ void render_tail(const Mlp16Args& a, const Pipe16& p, Carry16& cy, X16& xa, X16& xb, const A16& aux,
                                            const float (&nrm)[3], long long m, bool in, int wave, int lane) {
    constexpr int R = EPI_RELU, NONE = -1;
    constexpr int C0 = rn_first_chunk(MODE);
    float rgb[3] = {0.f, 0.f, 0.f};
    layer16<MODE, C0 + 0, 16, 3, 8, R, NONE, 0, 9, -1>(xb, aux, xa, xb, cy, rgb, p, wave, lane);      // R0: [features ; p, PE(d), n]
    layer16<MODE, C0 + 8, 16, 0, 8, R, R, 14, 10, 9>(xa, aux, xb, xa, cy, rgb, p, wave, lane);        // R1
    layer16<MODE, C0 + 16, 16, 0, 8, R, R, 14, 11, 10>(xb, aux, xa, xb, cy, rgb, p, wave, lane);      // R2
    layer16<MODE, C0 + 24, 16, 0, 8, R, R, 14, 12, 11>(xa, aux, xb, xa, cy, rgb, p, wave, lane);      // R3 -> xb
    layer16<MODE, C0 + 32, 16, 0, 1, EPI_HEAD_SIGMOID, R, 14, -1, 12>(xb, aux, xa, xb, cy, rgb, p, wave, lane);
    // outputs last: the only vector-memory stores of the kernel come after the last DMA wait
    const long long mo = m;        // fused launches: the point's own row; from blocks: its position among the sorted samples
    if (in && (threadIdx.x & 32) == 0) {
        a.out_vec[mo * 3 + 0] = nrm[0]; a.out_vec[mo * 3 + 1] = nrm[1]; a.out_vec[mo * 3 + 2] = nrm[2];
        a.out_colors[mo * 3 + 0] = rgb[0]; a.out_colors[mo * 3 + 1] = rgb[1]; a.out_colors[mo * 3 + 2] = rgb[2];
    }
}

// End of a launch: tell the caller when an operand left the range the split-f16 representation covers.  Lanes of rows past
// the end evaluate the point (0, 0, 0), which is an in-range input like any other; a divergent atomic only where it matters.
__device__ __forceinline__ void report_range(const Mlp16Args& a, unsigned long long sat, float ain) {
    if (a.status) {
        const unsigned bits = (sat != 0ull ? 1u : 0u) | (ain >= VFN16_CLAMP ? 2u : 0u);
        if (bits) atomicOr(a.status, bits);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void vfn_mlp16_kernel(const Mlp16Args a) {
    // ONE __shared__ object: a second one beside an LDS-DMA destination makes hipcc drain vmcnt(0) before every
    // first ds_read after a DMA issue (cdna_hip_programming.md, "three .s-level traps")
    __shared__ __attribute__((aligned(16))) uint4 s_ring[3 * VFN16_SLOT + 256 * 8];
    // per-thread parking space (128 B): the point, its view direction and the encoding operand of the VF net, which is
    // needed again only at the skip layer
    float* s_park = reinterpret_cast<float*>(s_ring + 3 * VFN16_SLOT) + threadIdx.x * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5;
#ifdef VFN16_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // clock probe (fused VF + rendering launches only; scalar registers, a uniform branch): what the shader clock was while THIS
    // workgroup ran — the kernel is power-limited, so the clock is a result, not a constant (bench.py: roofline.effective_clock_ghz)
    unsigned long long ck_t0 = 0ull, ck_r0 = 0ull;
    if constexpr ((MODE & M16_RENDER) != 0 && (MODE & M16_BLKIN) == 0 && (MODE & M16_TRAIN) == 0) {
        if (a.clock) { ck_t0 = __builtin_amdgcn_s_memtime(); ck_r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    const long long m = (long long)blockIdx.x * VFN16_PTS + wave * 32 + (lane & 31);
    long long n_live = a.n_points;
    if (a.n_dev) {             // (uniform: a scalar load and a scalar branch, before anything is staged or any barrier is reached)
        const long long nd = (long long)*a.n_dev;
        n_live = nd < n_live ? nd : n_live;
        if ((long long)blockIdx.x * VFN16_PTS >= n_live) return;
    }
    const bool in = m < n_live;

    if constexpr ((MODE & M16_BLKIN) != 0) {
        // ---- rendering net only: this point's feature operand, normal, position and view direction come from memory ----
        // row m of the block / vector buffers is this lane's sample; dst[m] is its position among the sorted samples
        // (points, view direction and outputs live there), negative for padding rows
        // Memory order of the prologue (loads return in order, so every wait is a wait for everything issued before it):
        // dst[m] alone and its round trip; then, in ONE batch, the three small loads that depend on it, the first two weight
        // chunks (LDS-DMA) and the 128 KiB of feature operands; one wait.
        const int dpos = in ? a.src[m] : -1;
        const bool live = dpos >= 0;
        float xr[3] = {0.f, 0.f, 0.f}, dr[3] = {0.f, 0.f, 0.f}, nrm[3] = {0.f, 0.f, 0.f};
        if (live) {
            const long long di = dpos / a.dirs_div;
#pragma unroll
            for (int c = 0; c < 3; ++c) { xr[c] = a.points[(long long)dpos * 3 + c]; dr[c] = a.ray_dirs[di * 3 + c]; nrm[c] = a.vec_in[m * 3 + c]; }
        }
        Pipe16 p;
        p.lds = s_ring;
        p.vf_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.rn_w), 0, (int)a.rn_bytes, 0x00020000);
        p.rn_w = p.vf_w;
        p.saved = nullptr; p.slot_floats = 0; p.slot_bytes = 0; p.save_voff = 0; p.blk_out = nullptr; p.blk_bytes = 0; p.blk_voff = 0;
        p.masks = nullptr; p.mask_bytes = 0; p.mask_voff = 0; p.save16 = 0; p.save_voff16 = 0; p.feat_voff = 0; p.st_tile = 0; p.st_q = 0;
        dma_chunk<MODE, 0>(p, wave, lane);
        dma_chunk<MODE, 1>(p, wave, lane);
        X16 xa, xb;
        A16 aux;
        float ain_blk = 0.f;
        {
            // 32 KiB per 32 rows, see store_blocks; rows past the end get an out-of-range offset and read as zeros (no branch)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs_blk = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint4*>(a.blk_in), 0, (int)(((a.n_points + 31) & ~31ll) * 1024), 0x00020000);
            const unsigned goff = in ? (unsigned)((m >> 5) * 32768 + lane * 16) : 0xfffffff0u;
            // all 32 loads first, the register pinning afterwards: with the pins (volatile asm) inside the load loop hipcc
            // waits for every group of four loads before it issues the next — eight exposed round trips per tile
            u32x4 q[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_blk, goff, i * 1024, 0);
            // the encoding of the view direction (12 sincosf) while the operands are on their way: the three small loads were
            // issued before them, so their wait is a counted one
            render_aux<MODE>(a, aux, xr, dr, nrm, -1, false, g, ain_blk);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                half8 h0 = __builtin_bit_cast(half8, q[4 * t + 0]), l0 = __builtin_bit_cast(half8, q[4 * t + 1]);
                half8 h1 = __builtin_bit_cast(half8, q[4 * t + 2]), l1 = __builtin_bit_cast(half8, q[4 * t + 3]);
                asm volatile("" : "+a"(h0)); asm volatile("" : "+a"(l0)); asm volatile("" : "+a"(h1)); asm volatile("" : "+a"(l1));
                xb.hi[2 * t] = h0; xb.lo[2 * t] = l0; xb.hi[2 * t + 1] = h1; xb.lo[2 * t + 1] = l1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunks 0 and 1 landed
        __builtin_amdgcn_s_barrier();
        Carry16 cy;
#pragma unroll
        for (int q = 0; q < 16; ++q) cy.pend[q] = 0.f;
        cy.lm[0] = 0; cy.lm[1] = 0; cy.lm[2] = 0; cy.lm[3] = 0; cy.sat = 0ull;
        prefetch_chunk<MODE, 0>(cy, p, lane);
#ifdef VFN16_STAMPS
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
        render_tail<MODE>(a, p, cy, xa, xb, aux, nrm, live ? (long long)dpos : -1, live, wave, lane);
        report_range(a, cy.sat, ain_blk);
#ifdef VFN16_STAMPS
        if (threadIdx.x == 0 && live) {
            const unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
            a.out_colors[(long long)dpos * 3 + 0] = (float)(st_t1 - st_t0); a.out_colors[(long long)dpos * 3 + 1] = (float)(t2 - st_t0);
            a.out_colors[(long long)dpos * 3 + 2] = (float)(r2 - st_r0);
        }
#endif
    } else {
    // this lane's point (the two lane halves of a wave share the 32 points); loaded BEFORE any DMA
    float x[3] = {0.f, 0.f, 0.f};
    if (in) { x[0] = a.points[m * 3 + 0]; x[1] = a.points[m * 3 + 1]; x[2] = a.points[m * 3 + 2]; }
    float d[3] = {0.f, 0.f, 0.f};
    int out_row = -1;              // scatter launches: where this point's outputs go
    if ((MODE & M16_RENDER) && in) {
        const long long di = m / a.dirs_div;
        d[0] = a.ray_dirs[di * 3 + 0]; d[1] = a.ray_dirs[di * 3 + 1]; d[2] = a.ray_dirs[di * 3 + 2];
        if (a.out_index) out_row = a.out_index[m];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE & M16_RENDER) {
        s_park[0] = x[0]; s_park[1] = x[1]; s_park[2] = x[2];
        s_park[4] = d[0]; s_park[5] = d[1]; s_park[6] = d[2];
    }

    Pipe16 p;
    p.lds = s_ring;
    p.vf_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.vf_w), 0, (int)a.vf_bytes, 0x00020000);
    p.rn_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>((MODE & M16_RENDER) ? a.rn_w : a.vf_w), 0,
                                               (int)((MODE & M16_RENDER) ? a.rn_bytes : a.vf_bytes), 0x00020000);
    const bool frag = (a.save_f16 & 2) != 0;          // fragment-ordered slots: groups of 32 points, 32 KiB each
    const long long mw = m + a.ws_first;              // this point's place in the workspace
    p.slot_floats = frag ? ((a.ws_points + 31) >> 5) * 8192 : a.ws_points * 256;
    // The slot descriptors start at THIS LAUNCH's first point (1 KiB per point in either order; ws_first is a multiple of 32 in fragment
    // order) and the 32-bit offsets below are launch-relative: a launch covers < 2^21 points, the workspace may hold any number
    // (round 5: a step of 8 192 rays x 128 samples is 2.3 M workspace points).
    p.saved = a.saved + a.ws_first * 256;
    {
        const long long left = p.slot_floats * 4 - a.ws_first * 1024;
        p.slot_bytes = (uint32_t)(left < 0x7fffffffll ? left : 0x7fffffffll);
    }
    // rows past the end get an offset beyond the descriptor's range (the store is dropped).  Fragment order adds scalar
    // offsets of up to 32 KiB to it, so that value must not wrap: 3 GiB, with a launch limited to 2 GiB per slot (the host checks)
    p.save_voff = !in ? (frag ? 0xc0000000u : 0xfffffff0u) : (frag ? (uint32_t)((m >> 5) * 32768 + lane * 16) : (uint32_t)(m * 1024 + g * 16));
    p.save_voff16 = !in ? (frag ? 0xc0000000u : 0xfffffff0u) : (frag ? (uint32_t)((m >> 5) * 32768 + lane * 8) : (uint32_t)(m * 1024 + g * 8));
    p.feat_voff = in ? (uint32_t)(m * 1024 + g * 16) : 0xfffffff0u;
    p.st_tile = frag ? 4096u : 128u; p.st_q = frag ? 1024u : 32u;
    p.masks = a.save_masks; p.mask_bytes = (uint32_t)(a.ws_points * 32); p.mask_voff = in ? (uint32_t)((2 * mw + g) * 16) : 0xfffffff0u;
    p.save16 = a.save_f16 & 1;
    p.blk_out = a.blk_out; p.blk_bytes = (uint32_t)(((a.n_points + 31) & ~31ll) * 1024);
    p.blk_voff = (uint32_t)((m >> 5) * 32768 + lane * 16);      // past the last group -> out of the descriptor's range, dropped
    dma_chunk<MODE, 0>(p, wave, lane);
    dma_chunk<MODE, 1>(p, wave, lane);
    if constexpr (mode_one(MODE)) static_for<VFN16_P1_RING - 3>([&](auto ic) { dma_chunk<MODE, 2 + decltype(ic)::value>(p, wave, lane); });
#ifdef ABL_NODMA
    // timing only (WRONG results): no DMA inside the layers; the three slots keep three full-size chunks of real weights,
    // so the matrix cores still see random operands (an empty ring would feed zeros, which raises the clock)
    dma_chunk<MODE, 9>(p, wave, lane);      // chunk 9 -> slot 0, 10 -> slot 1, 11 -> slot 2 (tiles of VF layer 1, 33 KiB each)
    dma_chunk<MODE, 10>(p, wave, lane);
    dma_chunk<MODE, 11>(p, wave, lane);
#endif

    // ---- positional encoding of the point -> aux operand (and its parked copy for the skip layer) -----------
    const int vf_multires = a.vf_multires;
    A16 aux;
    float ain = 0.f;
    {
        float sn[18], cs[18];
        encode_sincos(x, vf_multires, g, sn, cs);
        build_aux(aux, g, [&](int k) { return enc_value(x, sn, cs, vf_multires, k); }, ain);
        if ((MODE & M16_TRAIN) && in) save_aux(a.save_aux_vf, mw, g, [&](int k) { return enc_value(x, sn, cs, vf_multires, k); });
        half8* pk = reinterpret_cast<half8*>(s_park + 8);
#pragma unroll
        for (int q = 0; q < 3; ++q) { pk[2 * q] = aux.hi[q]; pk[2 * q + 1] = aux.lo[q]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunks 0 and 1 landed
    __builtin_amdgcn_s_barrier();
    Carry16 cy;
#pragma unroll
    for (int q = 0; q < 16; ++q) cy.pend[q] = 0.f;
    cy.lm[0] = 0; cy.lm[1] = 0; cy.lm[2] = 0; cy.lm[3] = 0; cy.sat = 0ull;
    prefetch_chunk<MODE, 0>(cy, p, lane);

    // ---- VF net: straight-line code with static operand-set roles, so that only one set plus the tiles produced so far
    // are ever live
    X16 xa, xb;
    float vec[3] = {0.f, 0.f, 0.f};
#ifdef VFN16_STAMPS
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();      // prologue done
#endif
    constexpr int R = EPI_RELU, T = EPI_TANH, NONE = -1;
    layer16<MODE, 0, 0, 3, 8, R, NONE, 0, 0, -1>(xa, aux, xb, xb, cy, vec, p, wave, lane);        // L0: encoding only
    layer16<MODE, 8, 16, 0, 8, R, R, 14, 1, 0>(xb, aux, xa, xb, cy, vec, p, wave, lane);         // L1
    layer16<MODE, 16, 16, 0, 8, R, R, 14, 2, 1>(xa, aux, xb, xa, cy, vec, p, wave, lane);        // L2
    layer16<MODE, 24, 16, 0, 7, R, R, 14, 3, 2>(xb, aux, xa, xb, cy, vec, p, wave, lane);        // L3: 217 outputs
    { A16 ax; load_aux(ax, s_park); layer16<MODE, 31, 14, 3, 8, R, R, 12, 4, 3>(xa, ax, xb, xa, cy, vec, p, wave, lane); }   // L4: skip
    layer16<MODE, 39, 16, 0, 8, R, R, 14, 5, 4>(xb, aux, xa, xb, cy, vec, p, wave, lane);        // L5
    layer16<MODE, 47, 16, 0, 8, R, R, 14, 6, 5>(xa, aux, xb, xa, cy, vec, p, wave, lane);        // L6
    layer16<MODE, 55, 16, 0, 8, R, R, 14, 7, 6>(xb, aux, xa, xb, cy, vec, p, wave, lane);        // L7 -> xa
    if constexpr (!(MODE & M16_FEAT)) {
        layer16<MODE, 63, 16, 0, 1, EPI_HEAD_TANH, R, 14, -1, 7>(xa, aux, xb, xa, cy, vec, p, wave, lane);
        if (in && g == 0) { a.out_vec[m * 3 + 0] = vec[0]; a.out_vec[m * 3 + 1] = vec[1]; a.out_vec[m * 3 + 2] = vec[2]; }
        report_range(a, cy.sat, ain);
    } else {
    // fused: feature block (tanh) -> xb, then the vector head from the same input; the head's tile hosts the epilogue
    // of the last feature tile
    constexpr bool BO = (MODE & M16_BLKOUT) != 0;
    layer16<MODE, 63, 16, 0, 8, T, R, 14, 8, 7, BO, false>(xa, aux, xb, xa, cy, vec, p, wave, lane);
    layer16<MODE, 71, 16, 0, 1, EPI_HEAD_TANH, T, 14, -1, 8, false, BO>(xa, aux, xb, xb, cy, vec, p, wave, lane);
    if constexpr (!(MODE & M16_RENDER)) {
        if (in && g == 0) { a.out_vec[m * 3 + 0] = vec[0]; a.out_vec[m * 3 + 1] = vec[1]; a.out_vec[m * 3 + 2] = vec[2]; }
        report_range(a, cy.sat, ain);
#ifdef VFN16_STAMPS
        if (threadIdx.x == 0) {   // timing-only build: prologue cycles, total cycles, total 100 MHz ticks of this workgroup
            const unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
            a.out_vec[m * 3 + 0] = (float)(st_t1 - st_t0); a.out_vec[m * 3 + 1] = (float)(t2 - st_t0); a.out_vec[m * 3 + 2] = (float)(r2 - st_r0);
        }
#endif
    } else {
    // the head's outputs sit in the lanes < 32; the other lane half of the same point needs them for the aux operand
    float nrm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = __shfl(vec[c], lane & 31, 64);
    const float xr[3] = {s_park[0], s_park[1], s_park[2]};
    const float dr[3] = {s_park[4], s_park[5], s_park[6]};
    A16 raux;
    render_aux<MODE>(a, raux, xr, dr, nrm, (MODE & M16_TRAIN) ? mw : m, in, g, ain);
    render_tail<MODE>(a, p, cy, xa, xb, raux, nrm, a.out_index ? (long long)out_row : m, a.out_index ? out_row >= 0 : in, wave, lane);
    report_range(a, cy.sat, ain);
    if constexpr ((MODE & M16_TRAIN) == 0) {
        if (a.clock && threadIdx.x == 0 && (long long)blockIdx.x < a.clock_slots) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            a.clock[2 * blockIdx.x] = t1 - ck_t0; a.clock[2 * blockIdx.x + 1] = r1 - ck_r0;
        }
    }
    const long long mo = (long long)blockIdx.x * VFN16_PTS + (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
    (void)mo;
#ifdef VFN16_STAMPS
    if (threadIdx.x == 0) {   // timing-only build: shader-clock and 100 MHz ticks of this workgroup, over its first colours
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        a.out_colors[mo * 3 + 0] = (float)(t1 - st_t0); a.out_colors[mo * 3 + 1] = (float)(r1 - st_r0);
    }
#endif
    }   // rendering net
    }   // feature block
    }   // not M16_BLKIN
}

// a network's plan against the compile-time tables
int check_vf16(const Plan16& vf, const char* what) {
    bool ok = vf.feat_layer == 1 && vf.n_hidden == 9 && (int)vf.head_off_kb == vf_off_kb(9);
    for (int h = 0; ok && h < 9; ++h)
        ok = vf.act16[h] == VF_ACT[h] && vf.aux16[h] == VF_AUX[h] && vf.n_tiles[h] == VF_TILES[h] && (int)vf.off_kb[h] == vf_off_kb(h);
    if (!ok) { vfn_set_error("%s: the f16x3 kernels are specialised for the shipped vector-field network shape", what); return VFN_ERR_UNSUPPORTED; }
    return VFN_OK;
}
int check_rn16(const Plan16& rn, const char* what) {
    bool ok = rn.n_hidden == 4 && (int)rn.head_off_kb == rn_off_kb(4);
    for (int h = 0; ok && h < 4; ++h)
        ok = rn.act16[h] == RN_ACT[h] && rn.aux16[h] == RN_AUX[h] && rn.n_tiles[h] == RN_TILES[h] && (int)rn.off_kb[h] == rn_off_kb(h);
    if (!ok) { vfn_set_error("%s: the f16x3 kernels are specialised for the shipped rendering network shape", what); return VFN_ERR_UNSUPPORTED; }
    return VFN_OK;
}

}  // namespace

static int check_kind16(int net_kind, const Plan16& p, const char* what) {
    return net_kind == VFN_NET_VF ? check_vf16(p, what) : check_rn16(p, what);
}

// Where the f16x3 launches of THIS thread report operands outside the split-f16 range (NULL: nowhere).
static thread_local uint32_t* t_status_word = nullptr;

// (csrc/vfn_bstat.hip: the split-f16 layer products report into the same word)
uint32_t* vfn_internal_f16x3_status() { return t_status_word; }

extern "C" int vfn_f16x3_set_status(uint32_t* status_word) {
    t_status_word = status_word;
    return VFN_OK;
}

// Where the fused VF + rendering launches of THIS thread leave their per-workgroup clock stamps (NULL: nowhere).
static thread_local unsigned long long* t_clock_probe = nullptr;
static thread_local long long t_clock_slots = 0;

extern "C" int vfn_f16x3_set_clock_probe(uint64_t* stamps, int64_t slots) {
    VFN_REQUIRE(slots >= 0 && (stamps || slots == 0), "vfn_f16x3_set_clock_probe: %lld slots without a buffer", (long long)slots);
    t_clock_probe = reinterpret_cast<unsigned long long*>(stamps);
    t_clock_slots = stamps ? slots : 0;
    return VFN_OK;
}

VfnReportScope::VfnReportScope(uint32_t* status_word, uint64_t* clock_stamps, int64_t clock_slots)
    : old_status(t_status_word), old_clock(t_clock_probe), old_slots(t_clock_slots) {
    if (status_word) t_status_word = status_word;
    if (clock_stamps && clock_slots > 0) { t_clock_probe = reinterpret_cast<unsigned long long*>(clock_stamps); t_clock_slots = clock_slots; }
}
VfnReportScope::~VfnReportScope() { t_status_word = old_status; t_clock_probe = old_clock; t_clock_slots = old_slots; }

extern "C" int vfn_vf_mlp16_fwd(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                                 float* out_vec, void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32; Plan16 vf;
    int rc = make_plan16(VFN_NET_VF, geom, &p32, &vf, "vfn_vf_mlp16_fwd");
    if (rc != VFN_OK) return rc;
    rc = check_vf16(vf, "vfn_vf_mlp16_fwd");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(packed16 && points && out_vec, "vfn_vf_mlp16_fwd: NULL argument");
    a.vf_w = (const uint4*)packed16; a.points = points; a.out_vec = out_vec; a.n_points = n_points; a.dirs_div = 1;
    a.vf_multires = vf.multires; a.vf_bytes = vf.total_kb * 1024u;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_VEC>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_mlp16_fwd");
}

// fused launch: proposal samples in place (out_index NULL) or scattered; colour_products 3 (fp32-equivalent everywhere) or 2
static int launch_fused16(const char* what, const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                          const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                          int32_t samples_per_ray, const int32_t* out_index, bool scatter, int32_t colour_products, float* normals,
                          float* colors, void* stream, const int32_t* n_dev = nullptr) {
    Mlp16Args a = {};
    a.n_dev = n_dev;
    VfnNetPlan p32; Plan16 vf, rn;
    VFN_REQUIRE(vf_geom && rn_geom, "%s: NULL argument", what);
    int rc = make_plan16(VFN_NET_VF, vf_geom, &p32, &vf, what);
    if (rc != VFN_OK) return rc;
    rc = make_plan16(VFN_NET_RENDER, rn_geom, &p32, &rn, what);
    if (rc != VFN_OK) return rc;
    rc = check_vf16(vf, what);
    if (rc != VFN_OK) return rc;
    rc = check_rn16(rn, what);
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(vf_geom->feature_dims == VFN_HIDDEN && rn_geom->feature_dims == VFN_HIDDEN, "%s: both nets need feature_dims == %d", what, VFN_HIDDEN);
    VFN_REQUIRE(colour_products == 2 || colour_products == 3, "%s: colour_products must be 2 or 3 (got %d)", what, colour_products);
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed16 && rn_packed16 && points && ray_dirs && normals && colors && (out_index || !scatter), "%s: NULL argument", what);
    VFN_REQUIRE(samples_per_ray > 0, "%s: samples_per_ray must be > 0", what);
    a.vf_w = (const uint4*)vf_packed16; a.rn_w = (const uint4*)rn_packed16; a.points = points; a.ray_dirs = ray_dirs;
    a.out_vec = normals; a.out_colors = colors; a.out_index = out_index; a.n_points = n_points; a.dirs_div = samples_per_ray;
    a.vf_multires = vf.multires; a.rn_multires = rn.multires; a.vf_bytes = vf.total_kb * 1024u; a.rn_bytes = rn.total_kb * 1024u;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    if (colour_products == 2) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED | M16_C2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch(what);
}

extern "C" int vfn_vf_render_fused16_fwd(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                          const void* rn_packed16, const float* points, const float* ray_dirs,
                                          int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                          void* stream) {
    return launch_fused16("vfn_vf_render_fused16_fwd", vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, n_points, samples_per_ray,
                          nullptr, false, 3, normals, colors, stream);
}

extern "C" int vfn_vf_render_fused16_scatter(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                          const void* rn_packed16, const float* points, const float* ray_dirs,
                                          int64_t n_points, int32_t samples_per_ray, const int32_t* out_index, float* normals,
                                          float* colors, void* stream) {
    return launch_fused16("vfn_vf_render_fused16_scatter", vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, n_points,
                          samples_per_ray, out_index, true, 3, normals, colors, stream);
}

extern "C" int vfn_vf_render_fused16_products(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                              const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                                              int32_t samples_per_ray, const int32_t* out_index, int32_t colour_products,
                                              float* normals, float* colors, void* stream) {
    return launch_fused16("vfn_vf_render_fused16_products", vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, n_points,
                          samples_per_ray, out_index, false, colour_products, normals, colors, stream);
}

int vfn_internal_fused16_products_dev(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom, const void* rn_packed16,
                                      const float* points, const float* ray_dirs, int64_t n_points, const int32_t* n_dev, int32_t samples_per_ray,
                                      const int32_t* out_index, int32_t colour_products, float* normals, float* colors, void* stream) {
    return launch_fused16("vfn_render_fwd (colour branch on the selected samples)", vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs,
                          n_points, samples_per_ray, out_index, false, colour_products, normals, colors, stream, n_dev);
}

// ------------------------------------------------------------------------------------------------
// split inference launches: the VF net once per distinct sample, the rendering net on gathered feature blocks
// ------------------------------------------------------------------------------------------------
extern "C" int vfn_vf_feat16_fwd(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                                 float* out_vec, void* out_blocks, void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32; Plan16 vf;
    int rc = make_plan16(VFN_NET_VF, geom, &p32, &vf, "vfn_vf_feat16_fwd");
    if (rc != VFN_OK) return rc;
    rc = check_vf16(vf, "vfn_vf_feat16_fwd");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(packed16 && points && out_vec && out_blocks, "vfn_vf_feat16_fwd: NULL argument");
    VFN_REQUIRE(n_points < (1ll << 22), "vfn_vf_feat16_fwd: at most 4194303 points per launch (32-bit block offsets)");
    a.vf_w = (const uint4*)packed16; a.points = points; a.out_vec = out_vec; a.n_points = n_points; a.dirs_div = 1;
    a.vf_multires = vf.multires; a.vf_bytes = vf.total_kb * 1024u; a.blk_out = (uint4*)out_blocks;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_BLK>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_feat16_fwd");
}

extern "C" int vfn_render16_from_blocks(const vfn_net_geom* rn_geom, const void* rn_packed16, const void* blocks, const float* vecs,
                                        const int32_t* dst, const float* points, const float* ray_dirs, int64_t n_points,
                                        int32_t samples_per_ray, float* normals, float* colors, void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32; Plan16 rn;
    int rc = make_plan16(VFN_NET_RENDER, rn_geom, &p32, &rn, "vfn_render16_from_blocks");
    if (rc != VFN_OK) return rc;
    rc = check_rn16(rn, "vfn_render16_from_blocks");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(rn_geom->feature_dims == VFN_HIDDEN, "vfn_render16_from_blocks: feature_dims must be %d", VFN_HIDDEN);
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(rn_packed16 && blocks && vecs && dst && points && ray_dirs && normals && colors, "vfn_render16_from_blocks: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_render16_from_blocks: samples_per_ray must be > 0");
    a.rn_w = (const uint4*)rn_packed16; a.vf_w = a.rn_w; a.points = points; a.ray_dirs = ray_dirs; a.out_vec = normals; a.out_colors = colors;
    a.n_points = n_points; a.dirs_div = samples_per_ray; a.rn_multires = rn.multires; a.rn_bytes = rn.total_kb * 1024u; a.vf_bytes = a.rn_bytes;
    a.blk_in = (const uint4*)blocks; a.vec_in = vecs; a.src = dst;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long nblocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    hipLaunchKernelGGL(vfn_mlp16_kernel<M16_RN_BLK>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_render16_from_blocks");
}

// ------------------------------------------------------------------------------------------------
// training forwards: same arithmetic, plus the workspace the backward kernels read (include/vfn.h, "slots")
// ------------------------------------------------------------------------------------------------
extern "C" int vfn_vf_mlp16_fwd_train(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                                      int32_t with_features, float* out_vec, float* saved, float* save_aux_vf, uint32_t* save_masks,
                                      int32_t save_f16, void* stream) {
    return vfn_vf_mlp16_fwd_train_at(geom, packed16, points, n_points, with_features, out_vec, saved, save_aux_vf, save_masks, save_f16, 0, n_points,
                                     stream);
}

extern "C" int vfn_vf_mlp16_fwd_train_at(const vfn_net_geom* geom, const void* packed16, const float* points, int64_t n_points,
                                         int32_t with_features, float* out_vec, float* saved, float* save_aux_vf, uint32_t* save_masks,
                                         int32_t save_f16, int64_t ws_first, int64_t ws_points, void* stream) {
    Mlp16Args a = {};
    VfnNetPlan p32; Plan16 vf;
    int rc = make_plan16(VFN_NET_VF, geom, &p32, &vf, "vfn_vf_mlp16_fwd_train");
    if (rc != VFN_OK) return rc;
    rc = check_vf16(vf, "vfn_vf_mlp16_fwd_train");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(packed16 && points && out_vec && saved && save_aux_vf && save_masks, "vfn_vf_mlp16_fwd_train: NULL argument");
    VFN_REQUIRE(ws_first >= 0 && ws_first + n_points <= ws_points, "vfn_vf_mlp16_fwd_train: points %lld .. %lld outside a workspace of %lld",
                (long long)ws_first, (long long)(ws_first + n_points), (long long)ws_points);
    VFN_REQUIRE(!(save_f16 & 2) || ws_first % 32 == 0, "vfn_vf_mlp16_fwd_train: ws_first must be a multiple of 32 in fragment order");
    VFN_REQUIRE(n_points < (1ll << 21) && ws_points < (1ll << 26), "vfn_vf_mlp16_fwd_train: at most 2097151 points per launch (32-bit slot offsets) and 67108863 per "
                "workspace (got %lld in %lld)", (long long)n_points, (long long)ws_points);
    a.vf_w = (const uint4*)packed16; a.points = points; a.out_vec = out_vec; a.n_points = n_points; a.dirs_div = 1;
    a.vf_multires = vf.multires; a.vf_bytes = vf.total_kb * 1024u; a.saved = saved; a.save_aux_vf = save_aux_vf;
    a.save_masks = save_masks; a.save_f16 = save_f16 & 3;
    a.ws_first = ws_first; a.ws_points = ws_points;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    if (save_f16 & 4) {        // single-product arithmetic (opt-in 16-bit-native training, vector columns only)
        VFN_REQUIRE(!with_features, "vfn_vf_mlp16_fwd_train: the single-product forward computes the vector columns only");
        hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_VEC_TRAIN | M16_P1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    else if (with_features && (save_f16 & 3) == 3) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_FULL_TRAIN | M16_SF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if (with_features) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_FULL_TRAIN>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if ((save_f16 & 3) == 3) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_VEC_TRAIN | M16_SF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);   // the default storages
    else hipLaunchKernelGGL(vfn_mlp16_kernel<M16_VF_VEC_TRAIN>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_mlp16_fwd_train");
}

extern "C" int vfn_vf_render_fused16_fwd_train(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                               const void* rn_packed16, const float* points, const float* ray_dirs,
                                               int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                               float* saved, float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks,
                                               int32_t save_f16, void* stream) {
    return vfn_vf_render_fused16_fwd_train_at(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, n_points, samples_per_ray, normals, colors,
                                              saved, save_aux_vf, save_aux_rn, save_masks, save_f16, 0, n_points, 3, stream);
}

extern "C" int vfn_vf_render_fused16_fwd_train_at(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                                  const void* rn_packed16, const float* points, const float* ray_dirs,
                                                  int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                                  float* saved, float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks,
                                                  int32_t save_f16, int64_t ws_first, int64_t ws_points, int32_t colour_products,
                                                  void* stream) {
    return vfn_internal_fused16_fwd_train_at(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, n_points, nullptr, samples_per_ray, normals,
                                             colors, saved, save_aux_vf, save_aux_rn, save_masks, save_f16, ws_first, ws_points, colour_products, stream);
}

// ... over min(n_points, *n_dev) points when n_dev (a device pointer) is given: n_points is then the CAPACITY the launch is sized for
int vfn_internal_fused16_fwd_train_at(const vfn_net_geom* vf_geom, const void* vf_packed16, const vfn_net_geom* rn_geom,
                                      const void* rn_packed16, const float* points, const float* ray_dirs, int64_t n_points,
                                      const int32_t* n_dev, int32_t samples_per_ray, float* normals, float* colors, float* saved,
                                      float* save_aux_vf, float* save_aux_rn, uint32_t* save_masks, int32_t save_f16, int64_t ws_first,
                                      int64_t ws_points, int32_t colour_products, void* stream) {
    Mlp16Args a = {};
    a.n_dev = n_dev;
    VfnNetPlan p32; Plan16 vf, rn;
    int rc = make_plan16(VFN_NET_VF, vf_geom, &p32, &vf, "vfn_vf_render_fused16_fwd_train");
    if (rc != VFN_OK) return rc;
    rc = make_plan16(VFN_NET_RENDER, rn_geom, &p32, &rn, "vfn_vf_render_fused16_fwd_train");
    if (rc != VFN_OK) return rc;
    rc = check_vf16(vf, "vfn_vf_render_fused16_fwd_train");
    if (rc != VFN_OK) return rc;
    rc = check_rn16(rn, "vfn_vf_render_fused16_fwd_train");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed16 && rn_packed16 && points && ray_dirs && normals && colors && saved && save_aux_vf && save_aux_rn && save_masks,
                "vfn_vf_render_fused16_fwd_train: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_vf_render_fused16_fwd_train: samples_per_ray must be > 0");
    VFN_REQUIRE(ws_first >= 0 && ws_first + n_points <= ws_points, "vfn_vf_render_fused16_fwd_train: points %lld .. %lld outside a workspace of %lld",
                (long long)ws_first, (long long)(ws_first + n_points), (long long)ws_points);
    VFN_REQUIRE(!(save_f16 & 2) || ws_first % 32 == 0, "vfn_vf_render_fused16_fwd_train: ws_first must be a multiple of 32 in fragment order");
    VFN_REQUIRE(n_points < (1ll << 21) && ws_points < (1ll << 26), "vfn_vf_render_fused16_fwd_train: at most 2097151 points per launch (32-bit slot offsets) and 67108863 per "
                "workspace (got %lld in %lld)", (long long)n_points, (long long)ws_points);
    a.ws_first = ws_first; a.ws_points = ws_points;
    a.vf_w = (const uint4*)vf_packed16; a.rn_w = (const uint4*)rn_packed16; a.points = points; a.ray_dirs = ray_dirs;
    a.out_vec = normals; a.out_colors = colors; a.n_points = n_points; a.dirs_div = samples_per_ray;
    a.vf_multires = vf.multires; a.rn_multires = rn.multires; a.vf_bytes = vf.total_kb * 1024u; a.rn_bytes = rn.total_kb * 1024u;
    a.saved = saved; a.save_aux_vf = save_aux_vf; a.save_aux_rn = save_aux_rn; a.save_masks = save_masks; a.save_f16 = save_f16 & 3;
    a.status = t_status_word;
    a.clock = t_clock_probe; a.clock_slots = t_clock_slots;
    const long long blocks = (n_points + VFN16_PTS - 1) / VFN16_PTS;
    VFN_REQUIRE(colour_products >= 1 && colour_products <= 3, "vfn_vf_render_fused16_fwd_train: colour_products must be 1, 2 or 3 (got %d)", colour_products);
    if (colour_products == 1) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED_TRAIN | M16_P1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);   // one product EVERYWHERE
    else if (colour_products == 2) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED_TRAIN | M16_C2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if ((save_f16 & 3) == 3) hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED_TRAIN | M16_SF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(vfn_mlp16_kernel<M16_FUSED_TRAIN>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_vf_render_fused16_fwd_train");
}
