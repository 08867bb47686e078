// vfn_mlp.hip — fused fp32-MFMA forward kernels for the two MLPs of the VF-NeRF hot path.
//
// Replaces models/vector_field/vector_field_network.py:177-208 (eval-mode _forward),
// models/vector_field/rendering_network.py:62-108 and models/helpers/embedder.py:11-37, i.e. the
// GEMM chains #1-#3 of VectorFieldNerf.render (models/nerf/vector_field_nerf.py:253,294,315).
//
// Design (CDNA4 / gfx950, wave64):
//  * one workgroup = 4 waves = 64 points; the whole layer chain runs in that workgroup with the
//    64x256 fp32 activation tile resident in LDS (64 KiB, XOR-swizzled 16-byte chunks) — no
//    activation ever goes to HBM between layers; two workgroups fit per CU (2 x 76 KiB LDS,
//    <= 256 VGPRs) so one workgroup's epilogue/barrier overlaps the other's matrix work;
//  * waves split the OUTPUT columns (2 tiles of 32 each) and share the A operand, so every weight
//    element is fetched exactly once per workgroup, straight from L2 to VGPRs as a coalesced
//    16-byte-per-lane stream in the fragment order produced by vfn_pack_weights (no LDS staging);
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD) for the 256-wide layers, 16 MFMAs per
//    pair of 16-byte A/B fragments; v_mfma_f32_16x16x4_f32 for the 3-channel heads;
//  * positional encodings are computed once per point into a small LDS "aux" tile that is consumed
//    as extra K blocks (layer 0, the skip layer, the rendering net's first layer);
//  * BatchNorm (eval) and the skip 1/sqrt(2) are folded into the packed weights; bias initialises
//    the accumulators; ReLU / tanh / sigmoid run in the epilogue on the accumulator registers.
#include <string.h>
#include "vfn_common.h"
#include "vfn_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = VFN_TM;       // 64 rows per workgroup
constexpr int ACT_LD = 256;      // floats per activation row
constexpr int AUX_LD = 44;       // floats per aux row (40 used; 44 keeps ds_read_b128 conflict-free)
constexpr int NTHREADS = 256;
constexpr int SMEM_FLOATS = TM * ACT_LD + TM * AUX_LD + TM * 3 + TM * 3;

enum : int { MODE_VF_VEC = 0, MODE_VF_FULL = 1, MODE_FUSED = 2, MODE_RENDER = 3 };
enum : int { ACT_RELU = 0, ACT_TANH = 1 };

struct MlpArgs {
    VfnNetPlan vf;
    VfnNetPlan rn;
    const float* vf_w;
    const float* rn_w;
    const float* points;     // [M,3]
    const float* view_dirs;  // [M / dirs_div, 3]
    const float* normals_in; // MODE_RENDER: [M,3]
    const float* feats_in;   // MODE_RENDER: [M,256]
    float* out_vec;          // normals [M,3] or full VF row [M, out_stride]
    float* out_colors;       // [M,3]
    float* out_feats;        // optional [M,256]
    long long n_points;
    int dirs_div;
    int out_stride;
};

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

__device__ __forceinline__ float act_fn(float v, int kind) { return kind == ACT_RELU ? fmaxf(v, 0.f) : tanhf(v); }

// Store this wave's accumulators into the activation tile (after the workgroup has finished
// reading it), applying the activation.  D layout of v_mfma_f32_32x32x2_f32:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <int NT>
__device__ __forceinline__ void store_tile_lds(const f32x16 (&acc)[2][2], float* s_act, int tile0, int lane, int kind) {
    const int c = lane & 31, h = lane >> 5;
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
    const int c = lane & 31, h = lane >> 5;
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

// One hidden layer: matrix work, barrier, epilogue into the act tile, barrier.
__device__ __forceinline__ void hidden_layer(const VfnLayerPlan& lp, const float* __restrict__ wbase, float* s_act,
                                             const float* s_aux, int wave, int lane, int kind) {
    const int tile0 = 2 * wave;
    const int nt = min(2, max(0, (int)lp.n_tiles - tile0));  // wave-uniform
    f32x16 acc[2][2];
    if (nt == 2) layer_mma<2>(acc, lp, wbase, s_act, s_aux, tile0, lane);
    else if (nt == 1) layer_mma<1>(acc, lp, wbase, s_act, s_aux, tile0, lane);
    __syncthreads();
    if (nt == 2) store_tile_lds<2>(acc, s_act, tile0, lane, kind);
    else if (nt == 1) store_tile_lds<1>(acc, s_act, tile0, lane, kind);
    __syncthreads();
}

// aux <- [x, sin(2^k x), cos(2^k x)]_{k<L} for the 3-vector at src[row*3..], columns col0..col0+3+6L-1.
// Four threads per row share the 3L (sin,cos) pairs.
__device__ __forceinline__ void fill_encoding(float* s_aux, const float* src3, int row, int part, int col0, int multires) {
    float x[3] = {src3[0], src3[1], src3[2]};
    if (part == 0) {
        s_aux[row * AUX_LD + col0 + 0] = x[0];
        s_aux[row * AUX_LD + col0 + 1] = x[1];
        s_aux[row * AUX_LD + col0 + 2] = x[2];
    }
    for (int q = part; q < 3 * multires; q += 4) {
        const int k = q / 3, c = q - 3 * k;
        const float arg = x[c] * (float)(1 << k);
        float s, cs;
        sincosf(arg, &s, &cs);
        s_aux[row * AUX_LD + col0 + 3 + 6 * k + c] = s;
        s_aux[row * AUX_LD + col0 + 3 + 6 * k + 3 + c] = cs;
    }
}

template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void vfn_mlp_kernel(const MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* s_act = smem;
    float* s_aux = smem + TM * ACT_LD;
    float* s_pts = s_aux + TM * AUX_LD;   // [64][3] points
    float* s_nrm = s_pts + TM * 3;        // [64][3] normals (fused / render)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long row0 = (long long)blockIdx.x * TM;
    const long long n_rows = a.n_points;

    // ---- stage the 64 points -------------------------------------------------------------
    if (tid < TM * 3) {
        const long long g = row0 * 3 + tid;
        s_pts[tid] = (g < n_rows * 3) ? a.points[g] : 0.f;
    }
    if (MODE == MODE_RENDER) {
        if (tid >= 64 && tid < 64 + TM * 3) {
            const long long g = row0 * 3 + (tid - 64);
            s_nrm[tid - 64] = (g < n_rows * 3) ? a.normals_in[g] : 0.f;
        }
        // features -> act tile (coalesced 16-byte loads, swizzled 16-byte stores)
        for (int i = tid; i < TM * (ACT_LD / 4); i += NTHREADS) {
            const int row = i >> 6, ch = i & 63;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + row < n_rows) v = *reinterpret_cast<const f32x4*>(a.feats_in + (row0 + row) * ACT_LD + ch * 4);
            *reinterpret_cast<f32x4*>(s_act + row * ACT_LD + ((ch ^ (row & 15)) << 2)) = v;
        }
    }
    __syncthreads();

    const int prow = tid >> 2, ppart = tid & 3;

    if (MODE != MODE_RENDER) {
        // ---- VF net: positional encoding of the point into the aux tile -------------------
        const VfnNetPlan& vf = a.vf;
        fill_encoding(s_aux, s_pts + prow * 3, prow, ppart, 0, vf.multires);
        if (ppart == 1)
            for (int c = vf.pe_dim; c < VFN_AUX_K; ++c) s_aux[prow * AUX_LD + c] = 0.f;
        __syncthreads();

        const int n_plain = vf.n_hidden - vf.feat_layer;
        for (int l = 0; l < n_plain; ++l) hidden_layer(vf.hidden[l], a.vf_w, s_act, s_aux, wave, lane, ACT_RELU);

        // ---- last Linear: 3-channel head (+ feature block) on the same input --------------
        f32x4 hd = head_mma(vf, a.vf_w, s_act, wave, lane);
        const int ch = lane & 15, q = lane >> 4;
        if (MODE == MODE_VF_VEC) {
            if (ch < 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long row = row0 + 16 * wave + 4 * q + r;
                    if (row < n_rows) a.out_vec[row * a.out_stride + ch] = tanhf(hd[r]);
                }
            }
            return;
        }
        // feature block (tanh), kept in registers until every wave has finished reading the tile
        const VfnLayerPlan& fl = vf.hidden[vf.n_hidden - 1];
        f32x16 acc[2][2];
        const int tile0 = 2 * wave;
        layer_mma<2>(acc, fl, a.vf_w, s_act, s_aux, tile0, lane);
        if (ch < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = 16 * wave + 4 * q + r;
                const float v = tanhf(hd[r]);
                if (row0 + lrow < n_rows) a.out_vec[(row0 + lrow) * a.out_stride + ch] = v;
                if (MODE == MODE_FUSED) s_nrm[lrow * 3 + ch] = v;
            }
        }
        if (MODE == MODE_VF_FULL) {
            store_tile_global<2>(acc, a.out_vec, row0, n_rows, a.out_stride, 3, tile0, lane, ACT_TANH);
            return;
        }
        if (a.out_feats) store_tile_global<2>(acc, a.out_feats, row0, n_rows, ACT_LD, 0, tile0, lane, ACT_TANH);
        __syncthreads();
        store_tile_lds<2>(acc, s_act, tile0, lane, ACT_TANH);
        // (the barrier that publishes the features is the one after the aux refill below)
    }

    // ---- rendering net: aux = [p(3), PE(d)(3+6L), n(3), 0...] ------------------------------
    {
        const VfnNetPlan& rn = a.rn;
        const long long grow = row0 + prow;
        float d[3] = {0.f, 0.f, 0.f};
        if (grow < n_rows) {
            const long long di = grow / a.dirs_div;
            d[0] = a.view_dirs[di * 3 + 0]; d[1] = a.view_dirs[di * 3 + 1]; d[2] = a.view_dirs[di * 3 + 2];
        }
        fill_encoding(s_aux, d, prow, ppart, 3, rn.multires);
        const int ncol = 3 + rn.pe_dim;  // first normal column
        if (ppart == 1) {
            s_aux[prow * AUX_LD + 0] = s_pts[prow * 3 + 0];
            s_aux[prow * AUX_LD + 1] = s_pts[prow * 3 + 1];
            s_aux[prow * AUX_LD + 2] = s_pts[prow * 3 + 2];
        }
        if (ppart == 2)
            for (int c = ncol + 3; c < VFN_AUX_K; ++c) s_aux[prow * AUX_LD + c] = 0.f;
        __syncthreads();  // s_nrm (head lanes of all waves) and features are now visible
        if (ppart == 3) {
            s_aux[prow * AUX_LD + ncol + 0] = s_nrm[prow * 3 + 0];
            s_aux[prow * AUX_LD + ncol + 1] = s_nrm[prow * 3 + 1];
            s_aux[prow * AUX_LD + ncol + 2] = s_nrm[prow * 3 + 2];
        }
        __syncthreads();

        for (int l = 0; l < rn.n_hidden; ++l) hidden_layer(rn.hidden[l], a.rn_w, s_act, s_aux, wave, lane, ACT_RELU);

        f32x4 hd = head_mma(rn, a.rn_w, s_act, wave, lane);
        const int ch = lane & 15, q = lane >> 4;
        if (ch < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = row0 + 16 * wave + 4 * q + r;
                if (row < n_rows) a.out_colors[row * 3 + ch] = 1.0f / (1.0f + expf(-hd[r]));
            }
        }
    }
}

template <int MODE>
int launch(const MlpArgs& a, hipStream_t stream, const char* what) {
    if (a.n_points <= 0) return VFN_OK;
    const long long blocks = (a.n_points + TM - 1) / TM;
    if (blocks > 0x7fffffffLL) { vfn_set_error("%s: too many points", what); return VFN_ERR_INVALID; }
    hipLaunchKernelGGL(vfn_mlp_kernel<MODE>, dim3((unsigned)blocks), dim3(NTHREADS), 0, stream, a);
    return vfn_check_launch(what);
}

int plan_or_error(int kind, const vfn_net_geom* g, VfnNetPlan* p, const char* what) {
    char err[256] = {0};
    int rc = vfn_make_plan(kind, g, p, err, sizeof(err));
    if (rc != VFN_OK) vfn_set_error("%s: %s", what, err);
    return rc;
}

}  // namespace

extern "C" int vfn_vf_mlp_fwd(const vfn_net_geom* geom, const float* packed, const float* points, int64_t n_points,
                              int32_t out_cols, float* out, void* stream) {
    MlpArgs a = {};
    int rc = plan_or_error(VFN_NET_VF, geom, &a.vf, "vfn_vf_mlp_fwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(packed && points && out, "vfn_vf_mlp_fwd: NULL argument");
    VFN_REQUIRE(out_cols == 3 || (geom->feature_dims > 0 && out_cols == 3 + geom->feature_dims),
                "vfn_vf_mlp_fwd: out_cols=%d must be 3 or 3+feature_dims", out_cols);
    a.vf_w = packed; a.points = points; a.out_vec = out; a.n_points = n_points; a.out_stride = out_cols; a.dirs_div = 1;
    if (out_cols == 3) return launch<MODE_VF_VEC>(a, (hipStream_t)stream, "vfn_vf_mlp_fwd");
    return launch<MODE_VF_FULL>(a, (hipStream_t)stream, "vfn_vf_mlp_fwd");
}

extern "C" int vfn_render_mlp_fwd(const vfn_net_geom* geom, const float* packed, const float* points,
                                  const float* normals, const float* view_dirs, const float* feats, int64_t n_points,
                                  float* colors, void* stream) {
    MlpArgs a = {};
    int rc = plan_or_error(VFN_NET_RENDER, geom, &a.rn, "vfn_render_mlp_fwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(geom->feature_dims == VFN_HIDDEN, "vfn_render_mlp_fwd: feature_dims must be %d", VFN_HIDDEN);
    VFN_REQUIRE(packed && points && normals && view_dirs && feats && colors, "vfn_render_mlp_fwd: NULL argument");
    a.rn_w = packed; a.points = points; a.normals_in = normals; a.view_dirs = view_dirs; a.feats_in = feats;
    a.out_colors = colors; a.n_points = n_points; a.dirs_div = 1;
    return launch<MODE_RENDER>(a, (hipStream_t)stream, "vfn_render_mlp_fwd");
}

extern "C" int vfn_vf_render_fused_fwd(const vfn_net_geom* vf_geom, const float* vf_packed, const vfn_net_geom* rn_geom,
                                       const float* rn_packed, const float* points, const float* ray_dirs,
                                       int64_t n_points, int32_t samples_per_ray, float* normals, float* colors,
                                       float* feats_out, void* stream) {
    MlpArgs a = {};
    int rc = plan_or_error(VFN_NET_VF, vf_geom, &a.vf, "vfn_vf_render_fused_fwd");
    if (rc != VFN_OK) return rc;
    rc = plan_or_error(VFN_NET_RENDER, rn_geom, &a.rn, "vfn_vf_render_fused_fwd");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(vf_geom->feature_dims == VFN_HIDDEN && rn_geom->feature_dims == VFN_HIDDEN,
                "vfn_vf_render_fused_fwd: both nets need feature_dims == %d", VFN_HIDDEN);
    VFN_REQUIRE(vf_packed && rn_packed && points && ray_dirs && normals && colors, "vfn_vf_render_fused_fwd: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_vf_render_fused_fwd: samples_per_ray must be > 0");
    a.vf_w = vf_packed; a.rn_w = rn_packed; a.points = points; a.view_dirs = ray_dirs; a.out_vec = normals;
    a.out_colors = colors; a.out_feats = feats_out; a.n_points = n_points; a.dirs_div = samples_per_ray; a.out_stride = 3;
    return launch<MODE_FUSED>(a, (hipStream_t)stream, "vfn_vf_render_fused_fwd");
}
