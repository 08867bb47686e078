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
#include "vfn_mlp_core.h"
using namespace vfn;

namespace {

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
    // training: post-activation output of every hidden layer -> save_act[slot][M][256] (slots: VF hidden
    // 0..n-1 incl. the feature block, then rendering hidden 0..n-1), aux tiles -> save_aux_*[M][40]
    float* save_act;
    float* save_aux_vf;
    float* save_aux_rn;
};

// One hidden layer: matrix work, barrier, epilogue into the act tile, barrier.
__device__ __forceinline__ void hidden_layer(const VfnLayerPlan& lp, const float* __restrict__ wbase, float* s_act,
                                             const float* s_aux, int wave, int lane, int kind, float* save,
                                             long long row0, long long n_rows) {
    const int tile0 = 2 * wave;
    const int nt = min(2, max(0, (int)lp.n_tiles - tile0));  // wave-uniform
    f32x16 acc[2][2];
    if (nt == 2) layer_mma<2>(acc, lp, wbase, s_act, s_aux, tile0, lane);
    else if (nt == 1) layer_mma<1>(acc, lp, wbase, s_act, s_aux, tile0, lane);
    if (save) {
        if (nt == 2) store_tile_global<2>(acc, save, row0, n_rows, ACT_LD, 0, tile0, lane, kind);
        else if (nt == 1) store_tile_global<1>(acc, save, row0, n_rows, ACT_LD, 0, tile0, lane, kind);
    }
    __syncthreads();
    if (nt == 2) store_tile_lds<2>(acc, s_act, tile0, lane, kind);
    else if (nt == 1) store_tile_lds<1>(acc, s_act, tile0, lane, kind);
    __syncthreads();
}

// aux tile (first VFN_AUX_K columns) -> global [M][40] for the weight-gradient kernels
__device__ __forceinline__ void save_aux_tile(const float* s_aux, float* dst, long long row0, long long n_rows, int tid) {
    const int row = tid >> 2, part = tid & 3;
    if (row0 + row < n_rows)
        for (int c = part * 10; c < part * 10 + 10; ++c) dst[(row0 + row) * VFN_AUX_K + c] = s_aux[row * AUX_LD + c];
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
        if (a.save_aux_vf) save_aux_tile(s_aux, a.save_aux_vf, row0, n_rows, tid);
        const long long slot = n_rows * ACT_LD;

        const int n_plain = vf.n_hidden - vf.feat_layer;
        for (int l = 0; l < n_plain; ++l)
            hidden_layer(vf.hidden[l], a.vf_w, s_act, s_aux, wave, lane, ACT_RELU,
                         a.save_act ? a.save_act + l * slot : nullptr, row0, n_rows);

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
            if (a.save_act)
                store_tile_global<2>(acc, a.save_act + (vf.n_hidden - 1) * slot, row0, n_rows, ACT_LD, 0, tile0, lane, ACT_TANH);
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
        if (a.save_aux_rn) save_aux_tile(s_aux, a.save_aux_rn, row0, n_rows, tid);
        const long long slot = n_rows * ACT_LD;
        const int slot0 = (MODE == MODE_RENDER) ? 0 : a.vf.n_hidden;

        for (int l = 0; l < rn.n_hidden; ++l)
            hidden_layer(rn.hidden[l], a.rn_w, s_act, s_aux, wave, lane, ACT_RELU,
                         a.save_act ? a.save_act + (slot0 + l) * slot : nullptr, row0, n_rows);

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
    if (n_points <= 0) return VFN_OK;
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
    if (n_points <= 0) return VFN_OK;
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
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed && rn_packed && points && ray_dirs && normals && colors, "vfn_vf_render_fused_fwd: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_vf_render_fused_fwd: samples_per_ray must be > 0");
    a.vf_w = vf_packed; a.rn_w = rn_packed; a.points = points; a.view_dirs = ray_dirs; a.out_vec = normals;
    a.out_colors = colors; a.out_feats = feats_out; a.n_points = n_points; a.dirs_div = samples_per_ray; a.out_stride = 3;
    return launch<MODE_FUSED>(a, (hipStream_t)stream, "vfn_vf_render_fused_fwd");
}

// ---- training variants: same kernels, additionally saving what the backward pass needs ---------------
extern "C" int vfn_vf_mlp_fwd_train(const vfn_net_geom* geom, const float* packed, const float* points, int64_t n_points,
                                    int32_t out_cols, float* out, float* save_act, float* save_aux_vf, void* stream) {
    MlpArgs a = {};
    int rc = plan_or_error(VFN_NET_VF, geom, &a.vf, "vfn_vf_mlp_fwd_train");
    if (rc != VFN_OK) return rc;
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(packed && points && out && save_act && save_aux_vf, "vfn_vf_mlp_fwd_train: NULL argument");
    VFN_REQUIRE(out_cols == 3 || (geom->feature_dims > 0 && out_cols == 3 + geom->feature_dims),
                "vfn_vf_mlp_fwd_train: out_cols=%d must be 3 or 3+feature_dims", out_cols);
    a.vf_w = packed; a.points = points; a.out_vec = out; a.n_points = n_points; a.out_stride = out_cols; a.dirs_div = 1;
    a.save_act = save_act; a.save_aux_vf = save_aux_vf;
    if (out_cols == 3) return launch<MODE_VF_VEC>(a, (hipStream_t)stream, "vfn_vf_mlp_fwd_train");
    return launch<MODE_VF_FULL>(a, (hipStream_t)stream, "vfn_vf_mlp_fwd_train");
}

extern "C" int vfn_vf_render_fused_fwd_train(const vfn_net_geom* vf_geom, const float* vf_packed,
                                             const vfn_net_geom* rn_geom, const float* rn_packed, const float* points,
                                             const float* ray_dirs, int64_t n_points, int32_t samples_per_ray,
                                             float* normals, float* colors, float* save_act, float* save_aux_vf,
                                             float* save_aux_rn, void* stream) {
    MlpArgs a = {};
    int rc = plan_or_error(VFN_NET_VF, vf_geom, &a.vf, "vfn_vf_render_fused_fwd_train");
    if (rc != VFN_OK) return rc;
    rc = plan_or_error(VFN_NET_RENDER, rn_geom, &a.rn, "vfn_vf_render_fused_fwd_train");
    if (rc != VFN_OK) return rc;
    VFN_REQUIRE(vf_geom->feature_dims == VFN_HIDDEN && rn_geom->feature_dims == VFN_HIDDEN,
                "vfn_vf_render_fused_fwd_train: both nets need feature_dims == %d", VFN_HIDDEN);
    if (n_points <= 0) return VFN_OK;
    VFN_REQUIRE(vf_packed && rn_packed && points && ray_dirs && normals && colors && save_act && save_aux_vf && save_aux_rn,
                "vfn_vf_render_fused_fwd_train: NULL argument");
    VFN_REQUIRE(samples_per_ray > 0, "vfn_vf_render_fused_fwd_train: samples_per_ray must be > 0");
    a.vf_w = vf_packed; a.rn_w = rn_packed; a.points = points; a.view_dirs = ray_dirs; a.out_vec = normals;
    a.out_colors = colors; a.n_points = n_points; a.dirs_div = samples_per_ray; a.out_stride = 3;
    a.save_act = save_act; a.save_aux_vf = save_aux_vf; a.save_aux_rn = save_aux_rn;
    a.out_feats = save_act + (size_t)(a.vf.n_hidden - 1) * n_points * ACT_LD;   // feature slot
    return launch<MODE_FUSED>(a, (hipStream_t)stream, "vfn_vf_render_fused_fwd_train");
}
