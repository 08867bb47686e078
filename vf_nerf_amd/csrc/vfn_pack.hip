// vfn_pack.hip — layer planning, BatchNorm folding and MFMA-fragment weight packing.
//
// Replaces the per-forward parameter reads of the reference's nn.Linear / nn.BatchNorm1d stacks
// (models/vector_field/vector_field_network.py:47-60,177-208; rendering_network.py:46-53,98-103):
// the fused kernels stream weights in exactly the order the matrix cores consume them, so the
// live nn.Parameter storage is re-packed once per optimizer step (3.2 MB, a few microseconds).
#include <string.h>
#include <string>
#include "vfn_common.h"
#include "vfn_plan.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void vfn_set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char* vfn_last_error(void) { return g_last_error.c_str(); }
extern "C" int vfn_abi_version(void) { return VFN_ABI_VERSION; }
extern "C" int32_t vfn_abi_struct_bytes(int32_t which) {
    switch (which) {
    case 0: return (int32_t)sizeof(vfn_net_geom);
    case 1: return (int32_t)sizeof(vfn_layer_params);
    case 2: return (int32_t)sizeof(vfn_raygen_params);
    case 3: return (int32_t)sizeof(vfn_density_params);
    case 4: return (int32_t)sizeof(vfn_fine_params);
    case 5: return (int32_t)sizeof(vfn_render_params);
    case 6: return (int32_t)sizeof(vfn_unfold_entry);
    case 7: return (int32_t)sizeof(vfn_wgrad_layer);
    case 8: return (int32_t)sizeof(vfn_loss_params);
    case 9: return (int32_t)sizeof(vfn_train_step_params);
    case 10: return (int32_t)sizeof(vfn_train_step_io);
    default: return -1;
    }
}

// ------------------------------------------------------------------------------------------------
// planning
// ------------------------------------------------------------------------------------------------
static int plan_fail(char* err, int errlen, const char* fmt, ...) {
    if (err && errlen > 0) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return VFN_ERR_UNSUPPORTED;
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int vfn_make_plan(int net_kind, const vfn_net_geom* g, VfnNetPlan* plan, char* err, int errlen) {
    memset(plan, 0, sizeof(*plan));
    if (!g) return plan_fail(err, errlen, "geometry is NULL");
    const int L = g->n_layers;
    if (L < 2 || L > VFN_MAX_LAYERS) return plan_fail(err, errlen, "n_layers=%d outside [2,%d]", L, VFN_MAX_LAYERS);
    if (g->multires < 0 || g->multires > 6)
        return plan_fail(err, errlen, "multires=%d unsupported (aux tile holds <= %d columns)", g->multires, VFN_AUX_K);
    const int pe_dim = g->multires > 0 ? 3 + 6 * g->multires : 3;
    const int F = g->feature_dims;
    if (F != 0 && F != VFN_HIDDEN) return plan_fail(err, errlen, "feature_dims=%d must be 0 or %d", F, VFN_HIDDEN);
    plan->multires = g->multires;
    plan->pe_dim = pe_dim;

    uint32_t off = 0, boff = 0;
    int nh = 0;
    auto add_hidden = [&](int ref_layer, uint32_t nkb_act, uint32_t nkb_aux, uint32_t n_tiles) {
        VfnLayerPlan& lp = plan->hidden[nh++];
        lp.bw_off = VFN_NO_BWD;
        if (nkb_act > 0) {  // act inputs carry gradient; aux inputs (encodings of points / dirs) never do
            lp.bw_off = boff;
            boff += (nkb_act / 4) * (4 * n_tiles) * 256u;
        }
        lp.ref_layer = (uint16_t)ref_layer;
        lp.nkb_act = (uint16_t)nkb_act;
        lp.nkb_aux = (uint16_t)nkb_aux;
        lp.n_tiles = (uint16_t)n_tiles;
        lp.w_off = off;
        off += n_tiles * (nkb_act + nkb_aux) * 256u;
        lp.b_off = off;
        off += n_tiles * 32u;
    };

    if (net_kind == VFN_NET_VF) {
        const int skip = g->skip_layer;
        if (skip == 0 || skip >= L - 1)
            return plan_fail(err, errlen, "skip_layer=%d must be inside the hidden stack (1..%d) or -1", skip, L - 2);
        for (int i = 0; i < L - 1; ++i) {
            const int expect_in = (i == 0) ? pe_dim : VFN_HIDDEN;
            if (g->in_dims[i] != expect_in)
                return plan_fail(err, errlen, "VF layer %d: in_features=%d, kernels need %d", i, g->in_dims[i], expect_in);
            const int expect_out = (i + 1 == skip) ? VFN_HIDDEN - pe_dim : VFN_HIDDEN;
            if (g->out_dims[i] != expect_out)
                return plan_fail(err, errlen, "VF layer %d: out_features=%d, kernels need %d", i, g->out_dims[i], expect_out);
            uint32_t nkb_act, nkb_aux;
            if (i == 0) { nkb_act = 0; nkb_aux = cdiv(pe_dim, 8); }
            else if (i == skip) { nkb_act = cdiv(g->out_dims[i - 1], 8); nkb_aux = cdiv(pe_dim, 8); }
            else { nkb_act = VFN_HIDDEN / 8; nkb_aux = 0; }
            if (nkb_act % 4) nkb_act += 4 - nkb_act % 4;  // act K is consumed/produced in 32-column tiles
            add_hidden(i, nkb_act, nkb_aux, cdiv(expect_out, 32));
        }
        if (g->in_dims[L - 1] != VFN_HIDDEN || g->out_dims[L - 1] != 3 + F)
            return plan_fail(err, errlen, "VF last layer %dx%d, kernels need %dx%d", g->out_dims[L - 1], g->in_dims[L - 1],
                             3 + F, VFN_HIDDEN);
        if (g->has_bn[L - 1]) return plan_fail(err, errlen, "BatchNorm on the last layer is not supported");
        if (F > 0) { add_hidden(L - 1, VFN_HIDDEN / 8, 0, F / 32); plan->feat_layer = 1; }
    } else if (net_kind == VFN_NET_RENDER) {
        const int aux_dim = 6 + pe_dim;  // p(3) ++ PE(d) ++ n(3)   (mode "idr")
        if (aux_dim > VFN_AUX_K) return plan_fail(err, errlen, "render aux width %d > %d", aux_dim, VFN_AUX_K);
        if (g->skip_layer >= 0) return plan_fail(err, errlen, "rendering net has no skip layer");
        if (g->in_dims[0] != aux_dim + F)
            return plan_fail(err, errlen, "render layer 0: in_features=%d, kernels need %d (mode idr)", g->in_dims[0], aux_dim + F);
        for (int i = 0; i < L - 1; ++i) {
            if (i > 0 && g->in_dims[i] != VFN_HIDDEN)
                return plan_fail(err, errlen, "render layer %d: in_features=%d, kernels need %d", i, g->in_dims[i], VFN_HIDDEN);
            if (g->out_dims[i] != VFN_HIDDEN)
                return plan_fail(err, errlen, "render layer %d: out_features=%d, kernels need %d", i, g->out_dims[i], VFN_HIDDEN);
            if (i == 0) add_hidden(0, F / 8, cdiv(aux_dim, 8), VFN_HIDDEN / 32);
            else add_hidden(i, VFN_HIDDEN / 8, 0, VFN_HIDDEN / 32);
        }
        if (g->in_dims[L - 1] != VFN_HIDDEN || g->out_dims[L - 1] != 3)
            return plan_fail(err, errlen, "render last layer %dx%d, kernels need 3x%d", g->out_dims[L - 1], g->in_dims[L - 1], VFN_HIDDEN);
        if (g->has_bn[L - 1]) return plan_fail(err, errlen, "BatchNorm on the last layer is not supported");
    } else {
        return plan_fail(err, errlen, "unknown net kind %d", net_kind);
    }
    plan->n_hidden = nh;
    plan->head_nkb16 = VFN_HIDDEN / 16;
    plan->head_w_off = off;
    off += plan->head_nkb16 * 256u;
    plan->head_b_off = off;
    off += 16u;
    plan->total_floats = off;
    plan->total_bwd_floats = boff;
    return VFN_OK;
}

extern "C" int64_t vfn_packed_size(int32_t net_kind, const vfn_net_geom* geom) {
    VfnNetPlan plan;
    char err[256] = {0};
    int rc = vfn_make_plan(net_kind, geom, &plan, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("vfn_packed_size: %s", err); return rc; }
    return (int64_t)plan.total_floats;
}

// ------------------------------------------------------------------------------------------------
// pack kernel
// ------------------------------------------------------------------------------------------------
struct PackEntry {
    const float* w;        // reference weight [out][in_dim]
    const float* b;        // reference bias [out]
    const float* bn_w;     // may be NULL
    const float* bn_b;
    const float* bn_mean;
    const float* bn_var;
    uint32_t w_off, b_off; // packed offsets
    uint32_t n_w;          // floats in the weight block
    uint32_t n_b;          // floats in the bias block
    uint32_t kb_total;     // K blocks per tile (8-wide; 16-wide for the head)
    uint32_t nkb_act;
    int32_t in_dim;        // reference row stride
    int32_t row_off;       // reference row of packed column 0
    int32_t n_rows;        // valid packed columns
    int32_t act_col_off, act_valid;
    int32_t aux_col_off, aux_valid;
    int32_t is_head;
    int32_t transposed;    // backward pack: tileT[kt][nb][lane][j] = W'[8nb+4h+j][32kt+(lane&31)]
    float scale;
};

struct PackArgs {
    PackEntry e[VFN_MAX_LAYERS + 2];
    int32_t n_entries;
    uint32_t total;
    float* out;
};

__global__ void vfn_pack_kernel(PackArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total) return;
    int ei = 0;
    for (int i = 1; i < a.n_entries; ++i)
        if (idx >= a.e[i].w_off) ei = i;
    const PackEntry& e = a.e[ei];
    uint32_t local = idx - e.w_off;
    float val = 0.f;
    if (local < e.n_w) {
        const uint32_t j = local & 3u;
        const uint32_t lane = (local >> 2) & 63u;
        const uint32_t blk = local >> 8;
        int n, kk;
        uint32_t kb;
        if (e.transposed) {
            const uint32_t nb = blk % e.kb_total, kt = blk / e.kb_total;
            n = (int)(8u * nb + 4u * (lane >> 5) + j);
            kk = (int)(32u * kt + (lane & 31u));
            kb = 0;
        } else if (e.is_head) {
            kb = blk;
            n = (int)(lane & 15u);
            kk = (int)(16u * kb + 4u * (lane >> 4) + j);
        } else {
            kb = blk % e.kb_total;
            const uint32_t nt = blk / e.kb_total;
            n = (int)(32u * nt + (lane & 31u));
            kk = (int)(8u * kb + 4u * (lane >> 5) + j);
        }
        int col = -1;
        const int act_k = e.transposed ? 0x7fffffff : (int)(e.nkb_act * (e.is_head ? 16u : 8u));
        if (kk < act_k) {
            if (kk < e.act_valid) col = e.act_col_off + kk;
        } else {
            const int k2 = kk - act_k;
            if (k2 < e.aux_valid) col = e.aux_col_off + k2;
        }
        if (n < e.n_rows && col >= 0) {
            const int row = e.row_off + n;
            float w = e.w[(size_t)row * e.in_dim + col];
            if (e.bn_w) w *= e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f);
            val = w * e.scale;
        }
    } else {
        const uint32_t n = local - e.n_w;  // bias block
        if ((int)n < e.n_rows) {
            const int row = e.row_off + (int)n;
            float b = e.b[row];
            if (e.bn_w) b = (b - e.bn_mean[row]) * (e.bn_w[row] / sqrtf(e.bn_var[row] + 1e-5f)) + e.bn_b[row];
            val = b;
        }
    }
    a.out[idx] = val;
}

// Builds the PackArgs entry list.  bwd == false: forward tiles + biases + head; bwd == true: transposed tiles.
static int build_pack_args(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                           const VfnNetPlan& plan, bool bwd, PackArgs& a) {
    memset(&a, 0, sizeof(a));
    const int L = geom->n_layers;
    const int pe_dim = plan.pe_dim;
    const int F = geom->feature_dims;

    auto fill_params = [&](PackEntry& e, int i) -> int {
        const vfn_layer_params& p = layers[i];
        VFN_REQUIRE(p.weight && p.bias, "vfn_pack_weights: layer %d has NULL weight/bias", i);
        e.w = p.weight; e.b = p.bias;
        if (geom->has_bn[i]) {
            VFN_REQUIRE(p.bn_weight && p.bn_bias && p.bn_mean && p.bn_var,
                        "vfn_pack_weights: layer %d is flagged has_bn but a BatchNorm pointer is NULL", i);
            e.bn_w = p.bn_weight; e.bn_b = p.bn_bias; e.bn_mean = p.bn_mean; e.bn_var = p.bn_var;
        }
        e.in_dim = geom->in_dims[i];
        e.scale = 1.0f;
        return VFN_OK;
    };

    for (int h = 0; h < plan.n_hidden; ++h) {
        const VfnLayerPlan& lp = plan.hidden[h];
        if (bwd && lp.bw_off == VFN_NO_BWD) continue;
        const int i = lp.ref_layer;
        PackEntry& e = a.e[a.n_entries++];
        int rc = fill_params(e, i);
        if (rc != VFN_OK) return rc;
        if (bwd) {
            e.w_off = lp.bw_off; e.b_off = 0;
            e.kb_total = 4u * lp.n_tiles;                  // nb blocks of 8 n's
            e.n_w = (lp.nkb_act / 4u) * e.kb_total * 256u;
            e.n_b = 0; e.transposed = 1;
        } else {
            e.w_off = lp.w_off; e.b_off = lp.b_off;
            e.kb_total = lp.nkb_act + lp.nkb_aux;
            e.n_w = lp.n_tiles * e.kb_total * 256u;
            e.n_b = lp.n_tiles * 32u;
        }
        e.nkb_act = lp.nkb_act;
        e.is_head = 0;
        const bool feat = plan.feat_layer && h == plan.n_hidden - 1;
        e.row_off = feat ? 3 : 0;
        e.n_rows = feat ? F : geom->out_dims[i];
        if (net_kind == VFN_NET_VF) {
            if (i == 0) { e.act_valid = 0; e.aux_col_off = 0; e.aux_valid = pe_dim; }
            else if (i == geom->skip_layer) {
                e.act_col_off = 0; e.act_valid = geom->out_dims[i - 1];
                e.aux_col_off = geom->out_dims[i - 1]; e.aux_valid = pe_dim;
                e.scale = 0.70710678118654752440f;  // cat([x, pe]) / sqrt(2), vector_field_network.py:193
            } else { e.act_col_off = 0; e.act_valid = VFN_HIDDEN; e.aux_valid = 0; }
        } else {
            if (i == 0) { e.act_col_off = 6 + pe_dim; e.act_valid = F; e.aux_col_off = 0; e.aux_valid = 6 + pe_dim; }
            else { e.act_col_off = 0; e.act_valid = VFN_HIDDEN; e.aux_valid = 0; }
        }
    }
    if (!bwd) {   // head: reference rows 0..2 of the last Linear
        PackEntry& e = a.e[a.n_entries++];
        int rc = fill_params(e, L - 1);
        if (rc != VFN_OK) return rc;
        e.w_off = plan.head_w_off; e.b_off = plan.head_b_off;
        e.kb_total = plan.head_nkb16; e.nkb_act = plan.head_nkb16;
        e.n_w = plan.head_nkb16 * 256u; e.n_b = 16u;
        e.is_head = 1; e.row_off = 0; e.n_rows = 3;
        e.act_col_off = 0; e.act_valid = VFN_HIDDEN; e.aux_valid = 0;
    }
    a.total = bwd ? plan.total_bwd_floats : plan.total_floats;
    return VFN_OK;
}

static int pack_common(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers, float* packed,
                       void* stream, bool bwd, const char* what) {
    VfnNetPlan plan;
    char err[256] = {0};
    int rc = vfn_make_plan(net_kind, geom, &plan, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("%s: %s", what, err); return rc; }
    VFN_REQUIRE(layers && packed, "%s: NULL argument", what);
    PackArgs a;
    rc = build_pack_args(net_kind, geom, layers, plan, bwd, a);
    if (rc != VFN_OK) return rc;
    a.out = packed;
    if (a.total == 0) return VFN_OK;
    const uint32_t threads = 256, blocks = (a.total + threads - 1) / threads;
    hipLaunchKernelGGL(vfn_pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, a);
    return vfn_check_launch(what);
}

extern "C" int vfn_pack_weights(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                                float* packed, void* stream) {
    return pack_common(net_kind, geom, layers, packed, stream, false, "vfn_pack_weights");
}

extern "C" int64_t vfn_packed_bwd_size(int32_t net_kind, const vfn_net_geom* geom) {
    VfnNetPlan plan;
    char err[256] = {0};
    int rc = vfn_make_plan(net_kind, geom, &plan, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("vfn_packed_bwd_size: %s", err); return rc; }
    return (int64_t)plan.total_bwd_floats;
}

extern "C" int vfn_pack_weights_bwd(int32_t net_kind, const vfn_net_geom* geom, const vfn_layer_params* layers,
                                    float* packed_bwd, void* stream) {
    return pack_common(net_kind, geom, layers, packed_bwd, stream, true, "vfn_pack_weights_bwd");
}
