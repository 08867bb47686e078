// vfn_wgrad.hip — the parameter gradients of ONE network from a fragment-ordered training workspace, as one entry point.
//
// loss.backward() through models/vector_field/vector_field_network.py:177-208 / rendering_network.py:62-108
// (train/vector_field_nerf_train.py:252) needs, per nn.Linear (+ eval-mode BatchNorm1d), dW = dY^T X, db and the BatchNorm
// gradients.  The facade used to issue these launch by launch — per layer entry a partial-slab launch for the activation
// columns, one for the encoding columns, then one un-fold launch per net — with a torch allocation per slab: ~45 Python-side
// operations per backward, 1.9 ms of host time at the reference's 1 024-ray batches where the device needs 1.2 ms.
// vfn_net_weight_grads_frag is that sequence issued from C out of one caller-supplied scratch buffer:
//     (round 3: the products of one shape and operand form leave as ONE launch of slabs x products workgroups, see the loop below)
//     for every layer entry h of the net (csrc/vfn_plan.h; vf_nerf_amd/backward.py::_entries is the same table):
//         vfn_weight_grad_frag(shape 0)  dY_h^T X_h      [256][256] slabs, X_h = the slot of entry h-1 (rendering net, h = 0: the
//                                                         fp32 feature rows)
//         vfn_weight_grad_frag(shape 1)  dY_h^T aux      [256][64] slabs (entries that read the encoding tile)
//     vfn_weight_grad_frag(shape 2)      dz_head^T X_last for the 3-channel head
//     vfn_unfold_weight_grads_acc        slabs summed, BatchNorm / skip scale un-folded, added into (or written over) the
//                                        parameters' gradient tensors
// Nothing here computes: it sequences entry points of this library, so every value is what those entry points produce.
#include <math.h>
#include <string.h>
#include "vfn_common.h"
#include "vfn_plan.h"

namespace {

struct Entry {
    int layer, rows, row_off;
    int act_c0, act_nc;      // act_nc = 0: no activation input
    int aux_c0, aux_nc;      // aux_nc = 0: no encoding input
    float scale;
    int x_slot;              // slot holding the activation input; -1: the caller's fp32 feature rows (rendering net, entry 0)
};

// the entries of a net in slot order, then the slot that feeds the 3-channel head
int make_entries(int net_kind, const vfn_net_geom* g, Entry* e, int* head_x_slot, const char* what) {
    VfnNetPlan plan;
    char err[256] = {0};
    int rc = vfn_make_plan(net_kind, g, &plan, err, sizeof(err));
    if (rc != VFN_OK) { vfn_set_error("%s: %s", what, err); return rc; }
    const int L = g->n_layers, pe = plan.pe_dim, F = g->feature_dims;
    int n = 0;
    if (net_kind == VFN_NET_VF) {
        for (int i = 0; i < L - 1; ++i) {
            Entry& x = e[n];
            x = Entry{i, g->out_dims[i], 0, 0, 0, 0, 0, 1.0f, n - 1};
            if (i == 0) { x.aux_c0 = 0; x.aux_nc = pe; }
            else if (i == g->skip_layer) {
                x.act_c0 = 0; x.act_nc = g->out_dims[i - 1]; x.aux_c0 = g->out_dims[i - 1]; x.aux_nc = pe;
                x.scale = 0.70710678118654752440f;
            } else { x.act_c0 = 0; x.act_nc = VFN_HIDDEN; }
            ++n;
        }
        const int last_plain = n - 1;
        if (F > 0) { e[n] = Entry{L - 1, F, 3, 0, VFN_HIDDEN, 0, 0, 1.0f, last_plain}; ++n; }
        *head_x_slot = last_plain;
    } else {
        for (int i = 0; i < L - 1; ++i) {
            Entry& x = e[n];
            x = Entry{i, g->out_dims[i], 0, 0, VFN_HIDDEN, 0, 0, 1.0f, n - 1};
            if (i == 0) { x.act_c0 = 6 + pe; x.act_nc = F; x.aux_c0 = 0; x.aux_nc = 6 + pe; x.x_slot = -1; }
            ++n;
        }
        *head_x_slot = n - 1;
    }
    if (n != plan.n_hidden) { vfn_set_error("%s: %d entries against %d planned layers", what, n, plan.n_hidden); return VFN_ERR_INVALID; }
    return n;
}

int groups_for(int64_t m) {
    const int64_t g = m / 256;
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

struct Carve {
    unsigned char* base;
    size_t off;
    float* take(size_t floats) {
        float* p = reinterpret_cast<float*>(base ? base + off : nullptr);
        off += ((floats * 4 + 255) / 256) * 256;
        return p;
    }
};

}  // namespace

extern "C" int32_t vfn_weight_grad_groups(int64_t n_points) { return groups_for(n_points); }

extern "C" int64_t vfn_net_weight_grads_scratch_bytes(int32_t net_kind, const vfn_net_geom* geom, int64_t n_points) {
    if (!geom || n_points < 0) return VFN_ERR_INVALID;
    Entry e[VFN_MAX_LAYERS + 1];
    int head_slot;
    const int n = make_entries(net_kind, geom, e, &head_slot, "vfn_net_weight_grads_scratch_bytes");
    if (n < 0) return n;
    const size_t G = (size_t)groups_for(n_points);
    Carve c{nullptr, 0};
    for (int h = 0; h < n; ++h) {
        c.take(G * VFN_HIDDEN);
        if (e[h].act_nc) c.take(G * VFN_HIDDEN * VFN_HIDDEN);
        if (e[h].aux_nc) c.take(G * VFN_HIDDEN * 64);
    }
    c.take(G * 32 * VFN_HIDDEN);
    c.take(G * 32);
    return (int64_t)c.off;
}

extern "C" int vfn_net_weight_grads_frag(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                                         const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                                         const float* aux, const float* dz_head, int64_t n_points, int32_t with_features,
                                         int32_t accumulate, void* scratch, void* stream) {
    return vfn_net_weight_grads_frag_part(net_kind, geom, layers, saved, dy, slot_bytes, dy_form, x_form, feats, aux, dz_head, n_points,
                                          with_features ? 0xffffffffu : ~VFN_WGRAD_FEATURES, accumulate, scratch, stream);
}

extern "C" int vfn_net_weight_grads_frag_part(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                                              const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                                              const float* aux, const float* dz_head, int64_t n_points, uint32_t parts,
                                              int32_t accumulate, void* scratch, void* stream) {
    return vfn_internal_net_weight_grads_frag_part(net_kind, geom, layers, saved, dy, slot_bytes, dy_form, x_form, feats, aux, dz_head, n_points, nullptr,
                                                   parts, accumulate, scratch, stream, 3);
}

// ... over min(n_points, *n_dev) points when n_dev (device memory) is given: the slab count and the scratch are sized for n_points, every
// slab's range is cut from the live count inside the kernels (slabs past it write zeros)
int vfn_internal_net_weight_grads_frag_part(int32_t net_kind, const vfn_net_geom* geom, const vfn_wgrad_layer* layers, const void* saved,
                                            const void* dy, int64_t slot_bytes, int32_t dy_form, int32_t x_form, const float* feats,
                                            const float* aux, const float* dz_head, int64_t n_points, const int32_t* n_dev, uint32_t parts,
                                            int32_t accumulate, void* scratch, void* stream, int32_t stages) {
    // stages: bit 0 the partial-slab products, bit 1 the un-fold launch — two calls with the same arguments otherwise, so that a caller can put
    // the products of one row range on a side stream beside another range's and order only the un-folds (both ADD into the same tensors)
    const char* what = "vfn_net_weight_grads_frag";
    const int slabs = n_dev ? (groups_for(n_points) < 64 ? groups_for(n_points) : 64) : groups_for(n_points);      // (see G below)
    auto one = [&](int32_t shape, const void* dy1, int32_t dyf, const void* x1, int32_t xf, float* dw1, float* db1) -> int {
        if (!(stages & 1)) return VFN_OK;
        return vfn_internal_weight_grad_frag_batch_dev(shape, dyf, xf, 1, &dy1, &x1, &dw1, &db1, n_points, n_dev, slabs, stream);
    };
    VFN_REQUIRE(geom && layers && saved && dy && aux && dz_head && scratch, "%s: NULL argument", what);
    VFN_REQUIRE(stages & 3, "%s: stages = 0", what);
    if (n_points <= 0) return VFN_OK;
    Entry e[VFN_MAX_LAYERS + 1];
    int head_slot;
    const int n = make_entries(net_kind, geom, e, &head_slot, what);
    if (n < 0) return n;
    VFN_REQUIRE(n + 1 <= 12, "%s: %d entries (the un-fold launch takes 12)", what, n + 1);
    VFN_REQUIRE(net_kind == VFN_NET_VF || feats, "%s: the rendering net's first layer reads the fp32 feature rows", what);
    VFN_REQUIRE(slot_bytes > 0 && slot_bytes % 1024 == 0, "%s: slot_bytes = %lld", what, (long long)slot_bytes);
    // Slabs: one per 256 points, at most 256 — but a launch over a device-side count (region 2 of a training step: a few percent of its
    // capacity) spreads the LIVE points over all of them, and every slab costs 256 KiB of partial sums written and read back whatever it
    // holds: with 256 slabs those launches moved 64 MB each for ~5 000 points of work (0.5 GB per step at 1 024 rays, a third of what the
    // step's real weight-gradient launches move).  64 slabs there.
    const int G = slabs;
    const unsigned char* sv = static_cast<const unsigned char*>(saved);
    const unsigned char* dyb = static_cast<const unsigned char*>(dy);
    Carve c{static_cast<unsigned char*>(scratch), 0};
    vfn_unfold_entry u[12];
    memset(u, 0, sizeof(u));
    int nu = 0;
    const bool has_feat = net_kind == VFN_NET_VF && geom->feature_dims > 0;
    VFN_REQUIRE(parts & (VFN_WGRAD_LAYERS | VFN_WGRAD_FEATURES | VFN_WGRAD_HEAD), "%s: parts = 0", what);
    // Products of one shape and operand form go out as ONE launch of (slabs x products) workgroups (csrc/vfn_dwf.hip, DwfBatch): the
    // activation-column products whose X is a workspace slot (7 + the feature block for the vector-field net, 3 for the rendering
    // net), and the encoding-column products.  A batch of k products runs G / k slabs each, so the chip is as full as with one launch
    // of G slabs per product while the partial slabs written here and read back by the un-fold shrink k-fold (64 MB per product
    // otherwise, whatever the batch size: a third of a weight-gradient launch's traffic at the reference's 1 024-ray batches).
    struct Batch { int n; const void* dy[8]; const void* x[8]; float* dw[8]; float* db[8]; int unfold_idx[8]; };
    Batch act = {}, enc = {};
    auto flush = [&](Batch& b, int shape, int xf) -> int {
        if (b.n == 0) return VFN_OK;
        int gb = G / b.n;
        gb = gb < 1 ? 1 : gb;
        int rc = (stages & 1) ? vfn_internal_weight_grad_frag_batch_dev(shape, dy_form, xf, b.n, b.dy, b.x, b.dw, b.db, n_points, n_dev, gb, stream)
                              : VFN_OK;
        for (int i = 0; i < b.n; ++i) {
            vfn_unfold_entry& o = u[b.unfold_idx[i]];
            if (shape == 0) { o.groups_act = gb; o.groups_db = gb; }
            else { o.groups_aux = gb; if (b.db[i]) o.groups_db = gb; }
        }
        b.n = 0;
        return rc;
    };
    for (int h = 0; h < n; ++h) {
        const Entry& x = e[h];
        float* db = c.take((size_t)G * VFN_HIDDEN);
        float* dw_act = x.act_nc ? c.take((size_t)G * VFN_HIDDEN * VFN_HIDDEN) : nullptr;
        float* dw_aux = x.aux_nc ? c.take((size_t)G * VFN_HIDDEN * 64) : nullptr;
        const bool is_feat = has_feat && h == n - 1;
        if (!(parts & (is_feat ? VFN_WGRAD_FEATURES : VFN_WGRAD_LAYERS))) continue;       // (vector-only forward: the feature block was never evaluated)
        const vfn_wgrad_layer& q = layers[x.layer];
        VFN_REQUIRE(q.weight && q.bias && q.g_weight && q.g_bias, "%s: layer %d has a NULL weight / bias / gradient pointer", what, x.layer);
        const void* dy_h = dyb + (size_t)h * slot_bytes;
        int rc;
        const int ui = nu;
        if (dw_act) {
            if (x.x_slot < 0) {       // the rendering net's first layer reads the fp32 feature ROWS: another operand form, its own launch
                rc = one(0, dy_h, dy_form, (const void*)feats, 2, dw_act, db);
                if (rc != VFN_OK) return rc;
            } else {
                if (act.n == 8) { rc = flush(act, 0, x_form); if (rc != VFN_OK) return rc; }
                act.dy[act.n] = dy_h; act.x[act.n] = sv + (size_t)x.x_slot * slot_bytes; act.dw[act.n] = dw_act; act.db[act.n] = db;
                act.unfold_idx[act.n++] = ui;
            }
        }
        if (dw_aux) {
            if (enc.n == 8) { rc = flush(enc, 1, 3); if (rc != VFN_OK) return rc; }
            enc.dy[enc.n] = dy_h; enc.x[enc.n] = aux; enc.dw[enc.n] = dw_aux; enc.db[enc.n] = dw_act ? nullptr : db;
            enc.unfold_idx[enc.n++] = ui;
        }
        vfn_unfold_entry& o = u[nu++];
        o.dw_act = dw_act; o.dw_aux = dw_aux; o.db = db;
        o.w = q.weight; o.b_lin = q.bias; o.g_w = q.g_weight; o.g_b = q.g_bias;
        if (geom->has_bn[x.layer]) {
            VFN_REQUIRE(q.bn_weight && q.bn_var && q.bn_mean && q.g_bn_weight && q.g_bn_bias, "%s: layer %d BatchNorm pointer NULL", what, x.layer);
            o.bn_w = q.bn_weight; o.bn_var = q.bn_var; o.bn_mean = q.bn_mean; o.g_bn_w = q.g_bn_weight; o.g_bn_b = q.g_bn_bias;
        }
        o.rows = x.rows; o.row_off = x.row_off; o.in_dim = geom->in_dims[x.layer]; o.slab_rows = VFN_HIDDEN;
        o.act_c0 = x.act_c0; o.act_nc = x.act_nc; o.aux_c0 = x.aux_c0; o.aux_nc = x.aux_nc; o.scale = x.scale;
    }
    {
        int rc = flush(act, 0, x_form);
        if (rc != VFN_OK) return rc;
        rc = flush(enc, 1, 3);
        if (rc != VFN_OK) return rc;
    }
    if (parts & VFN_WGRAD_HEAD) {   // 3-channel head = rows 0..2 of the last Linear (no BatchNorm)
        float* part = c.take((size_t)G * 32 * VFN_HIDDEN);
        float* dbp = c.take((size_t)G * 32);
        const int L = geom->n_layers;
        const vfn_wgrad_layer& q = layers[L - 1];
        VFN_REQUIRE(q.weight && q.bias && q.g_weight && q.g_bias, "%s: the last layer has a NULL weight / bias / gradient pointer", what);
        int rc = one(2, dz_head, 2, sv + (size_t)head_slot * slot_bytes, x_form, part, dbp);
        if (rc != VFN_OK) return rc;
        vfn_unfold_entry& o = u[nu++];
        o.dw_act = part; o.db = dbp; o.w = q.weight; o.b_lin = q.bias; o.g_w = q.g_weight; o.g_b = q.g_bias;
        o.rows = 3; o.row_off = 0; o.in_dim = geom->in_dims[L - 1]; o.slab_rows = 32; o.act_c0 = 0; o.act_nc = VFN_HIDDEN; o.scale = 1.0f;
    }
    if (nu == 0 || !(stages & 2)) return VFN_OK;
    return vfn_unfold_weight_grads_acc(u, nu, G, accumulate ? (1u << nu) - 1u : 0u, stream);
}
