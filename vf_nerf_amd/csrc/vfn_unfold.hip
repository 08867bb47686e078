// vfn_unfold.hip — from weight-gradient partial slabs to the reference's parameter gradients, all layers of a net in
// one launch.
//
// The weight-gradient kernels (vfn_dw16.hip, vfn_mlp_bwd.hip) produce, per layer, `groups` partial slabs of
// dW'[n][k] = sum_m dY[m][n] X[m][k] (act inputs [G][R][256], aux inputs [G][256][64]) and of db'[n] = sum_m dY[m][n],
// for the FOLDED layer W' = s * scale * W, b' = s (b - mu) + beta with s = gamma / sqrt(var + eps) (eval-mode BatchNorm,
// vector_field_network.py:47-60,177-208; skip scale 1/sqrt(2)).  This kernel sums the slabs and applies the chain rule
// back to the parameters the optimizer holds (train/vector_field_nerf_train.py:252-260):
//     dW[n][k] = s_n scale dW'[n][k]            db[n] = s_n db'[n]
//     dgamma[n] = (sum_k scale dW'[n][k] W[n][k] + db'[n] (b_n - mu_n)) / sqrt(var_n + eps)      dbeta[n] = db'[n]
// One workgroup per (output row, layer entry): thread k owns column k (256 act + 64 aux), reads it across the slabs
// (coalesced across threads), the row's gamma gradient is a block reduction.
#include <string.h>
#include "vfn_common.h"

namespace {

constexpr int UF_MAX = 12;
struct UnfoldEntry {
    const float* dw_act;   // [G][slab_rows][256] or NULL
    const float* dw_aux;   // [G][256][64] or NULL
    const float* db;       // [G][slab_rows]
    const float* w; const float* b_lin; const float* bn_w; const float* bn_var; const float* bn_mean;   // bn_* NULL: no BatchNorm
    float* g_w; float* g_b; float* g_bn_w; float* g_bn_b;
    int rows, row_off, in_dim, slab_rows;
    int act_c0, act_nc, aux_c0, aux_nc;
    float scale;
    int groups_act, groups_aux, groups_db;      // 0: the launch's `groups`
};
struct UnfoldArgs {
    UnfoldEntry e[UF_MAX];
    int n_entries;
    int groups;
    unsigned accumulate;   // bit i: entry i ADDS to what its gradient tensors hold (they are the parameters' .grad) instead of overwriting
};

__global__ __launch_bounds__(320) void vfn_unfold_kernel(const UnfoldArgs a) {
    __shared__ float red[320 / 64 + 1][2];
    const UnfoldEntry& e = a.e[blockIdx.y];
    const bool acc = (a.accumulate >> blockIdx.y) & 1u;
    const int n = blockIdx.x;
    if (n >= e.rows) return;
    const int tid = threadIdx.x;
    const int row = e.row_off + n;
    // this thread's column
    const bool is_act = tid < 256;
    const int k = is_act ? tid : tid - 256;
    const float* slab = is_act ? e.dw_act : e.dw_aux;
    const int nc = is_act ? e.act_nc : e.aux_nc, c0 = is_act ? e.act_c0 : e.aux_c0;
    const size_t gstride = is_act ? (size_t)e.slab_rows * 256 : (size_t)256 * 64;
    const int ld = is_act ? 256 : 64;
    const int G = is_act ? (e.groups_act ? e.groups_act : a.groups) : (e.groups_aux ? e.groups_aux : a.groups);
    const int GB = e.groups_db ? e.groups_db : a.groups;
    float s = 0.f;
    const bool valid = slab != nullptr && k < nc;
    if (valid) {
        const float* p = slab + (size_t)n * ld + k;
        int g = 0;
        for (; g + 8 <= G; g += 8) {          // eight slabs in flight per thread (the sum keeps the slab order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(g + u) * gstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < G; ++g) s += p[(size_t)g * gstride];
    }
    float dbp = 0.f;
    for (int g = tid; g < GB; g += blockDim.x) dbp += e.db[(size_t)g * e.slab_rows + n];
    const float inv = e.bn_w ? rsqrtf(e.bn_var[row] + 1e-5f) : 0.f;
    const float s_fold = e.bn_w ? e.bn_w[row] * inv : 1.0f;
    const float sub = s * e.scale;
    float dg = 0.f;
    if (valid) {
        const size_t o = (size_t)row * e.in_dim + c0 + k;
        e.g_w[o] = (acc ? e.g_w[o] : 0.f) + s_fold * sub;
        if (e.bn_w) dg = sub * e.w[o];
    }
    // block reduction of (dg, dbp)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { dg += __shfl_down(dg, off, 64); dbp += __shfl_down(dbp, off, 64); }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) { red[wave][0] = dg; red[wave][1] = dbp; }
    __syncthreads();
    if (tid == 0) {
        float dgam = 0.f, db = 0.f;
        for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) { dgam += red[w][0]; db += red[w][1]; }
        e.g_b[row] = (acc ? e.g_b[row] : 0.f) + s_fold * db;
        if (e.bn_w) {
            e.g_bn_w[row] = (acc ? e.g_bn_w[row] : 0.f) + (dgam + db * (e.b_lin[row] - e.bn_mean[row])) * inv;
            e.g_bn_b[row] = (acc ? e.g_bn_b[row] : 0.f) + db;
        }
    }
}

}  // namespace

// Flat C view of UnfoldEntry (include/vfn.h: vfn_unfold_entry has the same fields in the same order)
extern "C" int vfn_unfold_weight_grads(const vfn_unfold_entry* entries, int32_t n_entries, int32_t groups, void* stream) {
    return vfn_unfold_weight_grads_acc(entries, n_entries, groups, 0u, stream);
}

extern "C" int vfn_unfold_weight_grads_acc(const vfn_unfold_entry* entries, int32_t n_entries, int32_t groups, uint32_t accumulate_mask,
                                           void* stream) {
    VFN_REQUIRE(entries && n_entries >= 1 && n_entries <= UF_MAX, "vfn_unfold_weight_grads: n_entries=%d outside [1,%d]", n_entries, UF_MAX);
    VFN_REQUIRE(groups >= 1 && groups <= 4096, "vfn_unfold_weight_grads: groups=%d", groups);
    static_assert(sizeof(vfn_unfold_entry) == sizeof(UnfoldEntry), "ABI struct and kernel struct must match");
    UnfoldArgs a;
    memset(&a, 0, sizeof(a));
    int max_rows = 0;
    for (int i = 0; i < n_entries; ++i) {
        memcpy(&a.e[i], &entries[i], sizeof(UnfoldEntry));
        const UnfoldEntry& e = a.e[i];
        VFN_REQUIRE(e.db && e.w && e.g_w && e.g_b && (e.dw_act || e.dw_aux), "vfn_unfold_weight_grads: entry %d has a NULL pointer", i);
        VFN_REQUIRE(!e.bn_w || (e.bn_var && e.bn_mean && e.b_lin && e.g_bn_w && e.g_bn_b), "vfn_unfold_weight_grads: entry %d BatchNorm pointer NULL", i);
        VFN_REQUIRE(e.rows >= 1 && e.rows <= e.slab_rows && e.slab_rows <= 256 && e.act_nc <= 256 && e.aux_nc <= 64,
                    "vfn_unfold_weight_grads: entry %d has bad sizes", i);
        VFN_REQUIRE(e.groups_act >= 0 && e.groups_act <= 4096 && e.groups_aux >= 0 && e.groups_aux <= 4096 && e.groups_db >= 0 && e.groups_db <= 4096,
                    "vfn_unfold_weight_grads: entry %d has bad slab counts", i);
        max_rows = e.rows > max_rows ? e.rows : max_rows;
    }
    a.n_entries = n_entries; a.groups = groups; a.accumulate = accumulate_mask;
    hipLaunchKernelGGL(vfn_unfold_kernel, dim3(max_rows, n_entries), dim3(320), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_unfold_weight_grads");
}
