/* Pure-C driver of the C ABI (include/vfn.h): no Python, no torch — only the HIP runtime for device memory.
 *
 * What a maintainer binding the library from another host language would do: build a network geometry, hand over raw
 * parameter arrays in the reference's layout (nn.Linear weight[out][in], BatchNorm1d vectors), pack, launch on a stream.
 * Checks (the numbers are this program's own, no fixture needed):
 *   1. vfn_raygen_uniform on an identity pose: directions = ((u-cx)/fx, (v-cy)/fy, 1), z = near + (far-near) t;
 *   2. the f16x3 vector-field kernel against the exact-fp32 kernel on the same random weights: <= 2e-5;
 *   3. vfn_linear_rows (batch-statistics path) against a scalar loop on the host: <= 1e-5 of the largest entry;
 *   4. error behaviour: an unsupported geometry returns VFN_ERR_INVALID/UNSUPPORTED and a message, a NULL pointer returns
 *      VFN_ERR_INVALID — nothing crashes.
 * Build + run (tests/test_hip_parity.py::test_c_abi_from_plain_c does it):
 *   hipcc -x c -std=c11 tests/c_abi/abi_smoke.c -Iinclude -Lvf_nerf_amd/csrc -lvfn -Wl,-rpath,$PWD/vf_nerf_amd/csrc -o abi_smoke
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vfn.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_VFN(x) do { int r_ = (x); if (r_ != VFN_OK) { fprintf(stderr, "vfn status %d at %s:%d: %s\n", r_, __FILE__, __LINE__, vfn_last_error()); return 3; } } while (0)

static unsigned rng_state = 12345u;
static float frand(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }   /* [0,1) */

static float* to_device(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void) {
    if (vfn_abi_version() != VFN_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    /* ---- 1. rays ------------------------------------------------------------------------------------ */
    enum { N = 37, SC = 8 };
    float uv[N * 2], pose[N * 16], K[N * 16], t_vals[SC];
    const float fx = 60.f, fy = 50.f, cx = 31.5f, cy = 23.5f;
    for (int i = 0; i < N; ++i) {
        uv[2 * i] = floorf(frand() * 64.f); uv[2 * i + 1] = floorf(frand() * 48.f);
        memset(pose + 16 * i, 0, 64); memset(K + 16 * i, 0, 64);
        for (int d = 0; d < 4; ++d) pose[16 * i + 5 * d] = 1.f;
        K[16 * i + 0] = fx; K[16 * i + 5] = fy; K[16 * i + 2] = cx; K[16 * i + 6] = cy; K[16 * i + 10] = 1.f; K[16 * i + 15] = 1.f;
    }
    for (int j = 0; j < SC; ++j) t_vals[j] = (float)j / (float)(SC - 1);
    float *d_uv = to_device(uv, N * 2), *d_pose = to_device(pose, N * 16), *d_K = to_device(K, N * 16), *d_t = to_device(t_vals, SC);
    float *d_dir, *d_rd, *d_cam, *d_z, *d_pts;
    CHECK_HIP(hipMalloc((void**)&d_dir, N * 3 * 4)); CHECK_HIP(hipMalloc((void**)&d_rd, N * 3 * 4)); CHECK_HIP(hipMalloc((void**)&d_cam, N * 3 * 4));
    CHECK_HIP(hipMalloc((void**)&d_z, N * SC * 4)); CHECK_HIP(hipMalloc((void**)&d_pts, N * SC * 3 * 4));
    vfn_raygen_params rp = {N, SC, 0, 0.25f, 2.0f};
    CHECK_VFN(vfn_raygen_uniform(&rp, d_uv, d_pose, d_K, d_t, NULL, NULL, d_dir, d_rd, d_cam, d_z, d_pts, stream));
    float dir[N * 3], z[N * SC];
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(dir, d_dir, sizeof(dir), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(z, d_z, sizeof(z), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
        const float ex = (uv[2 * i] - cx) / fx, ey = (uv[2 * i + 1] - cy) / fy;
        if (fabsf(dir[3 * i] - ex) > 1e-6f || fabsf(dir[3 * i + 1] - ey) > 1e-6f || fabsf(dir[3 * i + 2] - 1.f) > 1e-6f) {
            fprintf(stderr, "ray %d: direction (%g %g %g), expected (%g %g 1)\n", i, dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], ex, ey); return 4; }
        for (int j = 0; j < SC; ++j)
            if (fabsf(z[i * SC + j] - (0.25f * (1.f - t_vals[j]) + 2.0f * t_vals[j])) > 1e-6f) { fprintf(stderr, "z mismatch\n"); return 4; }
    }

    /* ---- 2. the shipped vector-field network on random weights: f16x3 kernel vs exact-fp32 kernel ------ */
    vfn_net_geom g; memset(&g, 0, sizeof(g));
    g.n_layers = 9; g.multires = 6; g.skip_layer = 4; g.feature_dims = 256;
    const int in_d[9] = {39, 256, 256, 256, 256, 256, 256, 256, 256}, out_d[9] = {256, 256, 256, 217, 256, 256, 256, 256, 259};
    vfn_layer_params lp[9]; memset(lp, 0, sizeof(lp));
    for (int l = 0; l < 9; ++l) {
        g.in_dims[l] = in_d[l]; g.out_dims[l] = out_d[l]; g.has_bn[l] = l < 8;
        const size_t nw = (size_t)in_d[l] * out_d[l];
        float* w = (float*)malloc(nw * 4); float* v = (float*)malloc(out_d[l] * 4 * 5);
        const float bound = 2.0f / sqrtf((float)in_d[l]);
        for (size_t i = 0; i < nw; ++i) w[i] = (2.f * frand() - 1.f) * bound;
        for (int i = 0; i < out_d[l]; ++i) {
            v[i] = (2.f * frand() - 1.f) * 0.1f;                      /* Linear bias */
            v[out_d[l] + i] = 0.8f + 0.4f * frand();                   /* BN weight */
            v[2 * out_d[l] + i] = (2.f * frand() - 1.f) * 0.1f;        /* BN bias */
            v[3 * out_d[l] + i] = (2.f * frand() - 1.f) * 0.05f;       /* running mean */
            v[4 * out_d[l] + i] = 0.5f + frand();                      /* running var */
        }
        float* dw = to_device(w, nw); float* dv = to_device(v, (size_t)out_d[l] * 5);
        lp[l].weight = dw; lp[l].bias = dv;
        if (l < 8) { lp[l].bn_weight = dv + out_d[l]; lp[l].bn_bias = dv + 2 * out_d[l]; lp[l].bn_mean = dv + 3 * out_d[l]; lp[l].bn_var = dv + 4 * out_d[l]; }
        free(w); free(v);
    }
    const int64_t n32 = vfn_packed_size(VFN_NET_VF, &g), n16 = vfn_pack16_size(VFN_NET_VF, &g);
    if (n32 <= 0 || n16 <= 0) { fprintf(stderr, "packed sizes %lld / %lld: %s\n", (long long)n32, (long long)n16, vfn_last_error()); return 5; }
    float* d_p32; void* d_p16;
    CHECK_HIP(hipMalloc((void**)&d_p32, (size_t)n32 * 4)); CHECK_HIP(hipMalloc(&d_p16, (size_t)n16));
    CHECK_VFN(vfn_pack_weights(VFN_NET_VF, &g, lp, d_p32, stream));
    CHECK_VFN(vfn_pack16_weights(VFN_NET_VF, &g, lp, d_p16, stream));
    enum { M = 1000 };
    float pts[M * 3];
    for (int i = 0; i < M * 3; ++i) pts[i] = 2.f * frand() - 1.f;
    float* d_q = to_device(pts, M * 3); float *d_o32, *d_o16;
    CHECK_HIP(hipMalloc((void**)&d_o32, M * 3 * 4)); CHECK_HIP(hipMalloc((void**)&d_o16, M * 3 * 4));
    CHECK_VFN(vfn_vf_mlp_fwd(&g, d_p32, d_q, M, 3, d_o32, stream));
    CHECK_VFN(vfn_vf_mlp16_fwd(&g, d_p16, d_q, M, d_o16, stream));
    float o32[M * 3], o16[M * 3];
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(o32, d_o32, sizeof(o32), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(o16, d_o16, sizeof(o16), hipMemcpyDeviceToHost));
    float worst = 0.f, span = 0.f;
    for (int i = 0; i < M * 3; ++i) { worst = fmaxf(worst, fabsf(o32[i] - o16[i])); span = fmaxf(span, fabsf(o32[i])); }
    if (!(worst <= 2e-5f) || !(span > 0.05f)) { fprintf(stderr, "f16x3 vs fp32: max |diff| %g (outputs up to %g)\n", worst, span); return 6; }

    /* ---- 3. one layer of the batch-statistics path against the host ------------------------------------- */
    enum { LM = 300, LK = 40, LN = 24 };
    float a[LM * LK], w[LN * 39], b[LN], c[LM * LN];
    for (int i = 0; i < LM; ++i) for (int k = 0; k < LK; ++k) a[i * LK + k] = k < 39 ? 2.f * frand() - 1.f : 0.f;      /* pad column zero */
    for (int i = 0; i < LN * 39; ++i) w[i] = 2.f * frand() - 1.f;
    for (int i = 0; i < LN; ++i) b[i] = frand();
    float *d_a = to_device(a, LM * LK), *d_w = to_device(w, LN * 39), *d_b = to_device(b, LN), *d_c;
    CHECK_HIP(hipMalloc((void**)&d_c, LM * LN * 4));
    CHECK_VFN(vfn_linear_rows(0, d_a, LK, d_w, 39, d_b, LM, LN, 39, 0, d_c, LN, NULL, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(c, d_c, sizeof(c), hipMemcpyDeviceToHost));
    float lworst = 0.f, lspan = 0.f;
    for (int i = 0; i < LM; ++i) for (int n = 0; n < LN; ++n) {
        double s = b[n];
        for (int k = 0; k < 39; ++k) s += (double)a[i * LK + k] * (double)w[n * 39 + k];
        lworst = fmaxf(lworst, fabsf((float)s - c[i * LN + n])); lspan = fmaxf(lspan, fabsf((float)s));
    }
    if (!(lworst <= 1e-5f * lspan)) { fprintf(stderr, "vfn_linear_rows: max |diff| %g of %g\n", lworst, lspan); return 7; }

    /* ---- 4. errors are statuses with messages ----------------------------------------------------------- */
    vfn_net_geom bad = g; bad.out_dims[1] = 128; bad.in_dims[2] = 128;
    if (vfn_packed_size(VFN_NET_VF, &bad) >= 0 || strlen(vfn_last_error()) == 0) { fprintf(stderr, "unsupported geometry accepted\n"); return 8; }
    if (vfn_vf_mlp_fwd(&g, d_p32, NULL, M, 3, d_o32, stream) != VFN_ERR_INVALID) { fprintf(stderr, "NULL pointer accepted\n"); return 8; }

    printf("abi_smoke: ok (rays exact; f16x3 vs fp32 max |diff| %.2e; linear_rows max |diff| %.2e of %.2f)\n", worst, lworst, lspan);
    return 0;
}
