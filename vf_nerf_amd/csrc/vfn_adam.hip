// vfn_adam.hip — gradient clipping and Adam over ONE flat parameter buffer.
//
// The reference's optimizer side of a training step (train/vector_field_nerf_train.py:254-260):
//     torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm);  optimizer.step()
// over a parameter list that names every vector-field parameter twice (models/nerf/vector_field_nerf.py:57-63):
// with the sequential per-entry loops that list implies, a duplicated gradient counts twice in the total norm, is scaled by
// the clip coefficient twice, and its parameter receives two consecutive Adam updates from the same gradient per step().
// PyTorch's per-tensor / multi-tensor kernels need ~50 launches for the 89 list entries; here the unique parameters live in
// one flat fp32 buffer (gradients, first and second moments likewise), sorted into REGIONS of equal multiplicity, and a step
// is three launches: squared-norm partials (+ the final reduction by the last workgroup to finish), the in-place scaling, and
// the Adam update with `mult` consecutive updates per element.
//
// Arithmetic follows torch.optim.adam._single_tensor_adam, element for element:
//     m <- m + (1 - b1) (g - m);  v <- b2 v + (1 - b2) g g;  denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p <- p - (lr / (1 - b1^t)) m / denom
// with the bias corrections evaluated on the host in double precision, as PyTorch does.
#include "vfn_common.h"

namespace {

struct Regions {
    long long start[4], end[4];     // element ranges of the flat buffer, ascending
    int mult[4];                    // occurrences of these parameters in the optimizer's list
    int n;
};

__device__ __forceinline__ int mult_of(const Regions& r, long long i) {
    int m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < r.n && i >= r.start[k] && i < r.end[k]) m = r.mult[k];
    return m;
}

// partial[b] = sum over the block's elements of mult * g^2; the last block to arrive adds the partials up in index order
// (deterministic) and writes out[0] = total norm, out[1] = clip coefficient = min(max_norm / (total + 1e-6), 1).
__global__ __launch_bounds__(256) void vfn_grad_norm_kernel(const float* g, long long n, Regions r, float max_norm, double* partial,
                                                            unsigned* counter, float* out) {
    __shared__ double red[256];
    __shared__ bool last;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = g[i];
        s += (double)mult_of(r, i) * (double)v * (double)v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0];
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        double t = 0.0;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __builtin_nontemporal_load(partial + b);
        red[threadIdx.x] = t;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(red[0]);
            out[0] = total;
            out[1] = fminf(max_norm / (total + 1e-6f), 1.0f);
            *counter = 0u;            // ready for the next call
        }
    }
}

// g <- g * coef, once per occurrence (the sequential loop multiplies a duplicated gradient twice)
__global__ __launch_bounds__(256) void vfn_grad_scale_kernel(float* g, long long n, Regions r, const float* out) {
    const float coef = out[1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = g[i];
        const int m = mult_of(r, i);
        for (int k = 0; k < m; ++k) v *= coef;
        g[i] = v;
    }
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    Regions r;
    float beta1, beta2, eps, weight_decay, one_minus_b1, one_minus_b2;
    float step_size[4][2];          // [region][update k]: lr / (1 - beta1^t)
    float bc2_sqrt[4][2];           // sqrt(1 - beta2^t)
};

__global__ __launch_bounds__(256) void vfn_adam_kernel(const AdamArgs a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        int reg = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.r.n && i >= a.r.start[k] && i < a.r.end[k]) reg = k;
        if (reg < 0) continue;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        const float g0 = a.g[i];
        const int mult = a.r.mult[reg];
        for (int k = 0; k < mult; ++k) {
            const float g = a.weight_decay != 0.f ? g0 + a.weight_decay * p : g0;
            m = m + a.one_minus_b1 * (g - m);
            v = v * a.beta2 + (a.one_minus_b2 * g) * g;
            const float denom = sqrtf(v) / a.bc2_sqrt[reg][k] + a.eps;
            p = p - a.step_size[reg][k] * (m / denom);
        }
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}

int fill_regions(Regions& r, int n_regions, const int64_t* starts, const int64_t* ends, const int32_t* mults, long long n, const char* what) {
    VFN_REQUIRE(n_regions >= 1 && n_regions <= 4 && starts && ends && mults, "%s: 1..4 regions expected", what);
    r.n = n_regions;
    for (int k = 0; k < 4; ++k) { r.start[k] = r.end[k] = 0; r.mult[k] = 0; }
    for (int k = 0; k < n_regions; ++k) {
        VFN_REQUIRE(starts[k] >= 0 && ends[k] >= starts[k] && ends[k] <= n && mults[k] >= 1 && mults[k] <= 2,
                    "%s: bad region %d [%lld, %lld) x %d", what, k, (long long)starts[k], (long long)ends[k], mults[k]);
        r.start[k] = starts[k]; r.end[k] = ends[k]; r.mult[k] = mults[k];
    }
    return VFN_OK;
}

constexpr int NORM_BLOCKS = 256;

}  // namespace

extern "C" int64_t vfn_flat_clip_workspace_bytes(void) { return NORM_BLOCKS * 8 + 16; }

extern "C" int vfn_flat_clip_grad_norm(float* flat_grad, int64_t n, int32_t n_regions, const int64_t* starts, const int64_t* ends,
                                       const int32_t* mults, float max_norm, void* workspace, float* out2, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(flat_grad && workspace && out2, "vfn_flat_clip_grad_norm: NULL argument");
    Regions r;
    int rc = fill_regions(r, n_regions, starts, ends, mults, n, "vfn_flat_clip_grad_norm");
    if (rc != VFN_OK) return rc;
    double* partial = (double*)workspace;
    unsigned* counter = (unsigned*)((char*)workspace + NORM_BLOCKS * 8);     // zero on first use (the caller zero-fills once), reset by the kernel
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vfn_grad_norm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, flat_grad, (long long)n, r, max_norm, partial, counter, out2);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(vfn_grad_scale_kernel, dim3(blocks), dim3(256), 0, s, flat_grad, (long long)n, r, out2);
    return vfn_check_launch("vfn_flat_clip_grad_norm");
}

extern "C" int vfn_flat_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t n_regions,
                                  const int64_t* starts, const int64_t* ends, const int32_t* mults, const double* step_size,
                                  const double* bc2_sqrt, double beta1, double beta2, double eps, double weight_decay, void* stream) {
    if (n <= 0) return VFN_OK;
    VFN_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_size && bc2_sqrt, "vfn_flat_adam_step: NULL argument");
    AdamArgs a = {};
    int rc = fill_regions(a.r, n_regions, starts, ends, mults, n, "vfn_flat_adam_step");
    if (rc != VFN_OK) return rc;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    a.one_minus_b1 = (float)(1.0 - beta1); a.one_minus_b2 = (float)(1.0 - beta2);       // the Python doubles PyTorch hands its kernels
    for (int k = 0; k < n_regions; ++k)
        for (int u = 0; u < 2; ++u) { a.step_size[k][u] = (float)step_size[2 * k + u]; a.bc2_sqrt[k][u] = (float)bc2_sqrt[2 * k + u]; }
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(vfn_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return vfn_check_launch("vfn_flat_adam_step");
}
