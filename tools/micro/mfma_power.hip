// Microbenchmark behind DESIGN.md's "power ceiling": time per dependent v_mfma_f32_32x32x16_f16 with every CU busy
// (one wave per SIMD), for constant-ish vs random operands, operands in registers vs re-read from LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int VARIANT>   // bit 0: random operands; bit 2: A fragments from LDS (2 ds_read_b128 per 3 MFMAs); bit 3: two independent
                         // accumulator chains (the cross terms a_hi*b_lo + a_lo*b_hi in their own accumulator)
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) {
        uint4 v;
        if (VARIANT & 1) {   // random halves in [-1, 1)
            unsigned h[4];
            for (int q = 0; q < 4; ++q) {
                const unsigned r = hashu(i * 4 + q + blockIdx.x * 77777u);
                const _Float16 a = (_Float16)(((r & 0xffff) / 32768.0f) - 1.0f), b = (_Float16)(((r >> 16) / 32768.0f) - 1.0f);
                h[q] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
            }
            v = uint4{h[0], h[1], h[2], h[3]};
        } else v = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        lds[i] = v;
    }
    __syncthreads();
    half8 b[4];
    for (int q = 0; q < 4; ++q) b[q] = __builtin_bit_cast(half8, lds[(q * 64 + lane + 4096) & 8191]);
    half8 areg[2] = {__builtin_bit_cast(half8, lds[lane]), __builtin_bit_cast(half8, lds[64 + lane])};
    f32x16 acc, acc2;
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            half8 a0 = areg[0], a1 = areg[1];
            if (VARIANT & 4) {
                a0 = __builtin_bit_cast(half8, lds[(u * 128 + lane) & 4095]);
                a1 = __builtin_bit_cast(half8, lds[(u * 128 + 64 + lane) & 4095]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[u & 3], acc, 0, 0, 0);
            if (VARIANT & 8) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[(u + 1) & 3], acc2, 0, 0, 0);
                if (u & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[u & 3], acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[u & 3], acc2, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[(u + 1) & 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[u & 3], acc, 0, 0, 0);
            }
        }
        if ((it & 7) == 7) for (int q = 0; q < 16; ++q) { acc[q] *= 1e-3f; acc2[q] *= 1e-3f; }   // keep the sums finite
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 16; ++q) s += acc[q] + acc2[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V> void run(const char* name) {
    float* out; unsigned long long* st;
    const int blocks = 256, iters = 400;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[512];
    (void)hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    const double n = (double)iters * 48;
    printf("%-44s cycles/MFMA %.1f   shader clock %.0f MHz   ns/MFMA %.2f   dense-f16 %.0f TFLOP/s\n", name, cyc / blocks / n,
           cyc / real * 100.0, real / blocks / n * 10.0, 32768.0 * 1024 / (real / blocks / n * 10.0) / 1000.0);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    run<0>("constant operands, registers");
    run<1>("random operands, registers");
    run<4>("constant operands, A from LDS");
    run<5>("random operands, A from LDS");
    run<9>("random operands, registers, 2 chains");
    run<13>("random operands, A from LDS, 2 chains");
    run<8>("constant operands, registers, 2 chains");
    return 0;
}
