// Which MFMA shape should the f16x3 kernels be built on?  (MI355X_MICROARCH.md "DVFS give-back" (7):
// on random data a bare 16x16x32 loop delivered 1.12-1.15x the FLOP/s of a 32x32x16 loop at equal cycles per FLOP.)
//
// This loop has the SHAPE of the fused kernel's inner loop (csrc/vfn_mlp16.hip, layer16): one wave per SIMD, every CU busy;
// per slab of K = 32 a wave multiplies a 32-row weight tile (A fragments re-read from LDS as lane-linear 1-KiB blocks, hi and
// lo planes) with the activations of its 32 points (B operands resident in registers, hi and lo halves), three products per
// fp32-equivalent product, plus the VALU work the kernel hides behind the MFMAs (6 instructions per 3 MFMAs of the 32x32 form).
//   form 0: v_mfma_f32_32x32x16_f16   6 MFMAs per slab (2 K-steps x 3 products), one 32x32 accumulator (16 registers)
//   form 1: v_mfma_f32_16x16x32_f16  12 MFMAs per slab (2 row tiles x 2 point groups x 3 products), four 16x16 accumulators
// Same LDS bytes, same FLOPs, same B registers per slab.  Random operands in [-1, 1) (zeros rank the shapes by cycles only).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int FORM, int FILL, bool RANDOM>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 lds[8192];      // 128 KiB: 32 slabs x (2 tiles/steps x hi|lo) x 1 KiB
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) {
        unsigned h[4];
        for (int q = 0; q < 4; ++q) {
            const unsigned r = hashu(i * 4 + q + blockIdx.x * 77777u);
            const _Float16 a = RANDOM ? (_Float16)(((r & 0xffff) / 32768.0f) - 1.0f) : (_Float16)1.0f;
            const _Float16 b = RANDOM ? (_Float16)(((r >> 16) / 32768.0f) - 1.0f) : (_Float16)1.0f;
            h[q] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        }
        lds[i] = uint4{h[0], h[1], h[2], h[3]};
    }
    __syncthreads();
    // B operands of 4 slabs (K = 128), hi and lo: 4 slabs x 2 blocks x 2 halves = 16 registers-of-8-halves, cycled through
    half8 bh[8], bl[8];
    for (int q = 0; q < 8; ++q) {
        bh[q] = __builtin_bit_cast(half8, lds[(4096 + q * 64 + lane) & 8191]);
        bl[q] = __builtin_bit_cast(half8, lds[(6144 + q * 64 + lane) & 8191]);
    }
    f32x16 acc32;
    f32x4 acc16[4];
    for (int q = 0; q < 16; ++q) acc32[q] = 0.f;
    for (int t = 0; t < 4; ++t) acc16[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float fill[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {                    // 8 slabs of K = 32 = one 256-deep layer tile
            const uint4* cb = lds + ((it * 8 + s) & 31) * 256;
            const half8 a0h = __builtin_bit_cast(half8, cb[lane]), a0l = __builtin_bit_cast(half8, cb[64 + lane]);
            const half8 a1h = __builtin_bit_cast(half8, cb[128 + lane]), a1l = __builtin_bit_cast(half8, cb[192 + lane]);
            const half8 x0h = bh[(2 * s) & 7], x0l = bl[(2 * s) & 7], x1h = bh[(2 * s + 1) & 7], x1l = bl[(2 * s + 1) & 7];
            if (FORM == 0) {                              // two K-steps of 16: (a0, x0), (a1, x1)
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, x0h, acc32, 0, 0, 0);
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, x0l, acc32, 0, 0, 0);
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, x0h, acc32, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FILL; ++f) fill[f] = fmaf(fill[f], 1.0001f, 0.5f);
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, x1h, acc32, 0, 0, 0);
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, x1l, acc32, 0, 0, 0);
                acc32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, x1h, acc32, 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FILL; ++f) fill[f] = fmaf(fill[f], 1.0001f, 0.5f);
            } else {                                      // row tiles (a0, a1) x point groups (x0, x1), K = 32 each
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const half8 ah = (t & 2) ? a1h : a0h, al = (t & 2) ? a1l : a0l;
                    const half8 xh = (t & 1) ? x1h : x0h, xl = (t & 1) ? x1l : x0l;
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh, acc16[t], 0, 0, 0);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl, acc16[t], 0, 0, 0);
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh, acc16[t], 0, 0, 0);
                    if (t & 1) {
#pragma unroll
                        for (int f = 0; f < FILL; ++f) fill[f] = fmaf(fill[f], 1.0001f, 0.5f);
                    }
                }
            }
        }
        if ((it & 3) == 3) {                              // keep the sums finite
            for (int q = 0; q < 16; ++q) acc32[q] *= 1e-3f;
            for (int t = 0; t < 4; ++t) acc16[t] *= 1e-3f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int q = 0; q < 16; ++q) sum += acc32[q];
    for (int t = 0; t < 4; ++t) sum += acc16[t][0] + acc16[t][1] + acc16[t][2] + acc16[t][3];
    for (int f = 0; f < 6; ++f) sum += fill[f] * 1e-30f;
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int FORM, int FILL, bool RANDOM> void run(const char* name) {
    float* out; unsigned long long* st;
    const int blocks = 256, iters = 600;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&st, blocks * 16);
    for (int rep = 0; rep < 6; ++rep) hipLaunchKernelGGL((k<FORM, FILL, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[512];
    (void)hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    const double slabs = (double)iters * 8;                            // per wave
    const double flop_slab = 2.0 * 3 * 32 * 32 * 32;                  // executed f16 FLOPs per wave and slab (3 products)
    const double ns_slab = real / blocks / slabs * 10.0;
    printf("%-58s cycles/slab %6.1f   clock %4.0f MHz   ns/slab %6.2f   executed f16 %5.0f TFLOP/s   fp32-equivalent %4.0f TFLOP/s\n", name,
           cyc / blocks / slabs, cyc / real * 100.0, ns_slab, flop_slab * 1024 / ns_slab / 1000.0, flop_slab / 3 * 1024 / ns_slab / 1000.0);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    run<0, 0, true>("32x32x16, random, A from LDS, no VALU filler");
    run<1, 0, true>("16x16x32, random, A from LDS, no VALU filler");
    run<0, 6, true>("32x32x16, random, A from LDS, 6 VALU per 3 MFMAs");
    run<1, 6, true>("16x16x32, random, A from LDS, 6 VALU per 6 MFMAs (same)");
    run<0, 0, false>("32x32x16, constant operands, A from LDS");
    run<1, 0, false>("16x16x32, constant operands, A from LDS");
    run<0, 6, true>("32x32x16, random (repeat)");
    run<1, 6, true>("16x16x32, random (repeat)");
    return 0;
}
