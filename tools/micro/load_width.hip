// How fast does ONE 256-thread workgroup per CU stream the weight-gradient kernel's operands from HBM — and does the WIDTH of the loads
// matter?  (csrc/vfn_dwf.hip reads two 16-bit fragment-ordered slots with 8-byte loads per lane: 512 B per wave-instruction, 64 KiB per
// workgroup and step, 3.9-4.1 TB/s measured against ~6.3 achievable)
//
// Every workgroup walks its own slab of a large buffer in steps of 64 KiB laid out like the kernel's step (2 groups x 2 operands, i.e.
// four 16-KiB runs per step from four slots), double-buffered in registers exactly as the kernel holds them: the loads of step s + 1 are
// issued, then step s is consumed (a few VALU ops per loaded register, plus an optional spin of MFMA-like length), then the roles swap.
//   form 0: 32 x buffer_load_b64  per lane and step (what the kernel does)         form 1: 16 x buffer_load_b128 per lane and step
//   spin  : cycles of busy work per step between issue and consumption (0 = pure streaming; the kernel's MFMAs are ~2 000 cycles per step)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/load_width.hip -o /tmp/load_width && /tmp/load_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr long long SLOT = 1ll << 30;          // bytes per slot (four slots: 4 GiB read per launch at most)
constexpr int STEP = 16384;                    // bytes per slot and step

template <int FORM>
__global__ __launch_bounds__(256, 1) void stream(const unsigned char* base, long long steps_per_wg, int spin, unsigned* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long s0 = (long long)blockIdx.x * steps_per_wg;
    unsigned acc = 0;
    u32x4 reg[2][16];
    auto issue = [&](int b, long long s) {
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) {
            // a slab-relative descriptor per slot, as the kernel builds them
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base) + slot * SLOT + s0 * STEP, 0,
                                                                                (int)(steps_per_wg * STEP), 0x00020000);
            const int so = (int)(s - s0) * STEP;
            if (FORM == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {           // wave w reads pieces 8 w .. 8 w + 7 of 512 B
                    const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, so + (8 * wave + r) * 512, 0);
                    if (r & 1) { reg[b][4 * slot + (r >> 1)][2] = h[0]; reg[b][4 * slot + (r >> 1)][3] = h[1]; }
                    else { reg[b][4 * slot + (r >> 1)][0] = h[0]; reg[b][4 * slot + (r >> 1)][1] = h[1]; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)             // wave w reads pieces 4 w .. 4 w + 3 of 1 KiB
                    reg[b][4 * slot + r] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so + (4 * wave + r) * 1024, 0);
            }
        }
    };
    auto consume = [&](int b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += reg[b][i][0] ^ reg[b][i][1] ^ reg[b][i][2] ^ reg[b][i][3];
    };
    issue(0, s0);
    for (long long s = s0; s < s0 + steps_per_wg; s += 2) {
        if (s + 1 < s0 + steps_per_wg) issue(1, s + 1);
        if (spin) { const long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < spin) {} }
        consume(0);
        __syncthreads();
        if (s + 2 < s0 + steps_per_wg) issue(0, s + 2);
        if (spin) { const long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < spin) {} }
        if (s + 1 < s0 + steps_per_wg) consume(1);
        __syncthreads();
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + tid] = acc;
}

int main() {
    unsigned char* buf; unsigned* out;
    hipMalloc(&buf, 4 * SLOT); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 1, 4 * SLOT);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256;
    const long long steps_per_wg = SLOT / STEP / wgs;            // 256 steps of 64 KiB per workgroup: 4 GiB in all
    for (int spin : {0, 1000, 2000, 4000}) {
        for (int form = 0; form < 2; ++form) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (form == 0) hipLaunchKernelGGL(stream<0>, dim3(wgs), dim3(256), 0, 0, buf, steps_per_wg, spin, out);
                else hipLaunchKernelGGL(stream<1>, dim3(wgs), dim3(256), 0, 0, buf, steps_per_wg, spin, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = 4.0 * SLOT;
            printf("spin %5d cycles/step  %s: %8.3f ms  %6.2f TB/s  (%.1f B/cycle/CU at 2.1 GHz)\n", spin, form == 0 ? "32 x b64 " : "16 x b128", best,
                   bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.1e9);
        }
    }
    return 0;
}
