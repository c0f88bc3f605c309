// scripts/experiments/valu_clock.hip -- what the vector ALUs of this box actually issue under a full-chip integer load:
// every SIMD runs W waves of dependent 32-bit integer instructions (xor / add / rotate, the sampler's mix); the kernel's
// duration (HIP events) against the instruction count gives wave-instructions per second, and s_memtime (shader clock) against
// s_memrealtime (100 MHz) gives the engine clock while it runs.  hipcc --offload-arch=gfx950 -O2 valu_clock.hip -o valu_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(256) void spin(unsigned* out, unsigned long long* clk, int iters) {
    unsigned a = threadIdx.x, b = blockIdx.x * 7u + 1u, c = 0x9E3779B9u, d = a ^ b;
    const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {   // 8 vector instructions, two chains of four
            a ^= b; b += c; c = (c << 7) | (c >> 25); d += a;
            b ^= d; d += c; a = (a << 11) | (a >> 21); c += b;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
        unsigned* out; unsigned long long* clk;
        hipMalloc(&out, size_t(blocks) * 256 * 4);
        hipMalloc(&clk, size_t(blocks) * 16);
        const int iters = 200000, unroll = 8;
        hipLaunchKernelGGL(spin<8>, dim3(blocks), dim3(256), 0, 0, out, clk, 1000);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<8>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(size_t(blocks) * 2);
        hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int b = 0; b < blocks; ++b) { cyc += double(h[2 * b]); real += double(h[2 * b + 1]); }
        const double insts = double(iters) * unroll * 8;                    // per wave
        const double wave_insts = insts * blocks * 4;
        std::printf("{\"waves_per_simd\": %d, \"cus\": %d, \"ms\": %.3f, \"G_wave_insts_per_s\": %.1f, \"memtime_per_memrealtime\": %.3f, "
                    "\"shader_clock_ghz_if_memtime_counts_cycles\": %.3f, \"insts_per_simd_per_memtime_tick\": %.4f}\n",
                    waves_per_simd, cus, ms, wave_insts / (ms * 1e-3) / 1e9, cyc / real, cyc / real * 0.1,
                    insts * waves_per_simd / (cyc / blocks));
        hipFree(out); hipFree(clk);
    }
    return 0;
}
