// stream_probe.hip -- which form of a copy / triad reaches the most on this box (round 6: the yardstick of bn_debug_stream).
// hipcc --offload-arch=gfx950 -O3 scripts/experiments/stream_probe.hip -o build/stream_probe && build/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// grid-stride, U loads in flight; NT: 0 plain, 1 nt loads + nt stores, 2 nt stores only
template <int U, int NT, int MODE>
__global__ __launch_bounds__(256) void k_stride(v2d* __restrict__ d, const v2d* __restrict__ a, const v2d* __restrict__ b, size_t n) {
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        v2d x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = NT == 1 ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
            if (MODE) y[u] = NT == 1 ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2d r = MODE ? x[u] + 0.5 * y[u] : x[u];
            if (NT) __builtin_nontemporal_store(r, d + i + u * stride); else d[i + u * stride] = r;
        }
    }
}
// block-contiguous chunks: block b owns [b * chunk, (b + 1) * chunk), walks it 256 * U elements at a time
template <int U, int NT, int MODE>
__global__ __launch_bounds__(256) void k_chunk(v2d* __restrict__ d, const v2d* __restrict__ a, const v2d* __restrict__ b, size_t n) {
    const size_t chunk = n / gridDim.x;
    const size_t base = size_t(blockIdx.x) * chunk;
    for (size_t o = threadIdx.x; o + (U - 1) * 256 < chunk; o += U * 256) {
        v2d x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = NT == 1 ? __builtin_nontemporal_load(a + base + o + u * 256) : a[base + o + u * 256];
            if (MODE) y[u] = NT == 1 ? __builtin_nontemporal_load(b + base + o + u * 256) : b[base + o + u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2d r = MODE ? x[u] + 0.5 * y[u] : x[u];
            if (NT) __builtin_nontemporal_store(r, d + base + o + u * 256); else d[base + o + u * 256] = r;
        }
    }
}
__global__ void k_fill(v2d* p, size_t n, double v) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) p[i] = v2d{v, v};
}

template <typename F>
static double best_of(F launch, hipStream_t st, double bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0, st);
        launch();
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) best = std::max(best, bytes / (ms * 1e-3) / 1e9);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    for (size_t mib : {512, 1024, 2048}) {
        const size_t n = mib * (size_t(1) << 20) / 16;
        v2d *d, *a, *b;
        CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
        k_fill<<<2048, 256, 0, st>>>(a, n, 1.0); k_fill<<<2048, 256, 0, st>>>(b, n, 2.0); k_fill<<<2048, 256, 0, st>>>(d, n, 0.0);
        CK(hipStreamSynchronize(st));
        printf("== %zu MiB per array\n", mib);
        for (int grid : {256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
            printf("grid %6d copy: ", grid);
#define RUN(K, U, NT, MODE) printf(#K "<" #U "," #NT "> %6.0f  ", best_of([&] { K<U, NT, MODE><<<grid, 256, 0, st>>>(d, a, b, n); }, st, double(n) * 16 * (MODE ? 3 : 2)))
            RUN(k_stride, 1, 0, 0); RUN(k_stride, 4, 0, 0); RUN(k_stride, 4, 1, 0); RUN(k_stride, 4, 2, 0); RUN(k_stride, 8, 1, 0);
            RUN(k_chunk, 4, 0, 0); RUN(k_chunk, 4, 1, 0); RUN(k_chunk, 8, 2, 0);
            printf("\n            triad: ");
            RUN(k_stride, 4, 0, 1); RUN(k_stride, 4, 1, 1); RUN(k_stride, 4, 2, 1); RUN(k_chunk, 4, 1, 1); RUN(k_chunk, 4, 2, 1);
            printf("\n");
        }
        // hipMemcpyAsync D2D for comparison
        printf("hipMemcpyAsync D2D %6.0f\n", best_of([&] { hipMemcpyAsync(d, a, n * 16, hipMemcpyDeviceToDevice, st); }, st, double(n) * 32));
        hipFree(d); hipFree(a); hipFree(b);
    }
    return 0;
}
