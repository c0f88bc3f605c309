// bn_stream.hip -- the achievable-HBM yardstick SURVEY 8(d) asks for beside the nominal peak: a 16-byte-per-lane copy and a triad
// over arrays far beyond the 256 MiB Infinity Cache, non-temporal on both sides, timed with HIP events on a stream of its own.
// Not on any product path: bench.py reports its figure next to the roofline fractions (bn_debug_stream).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/bn_mi355x.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kThreads = 256;
constexpr int kUnroll = 4;   // 16-byte loads in flight per lane before the first store

// mode 0: dst = src (one read, one write per element); mode 1: dst = a + s * b (two reads, one write).
// One workgroup = one contiguous run of kUnroll x 256 elements (16 KiB per array), all of its loads in flight before the first store,
// no loop: of the forms scripts/experiments/stream_probe.hip compares (grid-stride with 256 .. 65 536 workgroups, contiguous chunks
// per workgroup, plain / non-temporal on either side) this one reaches the most on an MI355X -- 6.1-6.3 TB/s copy, 6.0-6.5 triad on
// 1-2 GiB arrays; persistent grid-stride workgroups stay at 4.7-5.4, hipMemcpyAsync and torch's copy_ at 5.0-5.6.
template <int MODE>
__global__ __launch_bounds__(kThreads) void stream_kernel(v2d* __restrict__ dst, const v2d* __restrict__ a, const v2d* __restrict__ b,
                                                          double s, size_t n) {
    const size_t base = size_t(blockIdx.x) * (kUnroll * kThreads) + threadIdx.x;
    if (base + (kUnroll - 1) * kThreads < n) {
        v2d x[kUnroll], y[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            x[u] = __builtin_nontemporal_load(a + base + u * kThreads);
            if (MODE == 1) y[u] = __builtin_nontemporal_load(b + base + u * kThreads);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            v2d r = x[u];
            if (MODE == 1) r = x[u] + s * y[u];
            __builtin_nontemporal_store(r, dst + base + u * kThreads);
        }
        return;
    }
    for (size_t i = base; i < n; i += kThreads) {   // the last, partial run
        v2d r = __builtin_nontemporal_load(a + i);
        if (MODE == 1) r = r + s * __builtin_nontemporal_load(b + i);
        __builtin_nontemporal_store(r, dst + i);
    }
}

__global__ __launch_bounds__(kThreads) void stream_fill_kernel(v2d* p, size_t n, double v) {
    for (size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += size_t(gridDim.x) * kThreads) p[i] = v2d{v, v + 1.0};
}

}  // namespace

// -> bytes moved per second / 1e9 (best of `reps`), or a negative hipError_t.  `bytes` = size of ONE array.
double bn_stream_measure(int mode, size_t bytes, int reps, double* check_out) {
    const size_t n = bytes / sizeof(v2d);
    v2d *d = nullptr, *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double best = 0.0;
    hipError_t rc = hipSuccess;
    auto done = [&](hipError_t e) { rc = e; return e != hipSuccess; };
    do {
        if (done(hipMalloc(&d, n * sizeof(v2d))) || done(hipMalloc(&a, n * sizeof(v2d)))) break;
        if (mode == 1 && done(hipMalloc(&b, n * sizeof(v2d)))) break;
        if (done(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) || done(hipEventCreate(&e0)) || done(hipEventCreate(&e1))) break;
        const int grid = int((n + kUnroll * kThreads - 1) / (kUnroll * kThreads));
        stream_fill_kernel<<<2048, kThreads, 0, st>>>(a, n, 1.0);
        if (b) stream_fill_kernel<<<2048, kThreads, 0, st>>>(b, n, 2.0);
        for (int r = 0; r < reps + 1; ++r) {   // (the first repetition also pages the code object in: not counted)
            if (done(hipEventRecord(e0, st))) break;
            if (mode == 1) stream_kernel<1><<<grid, kThreads, 0, st>>>(d, a, b, 0.5, n);
            else stream_kernel<0><<<grid, kThreads, 0, st>>>(d, a, nullptr, 0.0, n);
            if (done(hipEventRecord(e1, st)) || done(hipEventSynchronize(e1))) break;
            float ms = 0.f;
            if (done(hipEventElapsedTime(&ms, e0, e1))) break;
            if (r > 0 && ms > 0.f) best = std::max(best, double(n) * sizeof(v2d) * (mode == 1 ? 3 : 2) / (double(ms) * 1e-3) / 1e9);
        }
        if (rc == hipSuccess && check_out) {   // one element back: the kernel did what it says
            v2d h;
            if (!done(hipMemcpy(&h, d + (n - 1), sizeof h, hipMemcpyDeviceToHost))) *check_out = h[0] + h[1];
        }
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(d);
    (void)hipFree(a);
    (void)hipFree(b);
    return rc == hipSuccess ? best : -double(int(rc));
}
