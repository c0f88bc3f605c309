// Inter-kernel gap of a dependent chain of small kernels: plain stream launches vs. one hipGraph replay.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/graph_gap scripts/experiments/graph_gap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void work(double* p, int n, int spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double x = p[i];
        for (int k = 0; k < spin; ++k) x = x * 1.0000001 + 1e-9;
        p[i] = x;
    }
}

int main() {
    const int n = 1 << 16, chain = 16, reps = 200;
    double* d;
    CK(hipMalloc(&d, n * sizeof(double)));
    CK(hipMemset(d, 0, n * sizeof(double)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int spin : {1, 200, 1000}) {
        auto launch_chain = [&]() { for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, spin); };
        for (int w = 0; w < 5; ++w) launch_chain();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { launch_chain(); CK(hipStreamSynchronize(s)); }
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch_chain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // one kernel alone, for the floor
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, d, n, spin); CK(hipStreamSynchronize(s)); }
        double us_one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("spin %4d: chain of %d  stream %.1f us (%.2f per kernel)  graph %.1f us (%.2f per kernel)  single launch+sync %.1f us\n",
                    spin, chain, us_stream, us_stream / chain, us_graph, us_graph / chain, us_one);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
