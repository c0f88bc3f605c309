// Does global_load_lds reach LDS addresses beyond 64 KB on gfx950 (160 KB of LDS)?  One wave copies 1 KB from global memory to LDS
// offsets 0, 48 KB, 70 KB, 100 KB, 140 KB and reads it back; also a lane-masked copy (does a masked-off lane's slot stay untouched,
// and do the active lanes land at 16 * lane or compacted?).
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/glds_high.hip -o build/glds_high && build/glds_high
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
__global__ __launch_bounds__(64) void k(const unsigned* src, unsigned* out) {
    __shared__ uint4 lds[9216];  // 144 KB
    const int lane = threadIdx.x;
    for (int i = lane; i < 9216; i += 64) lds[i] = make_uint4(0xdeadu, 0, 0, 0);
    __syncthreads();
    const int offs[5] = {0, 3072, 4480, 6400, 8960};   // in 16-byte units: 0, 48 KB, 70 KB, 100 KB, 140 KB
    for (int t = 0; t < 5; ++t) {
        __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(src) + t * 1024 + lane * 16), (lptr)(lds + offs[t]), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < 5; ++t) out[t * 64 + lane] = lds[offs[t] + lane].x;
    // masked: only odd lanes, into offset 16 KB (1024 units)
    if (lane & 1) __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(src) + 5 * 1024 + lane * 16), (lptr)(lds + 1024), 16, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[5 * 64 + lane] = lds[1024 + lane].x;
}
int main() {
    unsigned h[6 * 256];
    for (int i = 0; i < 6 * 256; ++i) h[i] = i;   // word i; lane l of block t reads word t * 256 + 4 l
    unsigned *d, *o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 6 * 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    unsigned r[6 * 64];
    hipError_t e = hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("status %s\n", hipGetErrorString(e));
    for (int t = 0; t < 6; ++t) {
        int ok = 0;
        for (int l = 0; l < 64; ++l) ok += r[t * 64 + l] == unsigned(t * 256 + 4 * l);
        printf("copy %d: %d of 64 lanes as expected; lanes 0..3: %x %x %x %x\n", t, ok, r[t * 64], r[t * 64 + 1], r[t * 64 + 2], r[t * 64 + 3]);
    }
    return 0;
}
