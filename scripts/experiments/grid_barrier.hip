// Grid-barrier forms for persistent kernels of one 512-thread block per CU, measured alone (no work between barriers, or a small
// sc1 store + load burst per wave as the BP kernels have): us per barrier on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o build/grid_barrier scripts/experiments/grid_barrier.hip && build/grid_barrier [blocks]
// Forms:
//   0  granule pair per block (2 x 8 bytes {generation | payload half}), the first wave of EVERY block sweeps all pairs
//      (bn_resident.hip direct form, bn_dag.hip)
//   1  one 8-byte granule per block {generation | 32-bit payload}, every block sweeps
//   2  two levels: every S-th block sweeps all granule pairs and publishes {generation | verdict} on a line of its own, the
//      other blocks of its group poll that word
//   3  as 0, but the sweeping wave waits ~1 us after its own arrival before the first poll
//   4  a SERVICE wave per block (wave 7): it polls a compact table of one 4-byte {generation} word per block with three polls in
//      flight, a new one every DELTA x 64 cycles, from the start of the iteration on; the block's other waves tell it through LDS
//      when their stores are out, it publishes the block's word and hands them the outcome through LDS (no s_barrier)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Sync {
    unsigned long long blk[2][256][2];   // [iteration parity][block]
    unsigned long long one[2][256];
    struct { unsigned long long w; unsigned long long pad[15]; } lead[2][64];
    unsigned word[2][256];
};

__device__ __forceinline__ bool sweep16(const unsigned long long* tbl, int lane, int nb, unsigned gen) {
    const unsigned voff = unsigned(lane) * 16u;
    u32x4 r0, r1, r2, r3;
    if (nb <= 64)
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(voff), "s"(tbl) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(voff), "s"(tbl) : "memory");
    bool mine = true;
    auto take = [&](const u32x4& r, int b) { if (b < nb) mine = mine && r.y == gen && r.w == gen; };
    take(r0, lane);
    if (nb > 64) { take(r1, lane + 64); take(r2, lane + 128); take(r3, lane + 192); }
    return mine;
}
__device__ __forceinline__ bool sweep8(const unsigned long long* tbl, int lane, int nb, unsigned gen) {
    const unsigned voff = unsigned(lane) * 16u;   // two blocks per lane and load
    u32x4 r0, r1;
    if (nb <= 128)
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(voff), "s"(tbl) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1) : "v"(voff), "s"(tbl) : "memory");
    bool mine = true;
    auto take = [&](const u32x4& r, int b) { if (b < nb) mine = mine && r.y == gen; if (b + 1 < nb) mine = mine && r.w == gen; };
    take(r0, 2 * lane);
    if (nb > 128) take(r1, 2 * lane + 128);
    return mine;
}

template <int FORM, int S, int WORK>
__global__ __launch_bounds__(512) void bar_kernel(Sync* sy, double* data, int iters, unsigned long long* out) {
    __shared__ int dummy;
    const int nb = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const unsigned gen = unsigned(it) + 1u;
        if (WORK) {   // a wave's share of message traffic: 2 KB stored write-through, drained
            double2* d = reinterpret_cast<double2*>(data) + (size_t(it & 1) * nb * 8 + blockIdx.x * 8 + wave) * 128 + lane * 2;
            u32x4 v; v.x = unsigned(it); v.y = 1u; v.z = unsigned(acc); v.w = 3u;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(v) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1" :: "v"(d), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FORM == 1) {
                __hip_atomic_store(&sy->one[it & 1][blockIdx.x], ((unsigned long long)gen << 32) | 7u, RLX_AGENT);
            } else {
                unsigned long long* g = sy->blk[it & 1][blockIdx.x];
                __hip_atomic_store(g, ((unsigned long long)gen << 32) | 1u, RLX_AGENT);
                __hip_atomic_store(g + 1, ((unsigned long long)gen << 32) | 2u, RLX_AGENT);
            }
        }
        if (threadIdx.x < 64) {
            if (FORM == 3) for (int z = 0; z < 30; ++z) __builtin_amdgcn_s_sleep(1);
            const bool leader = FORM != 2 || (blockIdx.x % S) == 0;
            if (leader) {
                for (;;) {
                    const bool ok = FORM == 1 ? sweep8(&sy->one[it & 1][0], lane, nb, gen) : sweep16(&sy->blk[it & 1][0][0], lane, nb, gen);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (FORM == 2 && lane == 0) __hip_atomic_store(&sy->lead[it & 1][blockIdx.x / S].w, (unsigned long long)gen, RLX_AGENT);
            } else if (lane == 0) {
                while (unsigned(__hip_atomic_load(&sy->lead[it & 1][blockIdx.x / S].w, RLX_AGENT)) != gen) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (WORK) {   // ... and read back from another block's share
            const double2* d = reinterpret_cast<const double2*>(data) + (size_t(it & 1) * nb * 8 + ((blockIdx.x + 37) % nb) * 8 + wave) * 128 + lane * 2;
            u32x4 r;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(d) : "memory");
            acc += double(r.x & 1u);
        }
    }
    if (threadIdx.x == 0) dummy = int(acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = wall_clock64() - t0; out[1] = (unsigned long long)dummy; }
}

struct SvcShared { unsigned done[8]; unsigned vgen; };
__device__ __forceinline__ const unsigned* uniform_ptr(const unsigned* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    return reinterpret_cast<const unsigned*>((unsigned long long)hi << 32 | lo);
}
#define POLL_ISSUE(R) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(R) : "v"(voff), "s"(tbl) : "memory")
#define POLL_WAIT(N, R) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R) :: "memory")
template <int WORK, int DELTA>
__global__ __launch_bounds__(512) void svc_kernel(Sync* sy, double* data, int iters, unsigned long long* out) {
    __shared__ SvcShared sh;
    __shared__ int dummy;
    const int nb = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 8) sh.done[threadIdx.x] = 0u;
    if (threadIdx.x == 0) sh.vgen = 0u;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    if (wave == 7) {
        // a ring of three polls in flight, each tagged with the iteration it was issued for; a poll that comes due for an
        // iteration already decided is dropped, nothing is ever drained
        const unsigned voff = unsigned(lane) * 16u;
        int it = 0;
        u32x4 r0, r1, r2;
        int tag0, tag1, tag2;
        auto check = [&](const u32x4& r, unsigned gen) {
            const int b = 4 * lane;
            bool ok = true;
            ok = ok && (b + 0 >= nb || r.x == gen);
            ok = ok && (b + 1 >= nb || r.y == gen);
            ok = ok && (b + 2 >= nb || r.z == gen);
            ok = ok && (b + 3 >= nb || r.w == gen);
            return __all(ok) != 0;
        };
        auto nap = [&]() { for (int z = 0; z < DELTA; ++z) __builtin_amdgcn_s_sleep(1); };
        auto decided = [&]() {
            if (lane == 0) __hip_atomic_store(&sh.vgen, unsigned(it) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++it;
        };
#define ISSUE(R, T) do { const unsigned* tbl = uniform_ptr(&sy->word[it & 1][0]); POLL_ISSUE(R); T = it; } while (0)
        ISSUE(r0, tag0); nap(); ISSUE(r1, tag1); nap(); ISSUE(r2, tag2);
        while (it < iters) {
            POLL_WAIT(2, r0); if (tag0 == it && check(r0, unsigned(it) + 1u)) decided(); if (it < iters) { ISSUE(r0, tag0); } nap();
            if (it >= iters) break;
            POLL_WAIT(2, r1); if (tag1 == it && check(r1, unsigned(it) + 1u)) decided(); if (it < iters) { ISSUE(r1, tag1); } nap();
            if (it >= iters) break;
            POLL_WAIT(2, r2); if (tag2 == it && check(r2, unsigned(it) + 1u)) decided(); if (it < iters) { ISSUE(r2, tag2); } nap();
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2) :: "memory");
    } else {
        for (int it = 0; it < iters; ++it) {
            const unsigned gen = unsigned(it) + 1u;
            if (WORK) {
                double2* d = reinterpret_cast<double2*>(data) + (size_t(it & 1) * nb * 8 + blockIdx.x * 8 + wave) * 128 + lane * 2;
                u32x4 v; v.x = unsigned(it); v.y = 1u; v.z = unsigned(acc); v.w = 3u;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(v) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1" :: "v"(d), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {   // the last of the block's seven working waves publishes the block's word
                const unsigned old = __hip_atomic_fetch_add(&sh.done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == 7u * gen - 1u) __hip_atomic_store(&sy->word[it & 1][blockIdx.x], gen, RLX_AGENT);
            }
            while (__hip_atomic_load(&sh.vgen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
            if (WORK) {
                const double2* d = reinterpret_cast<const double2*>(data) + (size_t(it & 1) * nb * 8 + ((blockIdx.x + 37) % nb) * 8 + wave) * 128 + lane * 2;
                u32x4 r;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(d) : "memory");
                acc += double(r.x & 1u);
            }
        }
    }
    if (threadIdx.x == 0) dummy = int(acc);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = wall_clock64() - t0; out[1] = (unsigned long long)dummy; }
}
template <int WORK, int DELTA>
static void run_svc(const char* name, int nb, Sync* sy, double* data, unsigned long long* out) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(sy, 0, sizeof(Sync));
        hipLaunchKernelGGL((svc_kernel<WORK, DELTA>), dim3(nb), dim3(512), 0, 0, sy, data, iters, out);
        hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s blocks %3d  work %d  %.2f us per barrier (delta %d)\n", name, nb, WORK, double(h[0]) * 10.0 / iters / 1000.0, DELTA);
}

template <int FORM, int S, int WORK>
static void run(const char* name, int nb, Sync* sy, double* data, unsigned long long* out) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(sy, 0, sizeof(Sync));
        hipLaunchKernelGGL((bar_kernel<FORM, S, WORK>), dim3(nb), dim3(512), 0, 0, sy, data, iters, out);
        hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s blocks %3d  work %d  %.2f us per barrier\n", name, nb, WORK, double(h[0]) * 10.0 / iters / 1000.0);
}

int main(int argc, char** argv) {
    Sync* sy;
    double* data;
    unsigned long long* out;
    hipMalloc(&sy, sizeof(Sync));
    hipMalloc(&data, size_t(2) * 256 * 8 * 128 * 16);
    hipMalloc(&out, 16);
    std::vector<int> nbs;
    if (argc > 1) nbs.push_back(atoi(argv[1])); else nbs = {64, 128, 192, 224};
    for (int nb : nbs) {
        run<0, 1, 0>("pairs, every block sweeps", nb, sy, data, out);
        run<1, 1, 0>("8-byte granules, every block sweeps", nb, sy, data, out);
        run<2, 4, 0>("pairs, every 4th block sweeps, others poll its word", nb, sy, data, out);
        run<2, 8, 0>("pairs, every 8th block sweeps, others poll its word", nb, sy, data, out);
        run<2, 32, 0>("pairs, every 32nd block sweeps", nb, sy, data, out);
        run<3, 1, 0>("pairs, every block sweeps, first poll ~1 us late", nb, sy, data, out);
        run<0, 1, 1>("pairs, every block sweeps", nb, sy, data, out);
        run<1, 1, 1>("8-byte granules, every block sweeps", nb, sy, data, out);
        run<2, 8, 1>("pairs, every 8th block sweeps, others poll its word", nb, sy, data, out);
        run_svc<0, 2>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
        run_svc<0, 5>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
        run_svc<0, 10>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
        run_svc<1, 2>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
        run_svc<1, 5>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
        run_svc<1, 10>("service wave, 4-byte words, 3 polls in flight", nb, sy, data, out);
    }
    return 0;
}
