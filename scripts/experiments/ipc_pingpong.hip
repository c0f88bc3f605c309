// Two PROCESSES on one GPU, each running a kernel that waits for the other: do their kernels run at the same time, do
// hipIpc handles of a fine-grained allocation work, and what does one flag hand-off between them cost?
// (Feasibility of the peer-mapped halo exchange inside the resident kernel)
// Build: hipcc --offload-arch=gfx950 -O2 -o build/ipc_pingpong scripts/experiments/ipc_pingpong.hip
// Run:   build/ipc_pingpong            (forks the second process itself BEFORE any HIP call)
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("[%d] HIP error %s at line %d\n", me, hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

// every wait is bounded by the 100 MHz wall clock: a kernel that cannot see its partner gives up
__global__ void pingpong(unsigned long long* mine, unsigned long long* theirs, int me, int rounds, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned long long t_first = 0;
    int done = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {  // rank 0 serves, rank 1 returns
            __hip_atomic_store(theirs, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long w0 = wall_clock64();
        bool ok = false;
        while (wall_clock64() - w0 < 200000000ull) {  // 2 s
            if (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned long long)r) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) break;
        if (r == 1) t_first = wall_clock64();
        if (me == 1) __hip_atomic_store(theirs, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        done = r;
    }
    out[0] = done;
    out[1] = wall_clock64() - t0;
    out[2] = t_first ? wall_clock64() - t_first : 0;
}

int main(int argc, char** argv) {
    int pipe_ab[2], pipe_ba[2];
    if (pipe(pipe_ab) || pipe(pipe_ba)) return 1;
    const pid_t child = fork();  // before any HIP call
    const int me = child == 0 ? 1 : 0;
    const int rd = me == 0 ? pipe_ba[0] : pipe_ab[0], wr = me == 0 ? pipe_ab[1] : pipe_ba[1];
    const bool fine = argc > 1 ? std::atoi(argv[1]) != 0 : true;
    CK(hipSetDevice(0));
    unsigned long long* mine = nullptr;
    if (fine) CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&mine), 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc(reinterpret_cast<void**>(&mine), 4096));
    CK(hipMemset(mine, 0, 4096));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h_mine, h_theirs;
    CK(hipIpcGetMemHandle(&h_mine, mine));
    if (write(wr, &h_mine, sizeof h_mine) != (ssize_t)sizeof h_mine) return 3;
    if (read(rd, &h_theirs, sizeof h_theirs) != (ssize_t)sizeof h_theirs) return 3;
    unsigned long long* theirs = nullptr;
    CK(hipIpcOpenMemHandle(reinterpret_cast<void**>(&theirs), h_theirs, hipIpcMemLazyEnablePeerAccess));
    unsigned long long* out = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&out), 64, hipHostMallocMapped));
    std::memset(out, 0, 64);
    // both sides ready
    char c = 'r';
    if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 3;
    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, mine, theirs, me, rounds, out);
    CK(hipDeviceSynchronize());
    std::printf("[%d] %s memory: %llu of %d rounds, %.1f us total, %.3f us per round trip after the first\n", me,
                fine ? "fine-grained" : "coarse-grained", out[0], rounds, out[1] * 0.01, out[0] > 1 ? out[2] * 0.01 / (out[0] - 1) : 0.0);
    if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 3;  // nobody unmaps while the other still runs
    CK(hipIpcCloseMemHandle(theirs));
    CK(hipFree(mine));
    if (me == 0) { int st = 0; waitpid(child, &st, 0); return WEXITSTATUS(st); }
    return 0;
}
