// ipc_probe.hip -- development probe (not product): which cross-process primitives work on the box?
//   N processes (forked before any HIP call) share ONE device (or use device rank % ndev):
//   hipIpc handles of plain and fine-grained allocations, peer reads/writes from kernels, and a
//   device-side flag hand-off (producer kernel -> signal kernel -> peer's wait kernel -> consumer
//   kernel) with bounded spins, all enqueued without host synchronisation.
// build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip ; run: ./ipc_probe [nproc] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)
static int g_rank = -1;
enum { MAXP = 8 };
struct Shared {
    std::atomic<int> arrive[64];
    hipIpcMemHandle_t data[MAXP], flags[MAXP];
    int ok_fine[MAXP];
};
static void barrier(Shared *s, int idx, int n)
{
    s->arrive[idx].fetch_add(1);
    auto t0 = std::chrono::steady_clock::now();
    while (s->arrive[idx].load() < n) {
        usleep(50);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { printf("[%d] host barrier %d timeout\n", g_rank, idx); fflush(stdout); _exit(4); }
    }
}

__global__ void k_fill(float *p, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023);
}
__global__ void k_check(const float *p, size_t n, float v, unsigned *bad)
{
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v + (float)(i & 1023)) atomicAdd(bad, 1u);
}
// one wave: lane p stores `value` into word `slot` of peer p's flag array (system scope, after a system release)
__global__ void k_signal(unsigned *const *peer_flags, int n, int slot, unsigned value)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int p = threadIdx.x;
    if (p < n) __hip_atomic_store(peer_flags[p] + slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one wave: lane p waits until flags[base + p] >= target (bounded); err gets a code on timeout
__global__ void k_wait(const unsigned *flags, int base, int n, unsigned target, unsigned long long max_ticks, unsigned *err)
{
    const int p = threadIdx.x;
    if (p < n) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const unsigned v = __hip_atomic_load(flags + base + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - target) >= 0) break;
            if (wall_clock64() - t0 > max_ticks) { atomicExch(err, 1000u + (unsigned)p); break; }
            __builtin_amdgcn_s_sleep(20);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

static int child(Shared *s, int rank, int n, int iters)
{
    g_rank = rank;
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    const int dev = rank % ndev;
    CK(hipSetDevice(dev));
    const size_t N = 4u << 20;   // 16 MB of floats
    float *data; unsigned *flags, *bad, *err;
    CK(hipMalloc((void **)&data, N * 4));
    hipError_t ef = hipExtMallocWithFlags((void **)&flags, 4096, hipDeviceMallocFinegrained);
    s->ok_fine[rank] = ef == hipSuccess;
    if (ef != hipSuccess) { printf("[%d] fine-grained alloc failed (%s), using hipMalloc\n", rank, hipGetErrorString(ef)); CK(hipMalloc((void **)&flags, 4096)); }
    CK(hipMemset(flags, 0, 4096));
    CK(hipMalloc((void **)&bad, 8)); err = bad + 1;
    CK(hipMemset(bad, 0, 8));
    CK(hipIpcGetMemHandle(&s->data[rank], data));
    hipError_t eh = hipIpcGetMemHandle(&s->flags[rank], flags);
    if (eh != hipSuccess) { printf("[%d] hipIpcGetMemHandle(fine-grained) -> %s\n", rank, hipGetErrorString(eh)); fflush(stdout); _exit(5); }
    barrier(s, 0, n);
    float *pdata[MAXP]; unsigned *pflags[MAXP];
    for (int p = 0; p < n; ++p) {
        if (p == rank) { pdata[p] = data; pflags[p] = flags; continue; }
        CK(hipIpcOpenMemHandle((void **)&pdata[p], s->data[p], hipIpcMemLazyEnablePeerAccess));
        CK(hipIpcOpenMemHandle((void **)&pflags[p], s->flags[p], hipIpcMemLazyEnablePeerAccess));
    }
    unsigned **d_pflags; CK(hipMalloc((void **)&d_pflags, sizeof(void *) * MAXP));
    CK(hipMemcpy(d_pflags, pflags, sizeof(void *) * n, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // ---- test 1: host-synchronised peer read
    hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, st, data, N, (float)(100 * rank));
    CK(hipStreamSynchronize(st));
    barrier(s, 1, n);
    for (int p = 0; p < n; ++p) hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, st, pdata[p], N, (float)(100 * p), bad);
    CK(hipStreamSynchronize(st));
    unsigned hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("[%d] dev %d test1 peer reads: %u bad (fine-grained flags: %d)\n", rank, dev, hb[0], s->ok_fine[rank]); fflush(stdout);
    barrier(s, 2, n);
    // ---- test 2: device-side hand-off ring, no host sync inside: iteration it: fill own data with value(it, rank),
    // signal slot[rank] = it on every peer; wait for every peer's slot >= it; check every peer's data == value(it, p);
    // then signal slot[MAXP + rank] = it (ack: "I finished reading") and wait for all acks before the next fill.
    const unsigned long long max_ticks = 100000000ull * 5;   // 5 s at 100 MHz
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 1; it <= iters; ++it) {
        hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, st, data, N, (float)(it * 1000 + rank));
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, d_pflags, n, rank, (unsigned)it);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, flags, 0, n, (unsigned)it, max_ticks, err);
        for (int p = 0; p < n; ++p) hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, st, pdata[p], N, (float)(it * 1000 + p), bad);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, d_pflags, n, MAXP + rank, (unsigned)it);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, flags, MAXP, n, (unsigned)it, max_ticks, err);
    }
    CK(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("[%d] test2 device-side hand-off x%d: %u bad, err %u, %.3f ms/iter (%zu MB read per iter)\n", rank, iters, hb[0], hb[1], ms / iters, n * N * 4 >> 20);
    fflush(stdout);
    barrier(s, 3, n);
    for (int p = 0; p < n; ++p) if (p != rank) { CK(hipIpcCloseMemHandle(pdata[p])); CK(hipIpcCloseMemHandle(pflags[p])); }
    barrier(s, 4, n);
    return (hb[0] || hb[1]) ? 1 : 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 50;
    if (n < 1 || n > MAXP) return 2;
    Shared *s = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset((void *)s, 0, sizeof(Shared));
    pid_t pids[MAXP];
    for (int r = 0; r < n; ++r) {
        pids[r] = fork();
        if (pids[r] == 0) _exit(child(s, r, n, iters));
    }
    int rc = 0;
    for (int r = 0; r < n; ++r) { int st = 0; waitpid(pids[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
    printf("ipc_probe n=%d: %s\n", n, rc ? "FAIL" : "OK");
    return rc;
}
