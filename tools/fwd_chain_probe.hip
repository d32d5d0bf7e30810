// tools/fwd_chain_probe.hip -- development probe (round 4): the hidden layers' forward GEMMs as ONE launch with a
// per-m-tile hand-off instead of two launches.  Layer A = 256 x 2048 x 2048 (bp_gemm<32,64,64,1,2> body, bias + ReLU + Philox
// dropout epilogue), layer B the same on A's output.  In the fused launch every workgroup computes its (m-tile, n-tile) of layer A,
// publishes it, waits until the 32 workgroups of ITS m-tile have published (8 groups of 32, not grid-wide), and computes the
// same tile of layer B.  Two publish forms:
//   mode 1  plain epilogue stores -> every wave drains -> barrier -> one lane: agent-scope release fence + counter;
//           consumer: one lane polls, agent-scope acquire, barrier                      (the guide's Guideline 16 counter form)
//   mode 2  the same without the release fence (stores drained only): NOT a valid publish across XCDs -- timing lower bound only
// Results of the fused launches are compared with the two-launch result word for word.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fwd_chain_probe.hip -o tools/bin/fwd_chain_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

using KF = GemmKernel<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN>;

template <int MODE>
__global__ __launch_bounds__(256, KF::MIN_WG) void fwd_chain2(const GemmArgs g1, const EpiArgs e1, const GemmArgs g2, const EpiArgs e2, unsigned *cnt, unsigned epoch,
                                                               unsigned long long budget, unsigned *err)
{
    __shared__ __attribute__((aligned(16))) float smem[KF::SMEM];
    KF::run(g1, e1, blockIdx.x, gridDim.x, 0, smem);
    // which m-tile this workgroup holds (the XCD-aware map of GemmKernel::run, forward form)
    const int b = blockIdx.x, j = b >> 3;
    const int tile_m = (g1.tiles_n & 7) == 0 ? j % g1.tiles_m : b % g1.tiles_m;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(cnt + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)g1.tiles_n;
        const unsigned long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(cnt + tile_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > budget) { atomicExch(err, 1u + (unsigned)tile_m); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    KF::run(g2, e2, blockIdx.x, gridDim.x, 0, smem);
}

static unsigned long long rng_s = 0x9E3779B97F4A7C15ull;
static float frand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((double)(rng_s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
static float *dalloc_rand(size_t n, float scale)
{
    std::vector<float> h(n + 16384, 0.f);
    for (size_t i = 0; i < n; ++i) h[i] = scale * frand();
    float *d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
}

int main()
{
    const int B = 256, H = 2048;
    float *Y0 = dalloc_rand((size_t)B * H, 1.0f), *W1 = dalloc_rand((size_t)H * H, 0.03f), *W2 = dalloc_rand((size_t)H * H, 0.03f), *b1 = dalloc_rand(H, 0.1f), *b2 = dalloc_rand(H, 0.1f);
    float *Y1a = dalloc_rand((size_t)B * H, 0.f), *Y2a = dalloc_rand((size_t)B * H, 0.f), *Y1b = dalloc_rand((size_t)B * H, 0.f), *Y2b = dalloc_rand((size_t)B * H, 0.f);
    auto mk = [&](const float *A, const float *W, const float *bias, float *C, GemmArgs &g, EpiArgs &e, unsigned layer) {
        memset(&g, 0, sizeof(g)); memset(&e, 0, sizeof(e));
        g.A = A; g.lda = H; g.B = W; g.ldb = H; g.K = H; g.tiles_m = B / 32; g.tiles_n = H / 64;
        e.alpha = 1.f; e.C = C; e.ldc = H; e.m_limit = B; e.n_limit = H; e.n_true = H; e.bias = bias; e.act = 0;
        e.drop_thresh = 858993459u; e.seed_lo = 7; e.seed_hi = 0; e.step = 3; e.layer = layer;
    };
    GemmArgs g1a, g2a, g1b, g2b; EpiArgs e1a, e2a, e1b, e2b;
    mk(Y0, W1, b1, Y1a, g1a, e1a, 1); mk(Y1a, W2, b2, Y2a, g2a, e2a, 2);       // two launches
    mk(Y0, W1, b1, Y1b, g1b, e1b, 1); mk(Y1b, W2, b2, Y2b, g2b, e2b, 2);       // one fused launch
    unsigned *cnt, *err;
    CK(hipMalloc(&cnt, 64 * 4)); CK(hipMemset(cnt, 0, 64 * 4));
    CK(hipHostMalloc((void **)&err, 4, hipHostMallocMapped)); *err = 0;
    const int NWG = g1a.tiles_m * g1a.tiles_n;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned epoch = 0;
    auto two = [&]() {
        hipLaunchKernelGGL((bp_gemm<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN>), dim3(NWG), dim3(256), 0, st, g1a, e1a);
        hipLaunchKernelGGL((bp_gemm<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN>), dim3(NWG), dim3(256), 0, st, g2a, e2a);
    };
    auto fused = [&](int mode) {
        ++epoch;
        if (mode == 1) hipLaunchKernelGGL((fwd_chain2<1>), dim3(NWG), dim3(256), 0, st, g1b, e1b, g2b, e2b, cnt, epoch, 100000000ull, err);
        else hipLaunchKernelGGL((fwd_chain2<2>), dim3(NWG), dim3(256), 0, st, g1b, e1b, g2b, e2b, cnt, epoch, 100000000ull, err);
    };
    for (int mode = 1; mode <= 2; ++mode) {
        CK(hipMemset(Y1b, 0, (size_t)B * H * 4)); CK(hipMemset(Y2b, 0, (size_t)B * H * 4));
        two(); fused(mode);
        CK(hipStreamSynchronize(st));
        if (*err) { printf("mode %d: hand-off timed out (m-tile %u)\n", mode, *err - 1); return 1; }
        std::vector<float> a((size_t)B * H), c((size_t)B * H);
        size_t bad = 0;
        for (int rep = 0; rep < 20; ++rep) {             // race screen: the fused launch again and again, every word each time
            fused(mode); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(a.data(), Y2a, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), Y2b, c.size() * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < a.size(); ++k) if (a[k] != c[k]) ++bad;
        }
        printf("mode %d: layer-B output of the fused launch vs two launches: %zu differing words over 20 launches\n", mode, bad);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t[3];
    for (int round = 0; round < 10; ++round)
        for (int which = 0; which < 3; ++which) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < 100; ++it) { if (which == 0) two(); else fused(which); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round >= 2) t[which].push_back(ms / 100 * 1000);
        }
    for (auto &v : t) std::sort(v.begin(), v.end());
    printf("TIMING two launches: median %.2f us (min %.2f) | fused, release + acquire: %.2f (min %.2f) | fused, drained stores only (not a valid publish): %.2f (min %.2f)\n",
           t[0][t[0].size() / 2], t[0][0], t[1][t[1].size() / 2], t[1][0], t[2][t[2].size() / 2], t[2][0]);
    return 0;
}
