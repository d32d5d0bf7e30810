// tools/direct_probe.hip -- development probe: LDS-free, barrier-free "register-direct" fp32 MFMA GEMM for the
// M=256 forward shape (C[256][N] = A[256][K] . B[K][N], A k-contiguous, B n-contiguous).
// Idea: the MFMA's k index and its n (column) index are dummy labels, so each lane can load its operands
// straight from global memory in whatever assignment coalesces:
//   A (k-contiguous rows):  lane (m = l&31, h = l>>5) loads 8 consecutive k of row m  -> k(s,h) = kb + 8h + s, s = 0..7
//   B (n-contiguous rows):  lane (j = l&31, h)        loads B[k(s,h)][n0 + 2j .. 2j+1] -> two column blocks c: n = n0 + 2j + c
// 16 MFMAs per 2 dwordx4 + 8 dwordx2 loads, no LDS traffic, no barriers in the k-loop.  4 waves of a workgroup split K
// and reduce through LDS at the end.   usage: direct_probe [NST]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NST>
__global__ __launch_bounds__(256, 1) void direct_fwd(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                     float *__restrict__ C, int ldc, int K, int tiles_m, int tiles_n)
{
    __shared__ float red[4 * 2 * 16 * 64];                       // 32 KB: 4 waves x 2 blocks x 16 regs x 64 lanes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, xcd = b & 7, jj = b >> 3, per = tiles_n >> 3;
    const int tile_n = xcd * per + jj / tiles_m, tile_m = jj % tiles_m;
    const int m0 = tile_m * 32, n0 = tile_n * 64;
    const int j = lane & 31, h = lane >> 5;
    const int kw = wave * (K / 4);                               // this wave's k range [kw, kw + K/4)
    const float *pa = A + (size_t)(m0 + j) * lda + kw + 8 * h;   // + 16 t
    const float *pb = B + (size_t)(kw + 8 * h) * ldb + n0 + 2 * j;   // + (16 t + s) * ldb
    const int nt = K / 4 / 16;

    f32x16 acc[2][2];                                            // [chain][column block]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;

    float4 ra[NST][2];
    f32x2 rb[NST][8];
#define LOAD(st, t)                                                                                    \
    do {                                                                                               \
        const float *qa = pa + 16 * (t);                                                               \
        ra[st][0] = *reinterpret_cast<const float4 *>(qa);                                             \
        ra[st][1] = *reinterpret_cast<const float4 *>(qa + 4);                                         \
        const float *qb = pb + (size_t)(16 * (t)) * ldb;                                               \
        _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                  \
            rb[st][s] = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(qb + (size_t)s * ldb)); \
    } while (0)
#define COMPUTE(st)                                                                                    \
    do {                                                                                               \
        const float a8[8] = {ra[st][0].x, ra[st][0].y, ra[st][0].z, ra[st][0].w, ra[st][1].x, ra[st][1].y, ra[st][1].z, ra[st][1].w}; \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                \
            acc[s & 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a8[s], rb[st][s].x, acc[s & 1][0], 0, 0, 0); \
            acc[s & 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a8[s], rb[st][s].y, acc[s & 1][1], 0, 0, 0); \
        }                                                                                              \
    } while (0)

    // prologue: NST-1 macro-steps in flight
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) LOAD(p, p < nt ? p : nt - 1);
    int t = 0;
    for (; t + NST <= nt; t += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int tl = t + u + NST - 1;
            LOAD((u + NST - 1) % NST, tl < nt ? tl : nt - 1);    // unconditional (clamped) prefetch
            COMPUTE(u);
        }
    }
    // (nt is a multiple of NST in this probe)

    // ---- reduce the 4 k-slices through LDS, wave w finishes registers [4w, 4w+4) of both blocks
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 2 + d) * 16 + r) * 64 + lane] = acc[0][d][r] + acc[1][d][r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { v0 += red[((w * 2 + 0) * 16 + r) * 64 + lane]; v1 += red[((w * 2 + 1) * 16 + r) * 64 + lane]; }
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        *reinterpret_cast<float2 *>(C + (size_t)m * ldc + n0 + 2 * j) = make_float2(v0, v1);
    }
}

int main(int argc, char **argv)
{
    const int M = 256, N = 2048, K = 2048;
    const int LD = argc > 1 ? atoi(argv[1]) : 2048;
    printf("LD = %d\n", LD);
    std::vector<float> hA((size_t)M * LD), hB((size_t)K * LD), hC((size_t)M * LD);
    srand(1);
    for (auto &v : hA) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hB) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.03f;
    float *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 4 + 65536)); CK(hipMalloc(&B, hB.size() * 4 + 65536)); CK(hipMalloc(&C, hC.size() * 4 + 65536));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int nst) {
        if (nst == 2) hipLaunchKernelGGL(direct_fwd<2>, dim3(256), dim3(256), 0, st, A, LD, B, LD, C, LD, K, M / 32, N / 64);
        else if (nst == 4) hipLaunchKernelGGL(direct_fwd<4>, dim3(256), dim3(256), 0, st, A, LD, B, LD, C, LD, K, M / 32, N / 64);
        else hipLaunchKernelGGL(direct_fwd<8>, dim3(256), dim3(256), 0, st, A, LD, B, LD, C, LD, K, M / 32, N / 64);
    };
    for (int nst : {2, 4, 8}) {
        CK(hipMemset(C, 0, hC.size() * 4));
        for (int i = 0; i < 20; ++i) run(nst);
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) run(nst);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 20 * 1000.f);
        }
        CK(hipGetLastError());
        std::sort(ts.begin(), ts.end());
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int q = 0; q < 4000; ++q) {
            const int m = rand() % M, n = rand() % N;
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * LD + k] * hB[(size_t)k * LD + n];
            maxerr = std::max(maxerr, std::fabs(s - (double)hC[(size_t)m * LD + n]));
            if (q == 0) printf("  sample: ref %.9f got %.9f\n", s, (double)hC[(size_t)m * LD + n]); maxref = std::max(maxref, std::fabs(s));
        }
        printf("direct fwd NST=%d: med %.2f us min %.2f us  %.1f TF  (max err %.2e / max |ref| %.2e)\n", nst, ts[ts.size() / 2], ts[0],
               2.0 * M * N * K / ts[ts.size() / 2] * 1e-6, maxerr, maxref);
    }
    return 0;
}
