// tools/lds_feed_probe.hip -- development probe (not part of the product): does feeding v_mfma_f32_32x32x2_f32 from LDS cost matrix-pipe
// time by itself?  Each workgroup = 4 waves, each wave issues N MFMAs on 2 accumulator chains; the operand registers come from
//   mode 0: nowhere (constant registers: the bare pipe)
//   mode 1: 2 x ds_read_b32 per MFMA (what every fp32 GEMM of the library does: one A and one B value per lane per MFMA), 4 MFMAs ahead
//   mode 2: 2 x ds_read_b64 per 2 MFMAs      mode 3: 2 x ds_read_b128 per 4 MFMAs   (same bytes, fewer LDS instructions)
//   mode 4: 1 x ds_read_b32 per MFMA (A reused by both chains: a 32x64 wave tile)
//   mode 5: mode 1 + one s_barrier per 16 MFMAs (a 64-deep k-tile with the in-workgroup k-split of the shipped forward GEMM)
// No global loads, no LDS writes after the fill.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/lds_feed_probe tools/lds_feed_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void feed(float *out, int n)
{
    __shared__ __attribute__((aligned(16))) float smem[8192];          // 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) smem[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    const float *ap = smem + (lane & 31) + (lane >> 5) * 64, *bp = smem + 4096 + (lane & 31) + (lane >> 5) * 64;
    if constexpr (MODE == 0) {
        float a = tid * 1e-3f, b = blockIdx.x * 1e-3f;
        for (int i = 0; i < n; i += 2) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
        }
    } else if constexpr (MODE == 1 || MODE == 5) {
        for (int t = 0; t < n; t += 16) {                   // one "k-tile" of 16 MFMAs per wave, k-rows 128 floats apart
            float av[16], bv[16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) { av[s] = ap[s * 128]; bv[s] = bp[s * 128]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[s & 1], 0, 0, 0);
                if (s + 4 < 16) { av[s + 4] = ap[(s + 4) * 128]; bv[s + 4] = bp[(s + 4) * 128]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 5) __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (MODE == 2) {
        const f32x2 *a2 = (const f32x2 *)(smem + 2 * lane), *b2 = (const f32x2 *)(smem + 4096 + 2 * lane);
        for (int t = 0; t < n; t += 16) {
            f32x2 av[8], bv[8];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) { av[s] = a2[s * 128]; bv[s] = b2[s * 128]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0], bv[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1], bv[s][1], acc[1], 0, 0, 0);
                if (s + 2 < 8) { av[s + 2] = a2[(s + 2) * 128]; bv[s + 2] = b2[(s + 2) * 128]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (MODE == 3) {
        const f32x4 *a4 = (const f32x4 *)(smem + 4 * lane), *b4 = (const f32x4 *)(smem + 4096 + 4 * lane);
        for (int t = 0; t < n; t += 16) {
            f32x4 av[4], bv[4];
            __builtin_amdgcn_sched_barrier(0);
            av[0] = a4[0]; bv[0] = b4[0];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) { av[s + 1] = a4[(s + 1) * 64]; bv[s + 1] = b4[(s + 1) * 64]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][q], bv[s][q], acc[q & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (MODE == 4) {
        for (int t = 0; t < n; t += 16) {
            float av[8], bv[16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[s] = bp[s * 128];
            av[0] = ap[0]; av[1] = ap[128];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s >> 1], bv[s], acc[s & 1], 0, 0, 0);
                if (s + 4 < 16) { bv[s + 4] = bp[(s + 4) * 128]; if ((s & 1) == 0) av[(s + 4) >> 1] = ap[((s + 4) >> 1) * 128]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    if (s == 12345.678f) out[tid] = s;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Shape { int g, n; const char *what; } shapes[] = {{256, 512, "256 workgroups x 512 MFMAs/wave (hidden GEMM shape, 1 wave per SIMD)"},
                                                           {1024, 128, "1024 workgroups x 128 MFMAs/wave (4 waves per SIMD)"},
                                                           {3648, 128, "3648 workgroups x 128 MFMAs/wave (grouped wgrad shape)"}};
    const char *names[6] = {"no operand fetch", "2 ds_read_b32 per MFMA", "2 ds_read_b64 per 2 MFMAs", "2 ds_read_b128 per 4 MFMAs", "1 ds_read_b32 per MFMA", "2 ds_read_b32 per MFMA + s_barrier per 16"};
    for (auto &sh : shapes) {
        printf("== %s\n", sh.what);
        for (int mode = 0; mode < 6; ++mode) {
            std::vector<float> t;
            for (int r = 0; r < 7; ++r) {
                auto go = [&] {
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(feed<0>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    case 1: hipLaunchKernelGGL(feed<1>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    case 2: hipLaunchKernelGGL(feed<2>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    case 3: hipLaunchKernelGGL(feed<3>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    case 4: hipLaunchKernelGGL(feed<4>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    default: hipLaunchKernelGGL(feed<5>, dim3(sh.g), dim3(256), 0, st, out, sh.n); break;
                    }
                };
                go(); go();
                CK(hipEventRecord(a, st));
                for (int i = 0; i < 20; ++i) go();
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                t.push_back(ms / 20 * 1000.f);
            }
            std::sort(t.begin(), t.end());
            const double fl = (double)sh.g * 4 * sh.n * 4096.0;
            printf("  %-44s med %7.2f us  %6.1f TF (%.0f%% of 157.3)\n", names[mode], t[3], fl / t[3] * 1e-6, fl / t[3] * 1e-6 / 157.3 * 100);
        }
    }
    return 0;
}
