// tools/l2_probe.hip -- development probe: what L2 -> CU bandwidth does an MI355X deliver for the
// access pattern of the M=256 GEMMs (few waves per CU, 16-byte loads, L2-resident data)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// every workgroup streams `region_f4` float4 starting at base + (blockIdx.x / share) * region_f4, `passes` times
template <int UNROLL>
__global__ void stream(const float4 *base, size_t region_f4, int share, int passes, float *out)
{
    const float4 *p = base + (size_t)(blockIdx.x / share) * region_f4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t n = region_f4 / (blockDim.x * UNROLL);
    for (int it = 0; it < passes; ++it)
        for (size_t i = 0; i < n; ++i) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[(i * UNROLL + u) * blockDim.x + threadIdx.x];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// WG b streams slice (b % 8) of `base` (slice_f4 float4 each): all 32 WGs of an XCD read the same slice
template <int UNROLL>
__global__ void shared_first_touch(const float4 *base, size_t slice_f4, float *out, int phase_groups = 1)
{
    const float4 *p = base + (size_t)(blockIdx.x & 7) * slice_f4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t n = slice_f4 / (blockDim.x * UNROLL);
    const size_t shift = (size_t)((blockIdx.x >> 3) % phase_groups) * (n / phase_groups);   // de-phase the sharers
    for (size_t i0 = 0; i0 < n; ++i0) {
        const size_t i = (i0 + shift) % n;
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[(i * UNROLL + u) * blockDim.x + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int UNROLL>
static void run(const char *name, const float4 *buf, size_t region_bytes, int share, int nwg, int threads, float *out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int passes = std::max<size_t>(1, (64u << 20) / region_bytes / 4);
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(stream<UNROLL>, dim3(nwg), dim3(threads), 0, 0, buf, region_bytes / 16, share, passes, out);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(stream<UNROLL>, dim3(nwg), dim3(threads), 0, 0, buf, region_bytes / 16, share, passes, out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double bytes = (double)nwg * region_bytes * passes;
    printf("%-58s %7.2f TB/s  (%5.1f GB/s per WG, %d WGs x %d thr, %d loads in flight/thr)\n", name,
           bytes / (t[2] * 1e-3) / 1e12, bytes / nwg / (t[2] * 1e-3) / 1e9, nwg, threads, UNROLL);
}

int main()
{
    float4 *buf; CK(hipMalloc(&buf, 512u << 20)); CK(hipMemset(buf, 0, 512u << 20));
    float *out; CK(hipMalloc(&out, 64));
    // private regions: WG b streams its own 64 KB (32 WGs per XCD -> 2 MB per XCD: L2-resident, > L1)
    run<6>("private 64 KB/WG, 256 WG x 256 thr", buf, 64 << 10, 1, 256, 256, out);
    run<12>("private 64 KB/WG, 256 WG x 256 thr", buf, 96 << 10, 1, 256, 256, out);
    run<6>("private 64 KB/WG, 512 WG x 256 thr (2 per CU)", buf, 64 << 10, 1, 512, 256, out);
    run<6>("private 64 KB/WG, 1024 WG x 256 thr (4 per CU)", buf, 64 << 10, 1, 1024, 256, out);
    run<8>("private 64 KB/WG, 2048 WG x 256 thr (8 per CU)", buf, 64 << 10, 1, 2048, 256, out);
    // shared regions: `share` consecutive WGs (round-robin over XCDs!) stream the same region
    run<6>("8 WGs (one per XCD) share a 64 KB region, 256 WG", buf, 64 << 10, 8, 256, 256, out);
    run<6>("64 WGs share a 512 KB region (8 per XCD), 256 WG", buf, 512 << 10, 64, 256, 256, out);
    run<6>("256 WGs share a 2 MB region (32 per XCD), 256 WG", buf, 2 << 20, 256, 256, 256, out);
    run<6>("256 WGs share a 2 MB region, 1024 WG (4 per CU)", buf, 2 << 20, 256, 1024, 256, out);
    // GEMM-like first touch: the 32 WGs of an XCD (blockIdx%8 == xcd) stream the SAME 2 MB slice once, in
    // lockstep, slice taken from a rotating 512 MB pool so that it is neither in L2 nor (mostly) in MALL
    auto ft = [&](auto U, int groups, const char *nm) {
        constexpr int UN = decltype(U)::value;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, 0));
            const int launches = 24;
            for (int l = 0; l < launches; ++l)
                hipLaunchKernelGGL(shared_first_touch<UN>, dim3(256), dim3(256), 0, 0, buf + (size_t)l * (16u << 20) / 16, (size_t)(2u << 20) / 16, out, groups);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 1) printf("first-touch, 32 WGs/XCD share a 2 MB slice, %2d loads in flight/thr, %2d phase groups (%s): %6.2f us/launch, %.2f TB/s unique\n",
                                 UN, groups, nm, ms / launches * 1e3, (16u << 20) / (ms / launches * 1e-3) / 1e12);
        }
    };
    ft(std::integral_constant<int, 6>{}, 1, "lockstep"); ft(std::integral_constant<int, 12>{}, 1, "lockstep"); ft(std::integral_constant<int, 24>{}, 1, "lockstep");
    ft(std::integral_constant<int, 6>{}, 2, "2 groups"); ft(std::integral_constant<int, 6>{}, 4, "4 groups"); ft(std::integral_constant<int, 6>{}, 8, "8 groups");
    ft(std::integral_constant<int, 6>{}, 32, "all different"); ft(std::integral_constant<int, 12>{}, 4, "4 groups");
    // HBM streaming for reference (512 MB, no reuse)
    run<8>("HBM stream 512 MB, 2048 WG", buf, 256 << 10, 1, 2048, 256, out);
    return 0;
}
