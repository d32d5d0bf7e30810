// tools/hwid_probe.hip -- what HW_REG_HW_ID / HW_REG_XCC_ID say for the workgroups of one launch on gfx950, and in which order the
// dispatcher hands workgroups to CUs (round 6: placement of the early layer-1 forward inside the weight-gradient launch).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/hwid_probe tools/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256, 4) void probe(unsigned *out, int spin)
{
    __shared__ float pad[8192];                      // 32 KB like the weight-gradient kernel: 4 workgroups per CU
    pad[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);
        out[3 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
        out[3 * blockIdx.x + 2] = (unsigned)wall_clock64();
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
}
int main()
{
    const int N = 2048;
    unsigned *d; hipMalloc(&d, 3 * N * 4);
    std::vector<unsigned> h(3 * N);
    hipLaunchKernelGGL(probe, dim3(N), dim3(256), 0, 0, d, 2000);      // 20 us per workgroup
    hipMemcpy(h.data(), d, 3 * N * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; std::map<unsigned, int> first256;
    unsigned tmin = ~0u;
    for (int b = 0; b < N; ++b) tmin = h[3 * b + 2] < tmin ? h[3 * b + 2] : tmin;
    for (int b = 0; b < N; ++b) {
        const unsigned xcc = h[3 * b] & 7, hw = h[3 * b + 1];
        const unsigned cu = ((xcc * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u);
        cus.insert(cu);
        if (b < 256) first256[cu]++;
        if (b < 24 || (b >= 1024 && b < 1032)) printf("block %4d: xcc_id 0x%x hw_id 0x%08x -> xcc %u se %u sh %u cu %u simd %u | start +%u ticks\n", b, h[3 * b], hw, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, h[3 * b + 2] - tmin);
    }
    int mx = 0; for (auto &kv : first256) mx = kv.second > mx ? kv.second : mx;
    printf("distinct (xcc, se, sh, cu) over %d blocks: %zu; the first 256 blocks sit on %zu distinct CUs, at most %d on one\n", N, cus.size(), first256.size(), mx);
    return 0;
}
