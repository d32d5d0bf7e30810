// tools/wgrad_trace.hip -- development probe: per-workgroup entry / exit timestamps of the SHIPPED grouped weight-gradient +
// update kernel body (WgradDma<16,4,4,256>::run, bp_wgrad_dma.h) on the C2 shapes (2880x2048, 2048x2048 x2, 2048x320; 256
// frames), launched back to back: span of a launch, workgroup duration, workgroups alive per CU over time (ramp and tail),
// shader clock delivered during the kernel.     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_trace.hip -o tools/wgrad_trace.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#include "../dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
static float *dalloc(size_t n) { float *d; CK(hipMalloc(&d, n * 4 + 65536)); CK(hipMemset(d, 0, n * 4 + 65536)); return d; }

__global__ __launch_bounds__(256, 4) void traced(const MultiArgs a, unsigned long long *tr)
{
    using K = WgradDma<16, 4, 4, 256, false>;
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    K::run(a.g[p], a.e[p], b - a.first_tile[p], a.first_tile[p + 1] - a.first_tile[p], smem);
    if (threadIdx.x == 0) { tr[(size_t)b * 4] = w0; tr[(size_t)b * 4 + 1] = wall_clock64(); tr[(size_t)b * 4 + 2] = c0; tr[(size_t)b * 4 + 3] = clock64(); }
}

int main()
{
    const int B = 256, prev[4] = {2880, 2048, 2048, 2048}, cur[4] = {2048, 2048, 2048, 320};
    MultiArgs a; memset(&a, 0, sizeof(a));
    int t = 0;
    for (int i = 0; i < 4; ++i) {
        float *Y = dalloc((size_t)B * prev[i]), *dX = dalloc((size_t)B * cur[i]), *W = dalloc((size_t)prev[i] * cur[i]), *D = dalloc((size_t)prev[i] * cur[i]);
        float *bw = dalloc(cur[i]), *bd = dalloc(cur[i]);
        GemmArgs &g = a.g[i]; EpiArgs &e = a.e[i];
        g.A = Y; g.lda = prev[i]; g.B = dX; g.ldb = cur[i]; g.K = B; g.tiles_m = prev[i] / 64; g.tiles_n = cur[i] / 64;
        e.alpha = 1.f; e.C = W; e.ldc = cur[i]; e.m_limit = prev[i]; e.n_limit = cur[i]; e.n_true = cur[i]; e.aux2 = D; e.ldaux2 = cur[i];
        e.mom = 0.5f; e.c1 = 0.005f; e.ndiv = 256.f; e.bias_w = bw; e.bias_d = bd;
        a.first_tile[i] = t;
        t += (g.tiles_m * g.tiles_n + 7) & ~7;
    }
    a.first_tile[4] = t; a.n = 4;
    const int NWG = t, NL = 6;
    unsigned long long *tr; CK(hipMalloc(&tr, (size_t)NL * NWG * 4 * 8)); CK(hipMemset(tr, 0, (size_t)NL * NWG * 4 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(traced, dim3(NWG), dim3(256), 0, st, a, tr + (size_t)l * NWG * 4);
        CK(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> h((size_t)NL * NWG * 4);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    printf("grouped wgrad+update body, %d workgroups (4 problems), 256 frames; times in us\n", NWG);
    for (int l = 2; l < NL; ++l) {
        const unsigned long long *q = &h[(size_t)l * NWG * 4];
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < NWG; ++b) { lo = std::min(lo, q[b * 4]); hi = std::max(hi, q[b * 4 + 1]); }
        double dur = 0, dmin = 1e30, dmax = 0, mhz = 0;
        for (int b = 0; b < NWG; ++b) {
            const double d = (q[b * 4 + 1] - q[b * 4]) * 0.01;
            dur += d / NWG; dmin = std::min(dmin, d); dmax = std::max(dmax, d);
            mhz += (double)(q[b * 4 + 3] - q[b * 4 + 2]) / (double)(q[b * 4 + 1] - q[b * 4]) * 100.0 / NWG;
        }
        const double span = (hi - lo) * 0.01;
        const int nb = (int)(span / 4) + 1;
        std::vector<double> alive(nb, 0.0);
        for (int b = 0; b < NWG; ++b) {
            const double s = (q[b * 4] - lo) * 0.01, e = (q[b * 4 + 1] - lo) * 0.01;
            for (int k = 0; k < nb; ++k) alive[k] += std::max(0.0, std::min(k * 4.0 + 4.0, e) - std::max(k * 4.0, s)) / 4.0;
        }
        double full = 0;      // time-integral of residency relative to 4 workgroups per CU
        for (int k = 0; k < nb; ++k) full += alive[k] / 256 / 4.0 * 4.0;
        printf("launch %d: span %.1f | workgroup duration avg %.2f [%.2f..%.2f] | shader clock %.0f MHz | residency-equivalent full-occupancy time %.1f us\n",
               l, span, dur, dmin, dmax, mhz, full);
        printf("  workgroups alive per CU, 4-us bins:");
        for (int k = 0; k < nb; ++k) printf(" %.2f", alive[k] / 256);
        printf("\n");
    }
    return 0;
}
