// tools/gemm_trace.hip -- development probe: per-workgroup phase timestamps (100 MHz wall clock) of
// back-to-back bp_gemm launches: entry, end of prologue, end of k-loop, end of epilogue.
// usage: gemm_trace fwd|wgrad [K]
#define BP_TRACE 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
static float *dalloc(size_t n) { float *d; CK(hipMalloc(&d, n * 4 + 65536)); CK(hipMemset(d, 0, n * 4 + 65536)); return d; }
int main(int argc, char **argv)
{
    const bool wg64 = argc > 1 && !strcmp(argv[1], "wg64");     // grouped-launch-sized wgrad: 7168x2048, 64x64x32 tiles
    const bool wgrad = wg64 || (argc > 1 && !strcmp(argv[1], "wgrad"));
    const int MW = wg64 ? 7168 : 2048;
    const int B = 256, H = 2048, K = argc > 2 ? atoi(argv[2]) : (wgrad ? 256 : 2048);
    float *Y = dalloc((size_t)2048 * (wg64 ? MW : H)), *W = dalloc((size_t)MW * H), *D = dalloc((size_t)MW * H), *Yo = dalloc((size_t)2048 * H), *bias = dalloc(H), *bd = dalloc(H);
    const int NWG = wg64 ? (MW / 64) * 32 : wgrad ? 512 : 256;
    unsigned long long *tr; CK(hipMalloc(&tr, (size_t)NWG * 8 * 8 * 4)); CK(hipMemset(tr, 0, (size_t)NWG * 8 * 8 * 4));
    GemmArgs g; EpiArgs e; memset(&g, 0, sizeof(g)); memset(&e, 0, sizeof(e));
    e.alpha = 1.f;
    if (wgrad) {
        g.A = Y; g.lda = MW; g.B = Yo; g.ldb = H; g.K = K; g.tiles_m = wg64 ? MW / 64 : 16; g.tiles_n = 32;
        e.C = W; e.ldc = H; e.m_limit = MW; e.n_limit = H; e.n_true = H; e.aux2 = D; e.ldaux2 = H; e.mom = 0.5f; e.ndiv = 256.f; e.bias_w = bias; e.bias_d = bd;
    } else {
        g.A = Y; g.lda = H; g.B = W; g.ldb = H; g.K = K; g.tiles_m = 8; g.tiles_n = 32;
        e.C = Yo; e.ldc = H; e.m_limit = B; e.n_limit = H; e.n_true = H; e.bias = bias; e.drop_thresh = 858993459u;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        for (int l = 0; l < 4; ++l) {
            g.trace = tr + (size_t)l * NWG * 8;
            if (wg64) hipLaunchKernelGGL((bp_gemm<64, 64, 32, 2, 2, false, false, EPI_WGRAD_UPDATE, 1, 8>), dim3(NWG), dim3(256), 0, st, g, e);
            else if (wgrad && K == 256 && argc > 3) hipLaunchKernelGGL((bp_gemm<128, 64, 16, 2, 2, false, false, EPI_WGRAD_UPDATE, 1, 16>), dim3(NWG), dim3(256), 0, st, g, e);
            else if (wgrad) hipLaunchKernelGGL((bp_gemm<128, 64, 16, 2, 2, false, false, EPI_WGRAD_UPDATE, 1>), dim3(NWG), dim3(256), 0, st, g, e);
            else hipLaunchKernelGGL((bp_gemm<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1>), dim3(NWG), dim3(256), 0, st, g, e);
        }
        CK(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> h((size_t)NWG * 8 * 4);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int l = 0; l < 4; ++l) for (int b = 0; b < NWG; ++b) t0 = std::min(t0, h[((size_t)l * NWG + b) * 8]);
    for (int l = 1; l < 4; ++l) {
        double mn[4] = {1e30, 1e30, 1e30, 1e30}, mx[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};
        double dur[3] = {0, 0, 0};
        for (int b = 0; b < NWG; ++b) {
            double t[4];
            for (int i = 0; i < 4; ++i) {
                t[i] = (h[((size_t)l * NWG + b) * 8 + i] - t0) * 0.01;   // us
                mn[i] = std::min(mn[i], t[i]); mx[i] = std::max(mx[i], t[i]); av[i] += t[i] / NWG;
            }
            for (int i = 0; i < 3; ++i) dur[i] += (t[i + 1] - t[i]) / NWG;
        }
        if (wg64) {   // how many workgroups are alive / in the k-loop over time (1 us bins)
            double lo = 1e30, hi = 0;
            for (int b = 0; b < NWG; ++b) { lo = std::min(lo, (h[((size_t)l * NWG + b) * 8] - t0) * 0.01); hi = std::max(hi, (h[((size_t)l * NWG + b) * 8 + 3] - t0) * 0.01); }
            const int nb = (int)(hi - lo) / 4 + 1;
            std::vector<double> alive(nb, 0), inloop(nb, 0);
            for (int b = 0; b < NWG; ++b) {
                double t[4]; for (int i = 0; i < 4; ++i) t[i] = (h[((size_t)l * NWG + b) * 8 + i] - t0) * 0.01 - lo;
                for (int q = 0; q < nb; ++q) {
                    const double a = q * 4.0, z = a + 4.0;
                    alive[q] += std::max(0.0, std::min(z, t[3]) - std::max(a, t[0])) / 4.0;
                    inloop[q] += std::max(0.0, std::min(z, t[2]) - std::max(a, t[1])) / 4.0;
                }
            }
            printf("  span %.1f us; per 4-us bin: WGs alive / in k-loop (per CU):", hi - lo);
            for (int q = 0; q < nb; ++q) printf(" %.2f/%.2f", alive[q] / 256, inloop[q] / 256);
            printf("\n");
        }
        if (wg64) {
            double d[5] = {0, 0, 0, 0, 0};
            for (int b = 0; b < NWG; ++b) {
                const unsigned long long *q = &h[((size_t)l * NWG + b) * 8];
                d[0] += (q[4] - q[1]) * 0.01 / NWG; d[1] += (q[5] - q[4]) * 0.01 / NWG; d[2] += (q[6] - q[5]) * 0.01 / NWG;
                d[3] += (q[7] - q[6]) * 0.01 / NWG; d[4] += (q[2] - q[7]) * 0.01 / 4 / NWG;
            }
            printf("  k-tile durations: t0 %.2f t1 %.2f t2 %.2f t3 %.2f, t4..7 avg %.2f us\n", d[0], d[1], d[2], d[3], d[4]);
        }
        if (!wgrad) {     // drift inside the groups of 8 m-tile workgroups that stream the same weight panel through one XCD's L2
            double sp_entry = 0, sp_loop = 0, sp_max = 0;
            for (int xcd = 0; xcd < 8; ++xcd)
                for (int tn = 0; tn < 4; ++tn) {
                    double lo0 = 1e30, hi0 = 0, lo2 = 1e30, hi2 = 0;
                    for (int tm = 0; tm < 8; ++tm) {
                        const int b = xcd + 8 * (tn * 8 + tm);
                        const double t0b = (h[((size_t)l * NWG + b) * 8 + 0] - t0) * 0.01, t2b = (h[((size_t)l * NWG + b) * 8 + 2] - t0) * 0.01;
                        lo0 = std::min(lo0, t0b); hi0 = std::max(hi0, t0b); lo2 = std::min(lo2, t2b); hi2 = std::max(hi2, t2b);
                    }
                    sp_entry += (hi0 - lo0) / 32; sp_loop += (hi2 - lo2) / 32; sp_max = std::max(sp_max, hi2 - lo2);
                }
            printf("  8 workgroups of one weight panel: spread of entry %.2f us, of k-loop end %.2f us on average (max %.2f); one k-tile = %.2f us\n",
                   sp_entry, sp_loop, sp_max, dur[1] / ((K + 63) / 64));
        }
        if (!wg64) {
            double mhz = 0; int cnt = 0;
            for (int b = 0; b < NWG; ++b) {
                const unsigned long long *q = &h[((size_t)l * NWG + b) * 8];
                if (q[3] > q[0] && q[7] > q[6]) { mhz += (double)(q[7] - q[6]) / (double)(q[3] - q[0]) * 100.0; ++cnt; }
            }
            printf("  shader clock during the kernel (clock64 / wall_clock64): %.0f MHz\n", cnt ? mhz / cnt : 0.0);
        }
        printf("launch %d (%s K=%d): entry [%.2f..%.2f] prologue_done [%.2f..%.2f] loop_done [%.2f..%.2f] end [%.2f..%.2f] us | per-WG avg: prologue %.2f loop %.2f epilogue %.2f\n",
               l, wgrad ? "wgrad 128x64x16" : "fwd 32x64x64", K, mn[0], mx[0], mn[1], mx[1], mn[2], mx[2], mn[3], mx[3], dur[0], dur[1], dur[2]);
    }
    return 0;
}
