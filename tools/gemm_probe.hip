// tools/gemm_probe.hip -- development probe (not part of the product): times template variants
// of bp_gemm on the C2 shapes in one process so a single GPU call compares them (interleaved
// rounds, median).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/gemm_probe tools/gemm_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <string>
#include <vector>
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#include "wgrad_glds_probe.h"
#include "fwd_glds_probe.h"

template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    float s = 0.f;
    for (int c = 0; c < NACC; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct Variant { std::string name; std::function<void(hipStream_t)> run; double flops; };

static float *dalloc(size_t n, float scale, unsigned seed)
{
    std::vector<float> h(n);
    srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    float *d; CK(hipMalloc(&d, n * 4 + 65536)); CK(hipMemset(d, 0, n * 4 + 65536)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI, int PF, int NT_S = 0>
static void go(hipStream_t st, GemmArgs g, const EpiArgs &e, int M, int N, int dyn)
{
    g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
    int grid = g.tiles_m * g.tiles_n;
    if (dyn > 0 && grid > dyn) grid = dyn;      // `dyn` = persistent grid cap
    hipLaunchKernelGGL((bp_gemm<BM, BN, BK, WM, WN, A_KC, B_KC, EPI, PF, NT_S>), dim3(grid), dim3(256), 0, st, g, e);
}

int main(int argc, char **argv)
{
    const int B = 256, H = 2048;
    const int KW = argc > 3 ? atoi(argv[3]) : 256;    // reduction length (frames) of the wgrad variants
    int LD = 2048;
    const char *filter = argc > 1 ? argv[1] : "";
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t LDMAX = 2304;
    float *Y = dalloc((size_t)2048 * LDMAX, 1.f, 1), *W = dalloc((size_t)H * LDMAX, 0.03f, 2), *D = dalloc((size_t)H * LDMAX, 0.f, 3);
    float *Yo = dalloc((size_t)B * LDMAX, 0.f, 4), *dX = dalloc((size_t)2048 * LDMAX, 0.01f, 5), *bias = dalloc(LDMAX, 0.1f, 6);
    float *bd = dalloc(H, 0.f, 7);
    std::vector<Variant> vs;
    auto fwd_args = [&](GemmArgs &g, EpiArgs &e) {
        memset(&g, 0, sizeof(g)); memset(&e, 0, sizeof(e));
        g.A = Y; g.lda = LD; g.B = W; g.ldb = LD; g.K = H;
        e.C = Yo; e.ldc = LD; e.m_limit = B; e.n_limit = H; e.n_true = H; e.bias = bias; e.alpha = 1.f; e.drop_thresh = 858993459u;
        e.seed_lo = 1; e.step = 3; e.layer = 2;
    };
    auto dg_args = [&](GemmArgs &g, EpiArgs &e) {
        memset(&g, 0, sizeof(g)); memset(&e, 0, sizeof(e));
        g.A = dX; g.lda = LD; g.B = W; g.ldb = LD; g.K = H;
        e.C = Yo; e.ldc = LD; e.m_limit = B; e.n_limit = H; e.n_true = H; e.aux = Y; e.ldaux = LD; e.alpha = 1.f;
    };
    auto wg_args = [&](GemmArgs &g, EpiArgs &e) {
        memset(&g, 0, sizeof(g)); memset(&e, 0, sizeof(e));
        g.A = Y; g.lda = LD; g.B = dX; g.ldb = LD; g.K = KW;
        e.C = W; e.ldc = LD; e.m_limit = H; e.n_limit = H; e.n_true = H; e.aux2 = D; e.ldaux2 = LD; e.alpha = 1.f;
        e.mom = 0.5f; e.c1 = 0.0f; e.wc = 0.f; e.ndiv = 256.f; e.bias_w = bias; e.bias_d = bd;
    };
    const double fl = 2.0 * B * H * H;
#define FWD(BM, BN, BK, WM, WN, PF) vs.push_back({"fwd  " #BM "x" #BN "x" #BK " w" #WM "x" #WN " pf" #PF, [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); go<BM, BN, BK, WM, WN, true, false, EPI_FWD_HIDDEN, PF>(s, g, e, B, H, 0); }, fl})
#define DGR(BM, BN, BK, WM, WN, PF) vs.push_back({"dgrad " #BM "x" #BN "x" #BK " w" #WM "x" #WN " pf" #PF, [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); go<BM, BN, BK, WM, WN, true, true, EPI_DGRAD, PF>(s, g, e, B, H, 0); }, fl})
#define WGR(BM, BN, BK, WM, WN, PF, DYN) vs.push_back({"wgrad " #BM "x" #BN "x" #BK " w" #WM "x" #WN " pf" #PF " grid" #DYN, [&](hipStream_t s) { GemmArgs g; EpiArgs e; wg_args(g, e); go<BM, BN, BK, WM, WN, false, false, EPI_WGRAD_UPDATE, PF>(s, g, e, H, H, DYN); }, 2.0 * H * H * (double)KW})
    // split-K over 2 (4) workgroup rows: 512 (1024) workgroups = 2 (4) per CU running out of phase; partial sums
    // stored plainly (EPI_PARTIAL).  Compare with "fwdP1" = the same plain-store kernel without the split.
#define FWDSK(NS) vs.push_back({"fwdSK" #NS " 32x64x64 w1x2 partial", [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); g.K = H / NS; g.k_split = H / NS; g.slab_stride = (size_t)B * LDMAX; e.C = dX; \
        g.tiles_m = B / 32; g.tiles_n = H / 64; hipLaunchKernelGGL((bp_gemm<32, 64, 64, 1, 2, true, false, EPI_PARTIAL, 1>), dim3(g.tiles_m * g.tiles_n, NS), dim3(256), 0, s, g, e); }, fl})
    FWDSK(1); FWDSK(2); FWDSK(4);
    // round 4: LARGER tiles with the split (fewer operand bytes per flop into each CU: 24 B/clk at full MFMA rate for 32x64, 16 for
    // 64x64, 12 for 64x128 / 128x64), still 256 workgroups; plain partial stores, i.e. WITHOUT the cross-workgroup reduction a real
    // kernel would add -- an upper bound of what the shape can give
#define FWDSKT(BM, BN, BK, WM, WN, NS) vs.push_back({"fwdSK" #NS " " #BM "x" #BN "x" #BK " w" #WM "x" #WN " partial", [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); g.K = H / NS; g.k_split = H / NS; g.slab_stride = (size_t)B * LDMAX; e.C = dX; \
        g.tiles_m = B / BM; g.tiles_n = H / BN; hipLaunchKernelGGL((bp_gemm<BM, BN, BK, WM, WN, true, false, EPI_PARTIAL, 1>), dim3(g.tiles_m * g.tiles_n, NS), dim3(256), 0, s, g, e); }, fl})
    FWDSKT(64, 64, 64, 2, 2, 2); FWDSKT(64, 64, 32, 2, 2, 2); FWDSKT(64, 128, 32, 2, 2, 4); FWDSKT(128, 64, 32, 2, 2, 4); FWDSKT(64, 128, 64, 2, 2, 4); FWDSKT(64, 64, 64, 2, 2, 4); FWDSKT(128, 128, 32, 2, 2, 8);
    // same kernel, every k-row of W aliased to row 0 (ldb = 0): operands always hit L2 -> isolates HBM/MALL latency
    vs.push_back({"fwdL2 32x64x64 w1x2 pf1 (ldb=0)", [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); g.ldb = 0; go<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1>(s, g, e, B, H, 0); }, fl});
    FWD(32, 64, 64, 1, 2, 1); FWD(32, 64, 128, 1, 2, 1); FWD(64, 32, 128, 2, 1, 1); FWD(32, 64, 64, 1, 2, 2); FWD(32, 64, 32, 1, 2, 1); FWD(32, 64, 32, 1, 2, 2);
    FWD(64, 32, 64, 2, 1, 1); FWD(64, 32, 64, 2, 1, 2);
    FWD(32, 32, 64, 1, 1, 1); FWD(32, 32, 64, 1, 1, 2); FWD(32, 32, 128, 1, 1, 1);
    FWD(64, 64, 32, 2, 2, 1);
    DGR(32, 64, 64, 1, 2, 1); DGR(32, 64, 128, 1, 2, 1); DGR(32, 64, 64, 1, 2, 2); DGR(64, 32, 64, 2, 1, 1); DGR(32, 32, 64, 1, 1, 1); DGR(32, 32, 64, 1, 1, 2);
    WGR(64, 64, 32, 2, 2, 1, 0); WGR(64, 64, 32, 2, 2, 1, 512); WGR(64, 64, 32, 2, 2, 1, 256); WGR(64, 64, 32, 2, 2, 1, 768);
    WGR(128, 64, 16, 2, 2, 1, 0); WGR(128, 64, 16, 2, 2, 1, 256); WGR(128, 64, 32, 2, 2, 1, 256);
    WGR(64, 128, 16, 2, 2, 1, 0); WGR(64, 128, 16, 2, 2, 1, 256);
    WGR(128, 128, 16, 2, 2, 1, 0); WGR(128, 128, 16, 2, 2, 1, 128);
#define WGS(BM, BN, BK, WM, WN, NT) vs.push_back({"wgrad " #BM "x" #BN "x" #BK " w" #WM "x" #WN " static" #NT " grid0", [&](hipStream_t s) { GemmArgs g; EpiArgs e; wg_args(g, e); go<BM, BN, BK, WM, WN, false, false, EPI_WGRAD_UPDATE, 1, NT>(s, g, e, H, H, 0); }, 2.0 * H * H * (double)KW})
    WGS(64, 64, 64, 2, 2, 4); WGS(128, 64, 16, 2, 2, 16); WGS(64, 64, 32, 2, 2, 8); WGS(64, 64, 16, 2, 2, 16); WGS(64, 128, 16, 2, 2, 16); WGS(128, 64, 32, 2, 2, 8); WGS(128, 128, 16, 2, 2, 16);
    // LDS-DMA staged wgrad+update (bp_wgrad_glds.h), one 2048x2048 problem
    vs.push_back({"wgrad glds 64x64x32 3-stage", [&](hipStream_t s) { MultiArgs a; memset(&a, 0, sizeof(a)); wg_args(a.g[0], a.e[0]);
        a.g[0].tiles_m = H / 64; a.g[0].tiles_n = H / 64; a.first_tile[0] = 0; a.first_tile[1] = a.g[0].tiles_m * a.g[0].tiles_n; a.n = 1;
        hipLaunchKernelGGL(bp_wgrad_glds_multi<1>, dim3(a.first_tile[1]), dim3(256), 0, s, a); }, 2.0 * H * H * (double)KW});
    vs.push_back({"wgrad glds 128x64x16 3-stage", [&](hipStream_t s) { MultiArgs a; memset(&a, 0, sizeof(a)); wg_args(a.g[0], a.e[0]);
        a.g[0].tiles_m = H / 128; a.g[0].tiles_n = H / 64; a.first_tile[0] = 0; a.first_tile[1] = a.g[0].tiles_m * a.g[0].tiles_n; a.n = 1;
        hipLaunchKernelGGL(bp_wgrad_glds_multi<2>, dim3(a.first_tile[1]), dim3(256), 0, s, a); }, 2.0 * H * H * (double)KW});
    // LDS-DMA staged M = 256 GEMMs (fwd_glds_probe.h)
    vs.push_back({"fwd  glds 32x64x64 4-stage", [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma<false, EPI_FWD_HIDDEN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"dgrad glds 32x64x64 4-stage", [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma<true, EPI_DGRAD>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"fwd  glds2 32x64x64 (asm reads, mid-tile barrier)", [&](hipStream_t s) { GemmArgs g; EpiArgs e; fwd_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma2<false, EPI_FWD_HIDDEN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"dgrad glds2 32x64x64 (asm reads, mid-tile barrier)", [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma2<true, EPI_DGRAD>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"dgrad glds2 ablation: no DMA in the loop", [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma2<true, EPI_DGRAD, 1>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"dgrad glds2 ablation: no barrier", [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma2<true, EPI_DGRAD, 2>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    vs.push_back({"dgrad glds2 ablation: no DMA, no barrier", [&](hipStream_t s) { GemmArgs g; EpiArgs e; dg_args(g, e); g.tiles_m = B / 32; g.tiles_n = H / 64;
        hipLaunchKernelGGL((bp_gemm_dma2<true, EPI_DGRAD, 3>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g, e); }, fl});
    // short-kernel MFMA shapes: G workgroups x 4 waves x N MFMAs per wave, NACC chains (what a 64x64x256 wgrad tile issues: 128 per wave)
#define SHAPE(G, N, NACC) vs.push_back({"mfma shape G" #G " n" #N " chains" #NACC, [&](hipStream_t s) { hipLaunchKernelGGL(mfma_peak<NACC>, dim3(G), dim3(256), 0, s, Yo, N / NACC); }, (double)G * 4 * N * 4096.0})
    SHAPE(1024, 128, 2); SHAPE(768, 128, 2); SHAPE(256, 128, 2); SHAPE(1024, 128, 4); SHAPE(1024, 128, 1); SHAPE(256, 512, 2); SHAPE(3648, 128, 2); SHAPE(2048, 128, 2);
    // calibration: what the matrix pipe delivers on this box (one wave per SIMD, 256 workgroups)
    vs.push_back({"mfma peak: 1 dependent chain/wave", [&](hipStream_t s) { hipLaunchKernelGGL(mfma_peak<1>, dim3(256), dim3(256), 0, s, Yo, 512); }, 256.0 * 4 * 512 * 4096.0});
    vs.push_back({"mfma peak: 4 chains/wave", [&](hipStream_t s) { hipLaunchKernelGGL(mfma_peak<4>, dim3(256), dim3(256), 0, s, Yo, 512); }, 256.0 * 4 * 512 * 4096.0 * 4});

    if (getenv("PROBE_CHECK")) {
        const size_t nW = (size_t)H * LD;
        std::vector<float> w0(nW), r1(nW), r2(nW), d1(nW), d2(nW), b1(H), b2(H);
        CK(hipMemcpy(w0.data(), W, nW * 4, hipMemcpyDeviceToHost));
        for (int which = 0; which < 2; ++which) {
            CK(hipMemcpy(W, w0.data(), nW * 4, hipMemcpyHostToDevice)); CK(hipMemset(D, 0, nW * 4)); CK(hipMemset(bias, 0, H * 4)); CK(hipMemset(bd, 0, H * 4));
            GemmArgs g; EpiArgs e; wg_args(g, e); e.c1 = 0.5f; e.wc = 0.01f;
            if (which == 0) go<64, 64, 32, 2, 2, false, false, EPI_WGRAD_UPDATE, 1, 8>(st, g, e, H, H, 0);
            else { MultiArgs a; memset(&a, 0, sizeof(a)); a.g[0] = g; a.e[0] = e; a.g[0].tiles_m = H / 128; a.g[0].tiles_n = H / 64; a.first_tile[1] = (H / 128) * (H / 64); a.n = 1;
                   hipLaunchKernelGGL(bp_wgrad_glds_multi<2>, dim3(a.first_tile[1]), dim3(256), 0, st, a); }
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(which ? r2.data() : r1.data(), W, nW * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(which ? d2.data() : d1.data(), D, nW * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(which ? b2.data() : b1.data(), bias, H * 4, hipMemcpyDeviceToHost));
        }
        double mw = 0, md = 0, mb = 0, sw = 0, sd = 0, sb = 0;
        for (size_t i = 0; i < nW; ++i) { mw = std::max(mw, (double)fabsf(r1[i] - r2[i])); md = std::max(md, (double)fabsf(d1[i] - d2[i])); sw = std::max(sw, (double)fabsf(r1[i] - w0[i])); sd = std::max(sd, (double)fabsf(d1[i])); }
        for (int i = 0; i < H; ++i) { mb = std::max(mb, (double)fabsf(b1[i] - b2[i])); sb = std::max(sb, (double)fabsf(b1[i])); }
        printf("CHECK glds vs static8: max|dW| %.3e (update size %.3e)  max|dDelta| %.3e (|delta| %.3e)  max|dbias| %.3e (|bias| %.3e)\n", mw, sw, md, sd, mb, sb);
        CK(hipMemcpy(W, w0.data(), nW * 4, hipMemcpyHostToDevice)); CK(hipMemset(D, 0, nW * 4));
    }
    if (getenv("PROBE_CHECK2")) {
        const size_t nY = (size_t)B * LD;
        std::vector<float> r1(nY), r2(nY);
        for (int kind = 0; kind < 2; ++kind) {
            for (int which = 0; which < 2; ++which) {
                CK(hipMemset(Yo, 0, nY * 4));
                GemmArgs g; EpiArgs e;
                if (kind == 0) { fwd_args(g, e); e.drop_thresh = 0; } else dg_args(g, e);
                if (which == 0) { if (kind == 0) go<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1>(st, g, e, B, H, 0); else go<32, 64, 64, 1, 2, true, true, EPI_DGRAD, 1>(st, g, e, B, H, 0); }
                else { g.tiles_m = B / 32; g.tiles_n = H / 64;
                       if (kind == 0) hipLaunchKernelGGL((bp_gemm_dma<false, EPI_FWD_HIDDEN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, st, g, e);
                       else hipLaunchKernelGGL((bp_gemm_dma<true, EPI_DGRAD>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, st, g, e); }
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                CK(hipMemcpy(which ? r2.data() : r1.data(), Yo, nY * 4, hipMemcpyDeviceToHost));
            }
            double md = 0, mx = 0; size_t nz = 0;
            for (size_t i = 0; i < nY; ++i) { md = std::max(md, (double)fabsf(r1[i] - r2[i])); mx = std::max(mx, (double)fabsf(r1[i])); nz += r2[i] != 0.f; }
            printf("CHECK2 %s glds vs product: max|diff| %.3e, max|ref| %.3e, nonzero outputs %zu of %zu\n", kind ? "dgrad" : "fwd", md, mx, nz, nY);
        }
    }
    if (getenv("PROBE_CHECK3")) {
        const size_t nY = (size_t)B * LD;
        std::vector<float> r1(nY), r2(nY);
        for (int kind = 0; kind < 2; ++kind) {
            for (int which = 0; which < 2; ++which) {
                CK(hipMemset(Yo, 0, nY * 4));
                GemmArgs g; EpiArgs e;
                if (kind == 0) { fwd_args(g, e); e.drop_thresh = 0; } else dg_args(g, e);
                if (which == 0) { if (kind == 0) go<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1>(st, g, e, B, H, 0); else go<32, 64, 64, 1, 2, true, true, EPI_DGRAD, 1>(st, g, e, B, H, 0); }
                else { g.tiles_m = B / 32; g.tiles_n = H / 64;
                       if (kind == 0) hipLaunchKernelGGL((bp_gemm_dma2<false, EPI_FWD_HIDDEN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, st, g, e);
                       else hipLaunchKernelGGL((bp_gemm_dma2<true, EPI_DGRAD>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, st, g, e); }
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                CK(hipMemcpy(which ? r2.data() : r1.data(), Yo, nY * 4, hipMemcpyDeviceToHost));
            }
            double md = 0, mx = 0; size_t nz = 0;
            for (size_t i = 0; i < nY; ++i) { md = std::max(md, (double)fabsf(r1[i] - r2[i])); mx = std::max(mx, (double)fabsf(r1[i])); nz += r2[i] != 0.f; }
            printf("CHECK3 %s glds2 vs product: max|diff| %.3e, max|ref| %.3e, nonzero outputs %zu of %zu\n", kind ? "dgrad" : "fwd", md, mx, nz, nY);
        }
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int lds[] = {2048, 2112, 2176, 2304};
    const int nld = getenv("PROBE_LDS") ? 4 : 1;
    for (int li = 0; li < nld; ++li) {
    LD = lds[li];
    printf("==== leading dimension %d floats\n", LD);
    const int rounds = 5, iters = 20;
    std::vector<std::vector<float>> t(vs.size());
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            if (filter[0]) {      // comma-separated substrings, any match
                bool hit = false; std::string f(filter); size_t p0 = 0;
                while (p0 <= f.size()) { size_t p1 = f.find(',', p0); if (p1 == std::string::npos) p1 = f.size();
                    if (p1 > p0 && vs[v].name.find(f.substr(p0, p1 - p0)) != std::string::npos) hit = true; p0 = p1 + 1; }
                if (!hit) continue;
            }
            vs[v].run(st); vs[v].run(st);
            CK(hipEventRecord(a, st));
            for (int i = 0; i < iters; ++i) vs[v].run(st);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            t[v].push_back(ms / iters * 1000.f);
        }
    for (size_t v = 0; v < vs.size(); ++v) {
        if (t[v].empty()) continue;
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%-34s med %7.2f us  min %7.2f us  %6.1f TF (%.0f%% of 157.3)\n", vs[v].name.c_str(), med, t[v][0],
               vs[v].flops / med * 1e-6, vs[v].flops / med * 1e-6 / 157.3 * 100);
    }
    }
    return 0;
}
