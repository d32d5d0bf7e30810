// tools/wgrad_sk_probe.hip -- development probe: the persistent k-balanced grouped wgrad+update (bp_wgrad_sk.h) against the shipped
// one-workgroup-per-tile kernel (bp_wgrad_dma.h) on the C2 / C3 shapes: results (W, delta, bias after several updates, random data,
// every word compared) and interleaved A/B timing of back-to-back launches.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_sk_probe.hip -o tools/wgrad_sk_probe.bin
//     tools/wgrad_sk_probe.bin [c2|c3] [grid]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#include "../dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h"
#include "wgrad_sk_probe.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

static unsigned long long rng_s = 0x9E3779B97F4A7C15ull;
static float frand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (float)((double)(rng_s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
static float *dalloc_rand(size_t n, float scale)
{
    std::vector<float> h(n + 16384, 0.f);
    for (size_t i = 0; i < n; ++i) h[i] = scale * frand();
    float *d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
}
static float *dclone(const float *src, size_t n) { float *d; CK(hipMalloc(&d, (n + 16384) * 4)); CK(hipMemcpy(d, src, (n + 16384) * 4, hipMemcpyDeviceToDevice)); return d; }

template <bool STORE>
static int run_case(const char *name, const int *prev, const int *cur, int np, int grid_arg)
{
    const int B = 256;
    MultiArgs ma; memset(&ma, 0, sizeof(ma));
    SkArgs sa; memset(&sa, 0, sizeof(sa));
    std::vector<float *> Wa(np), Da(np), Wb(np), Db(np), ba(np), bb(np);
    std::vector<size_t> nw(np);
    int t = 0, cum = 0;
    for (int i = 0; i < np; ++i) {
        nw[i] = (size_t)prev[i] * cur[i];
        float *Y = dalloc_rand((size_t)B * prev[i], 1.0f), *dX = dalloc_rand((size_t)B * cur[i], 0.01f);
        Wa[i] = dalloc_rand(nw[i], 0.05f); Da[i] = dalloc_rand(nw[i], 0.001f); ba[i] = dalloc_rand(2 * (size_t)cur[i], 0.01f);
        Wb[i] = dclone(Wa[i], nw[i]); Db[i] = dclone(Da[i], nw[i]); bb[i] = dclone(ba[i], 2 * (size_t)cur[i]);
        GemmArgs g; memset(&g, 0, sizeof(g));
        g.A = Y; g.lda = prev[i]; g.B = dX; g.ldb = cur[i]; g.K = B; g.tiles_m = prev[i] / 64; g.tiles_n = cur[i] / 64;
        EpiArgs e; memset(&e, 0, sizeof(e));
        e.alpha = 1.f; e.ldc = cur[i]; e.m_limit = prev[i]; e.n_limit = cur[i]; e.n_true = cur[i]; e.ldaux2 = cur[i];
        e.mom = 0.5f; e.c1 = 0.5f; e.wc = 0.0f; e.ndiv = 256.f;
        ma.g[i] = g; ma.e[i] = e; sa.g[i] = g; sa.e[i] = e;
        if (STORE) { ma.e[i].C = Wa[i]; ma.e[i].bias_g = ba[i]; sa.e[i].C = Wb[i]; sa.e[i].bias_g = bb[i]; }
        else {
            ma.e[i].C = Wa[i]; ma.e[i].aux2 = Da[i]; ma.e[i].bias_w = ba[i]; ma.e[i].bias_d = ba[i] + cur[i];
            sa.e[i].C = Wb[i]; sa.e[i].aux2 = Db[i]; sa.e[i].bias_w = bb[i]; sa.e[i].bias_d = bb[i] + cur[i];
        }
        ma.first_tile[i] = t; t += (g.tiles_m * g.tiles_n + 7) & ~7;
        if ((g.tiles_m * g.tiles_n) % 8) { printf("tiles of problem %d not a multiple of 8\n", i); return 1; }
        sa.cum[i] = cum; cum += g.tiles_m * g.tiles_n / 8;
    }
    ma.first_tile[np] = t; ma.n = np;
    sa.cum[np] = cum; sa.n = np;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int grid = grid_arg > 0 ? grid_arg : pr.multiProcessorCount * 4;
    using SK = WgradSk<256, STORE>;
    CK(hipMalloc(&sa.ws, (size_t)grid * SK::WS * 4)); CK(hipMemset(sa.ws, 0xff, (size_t)grid * SK::WS * 4));
    CK(hipMalloc(&sa.flags, (size_t)grid * 4)); CK(hipMemset(sa.flags, 0, (size_t)grid * 4));
    CK(hipHostMalloc((void **)&sa.err, 4, hipHostMallocMapped)); *sa.err = 0;
    sa.budget = 200000000ull;     // 2 s
    sa.epoch = 0;
    sa.rounds = cum / (grid / 8);
    printf("%s (%s): %d tiles, %d workgroups one-per-tile | persistent grid %d: %d full rounds + %d tiles per XCD shared out as %.2f k-tiles per workgroup\n", name,
           STORE ? "gradient store" : "fused update", cum * 8, t, grid, sa.rounds, cum - sa.rounds * (grid / 8), (double)(cum - sa.rounds * (grid / 8)) * 16 / (grid / 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    // ---- results: NUP updates with each kernel on its own copy
    const int NUP = STORE ? 1 : 4;
    int bad_total = 0;
    for (int it = 0; it < NUP; ++it) {
        hipLaunchKernelGGL((bp_wgrad_dma<16, 4, 4, 256, STORE>), dim3(t), dim3(256), 0, st, ma);
        sa.epoch++;
        hipLaunchKernelGGL((bp_wgrad_sk<256, STORE>), dim3(grid), dim3(256), 0, st, sa);
        CK(hipStreamSynchronize(st));
        if (*sa.err) { printf("spin timeout, err word %u\n", *sa.err); return 1; }
        for (int i = 0; i < np; ++i) {
            std::vector<float> a(nw[i]), b(nw[i]);
            for (int which = 0; which < (STORE ? 1 : 2); ++which) {
                CK(hipMemcpy(a.data(), which ? Da[i] : Wa[i], nw[i] * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), which ? Db[i] : Wb[i], nw[i] * 4, hipMemcpyDeviceToHost));
                double mx = 0, ref = 0; size_t nbad = 0, nexact = 0;
                double ref_scale = 0; for (size_t k = 0; k < nw[i]; k += 97) ref_scale = std::max(ref_scale, fabs((double)a[k]));
                for (size_t k = 0; k < nw[i]; ++k) {
                    const double d = fabs((double)a[k] - b[k]); mx = std::max(mx, d); ref = std::max(ref, fabs((double)a[k]));
                    if (a[k] == b[k]) ++nexact;
                    if (!(d <= 1e-5 * (1e-3 + fabs((double)a[k])) + 2e-6 * ref_scale)) ++nbad;
                }
                printf("  update %d problem %d %s: max |diff| %.3e (max |ref| %.3e), identical words %.1f %%, out of tolerance %zu\n", it, i,
                       STORE ? "G" : (which ? "delta" : "W"), mx, ref, 100.0 * nexact / nw[i], nbad);
                bad_total += (int)std::min<size_t>(nbad, 1000000);
            }
            std::vector<float> x(2 * cur[i]), y(2 * cur[i]);
            CK(hipMemcpy(x.data(), ba[i], x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), bb[i], y.size() * 4, hipMemcpyDeviceToHost));
            double mx = 0; for (size_t k = 0; k < x.size(); ++k) mx = std::max(mx, fabs((double)x[k] - y[k]));
            printf("  update %d problem %d bias/delta_bias: max |diff| %.3e\n", it, i, mx);
            if (mx > 1e-5) ++bad_total;
        }
    }
    // ---- race screen: many launches, compare again at the end (the two copies see the same sequence)
    for (int it = 0; it < 200; ++it) {
        hipLaunchKernelGGL((bp_wgrad_dma<16, 4, 4, 256, STORE>), dim3(t), dim3(256), 0, st, ma);
        sa.epoch++;
        hipLaunchKernelGGL((bp_wgrad_sk<256, STORE>), dim3(grid), dim3(256), 0, st, sa);
    }
    CK(hipStreamSynchronize(st));
    if (*sa.err) { printf("spin timeout, err word %u\n", *sa.err); return 1; }
    for (int i = 0; i < np; ++i) {
        std::vector<float> a(nw[i]), b(nw[i]);
        CK(hipMemcpy(a.data(), Wa[i], nw[i] * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Wb[i], nw[i] * 4, hipMemcpyDeviceToHost));
        double mx = 0, ref = 0; size_t nbad = 0;
        for (size_t k = 0; k < nw[i]; ++k) {
            const double d = fabs((double)a[k] - b[k]); mx = std::max(mx, d); ref = std::max(ref, fabs((double)a[k]));
            if (!(d <= 2e-4 * (1e-2 + fabs((double)a[k])))) ++nbad;
        }
        printf("  after 200 more launches, problem %d W: max |diff| %.3e (max |ref| %.3e), out of tolerance %zu\n", i, mx, ref, nbad);
        bad_total += (int)std::min<size_t>(nbad, 1000000);
    }
    // ---- timing: interleaved rounds of 50 back-to-back launches
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ta, tb;
    for (int round = 0; round < 8; ++round) {
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < 50; ++it) {
                if (which == 0) hipLaunchKernelGGL((bp_wgrad_dma<16, 4, 4, 256, STORE>), dim3(t), dim3(256), 0, st, ma);
                else { sa.epoch++; hipLaunchKernelGGL((bp_wgrad_sk<256, STORE>), dim3(grid), dim3(256), 0, st, sa); }
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round >= 2) (which ? tb : ta).push_back(ms / 50 * 1000);
        }
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
#ifdef BP_SK_TRACE
    {   // per-workgroup timeline of ONE launch: code 0 = start, 1 = piece published, 2 = whole tile (unrolled body), 3 = whole tile (loop body), 4 = shared tile finished
        CK(hipMalloc(&sa.trace, (size_t)grid * 16 * 8)); CK(hipMemset(sa.trace, 0, (size_t)grid * 16 * 8));
        sa.epoch++; hipLaunchKernelGGL((bp_wgrad_sk<256, STORE>), dim3(grid), dim3(256), 0, st, sa);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)grid * 16);
        CK(hipMemcpy(h.data(), sa.trace, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < grid; ++b) for (int k = 0; k < 15; ++k) if (h[b * 16 + k]) { lo = std::min(lo, h[b * 16 + k] >> 4); hi = std::max(hi, h[b * 16 + k] >> 4); }
        double sum[5] = {0}, mx[5] = {0}; long cnt[5] = {0};
        double start_max = 0, end_min = 1e30, end_max = 0;
        for (int b = 0; b < grid; ++b) {
            start_max = std::max(start_max, ((h[b * 16] >> 4) - lo) * 0.01);
            for (int k = 1; k < 15 && h[b * 16 + k]; ++k) {
                const int code = (int)(h[b * 16 + k] & 15); const double d = ((h[b * 16 + k] >> 4) - (h[b * 16 + k - 1] >> 4)) * 0.01;
                sum[code] += d; cnt[code]++; mx[code] = std::max(mx[code], d);
                if (k == 14 || !h[b * 16 + k + 1]) { end_min = std::min(end_min, ((h[b * 16 + k] >> 4) - lo) * 0.01); end_max = std::max(end_max, ((h[b * 16 + k] >> 4) - lo) * 0.01); }
            }
        }
        printf("  TRACE span %.1f us; workgroup starts within %.2f us; ends in [%.1f, %.1f] us\n", (hi - lo) * 0.01, start_max, end_min, end_max);
        const char *nm[5] = {"", "piece published", "whole tile, unrolled body", "whole tile, loop body", "shared tile finished (own piece + neighbours')"};
        for (int c = 1; c < 5; ++c) if (cnt[c]) printf("  TRACE %-48s n %5ld avg %.2f us max %.2f us\n", nm[c], cnt[c], sum[c] / cnt[c], mx[c]);
        for (int b = 0; b < 3; ++b) { printf("  TRACE workgroup %d:", b * 9); for (int k = 0; k < 15 && h[(size_t)b * 9 * 16 + k]; ++k) printf(" %.1f(%d)", ((h[(size_t)b * 9 * 16 + k] >> 4) - lo) * 0.01, (int)(h[(size_t)b * 9 * 16 + k] & 15)); printf("\n"); }
        {   // workgroups that shared a CU: in the order they started, when they ended
            struct R { unsigned long long key; double s, e; int b; };
            std::vector<R> rs;
            for (int b = 0; b < grid; ++b) {
                const unsigned long long id = h[(size_t)b * 16 + 15]; const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 15;
                const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                double e = 0; for (int k = 1; k < 15 && h[(size_t)b * 16 + k]; ++k) e = ((h[(size_t)b * 16 + k] >> 4) - lo) * 0.01;
                rs.push_back({((unsigned long long)xcc << 12) | (se << 8) | (sh << 4) | cu, ((h[(size_t)b * 16] >> 4) - lo) * 0.01, e, b});
            }
            std::sort(rs.begin(), rs.end(), [](const R &x, const R &y) { return x.key != y.key ? x.key < y.key : x.s < y.s; });
            double spread = 0, first_last = 0; int ncu = 0; size_t i0 = 0; std::vector<int> hist(9, 0);
            double rank_end[8] = {0}; int rank_n[8] = {0};
            for (size_t i = 1; i <= rs.size(); ++i) if (i == rs.size() || rs[i].key != rs[i0].key) {
                double mn = 1e30, mxe = 0; for (size_t k = i0; k < i; ++k) { mn = std::min(mn, rs[k].e); mxe = std::max(mxe, rs[k].e); if (k - i0 < 8) { rank_end[k - i0] += rs[k].e; rank_n[k - i0]++; } }
                spread += mxe - mn; first_last += rs[i - 1].e - rs[i0].e; ++ncu; hist[std::min<size_t>(i - i0, 8)]++;
                if (ncu <= 4) { printf("  TRACE CU %llx:", rs[i0].key); for (size_t k = i0; k < i; ++k) printf(" wg %d start %.2f end %.1f |", rs[k].b, rs[k].s, rs[k].e); printf("\n"); }
                i0 = i;
            }
            printf("  TRACE %d distinct CU ids; workgroups per CU histogram:", ncu); for (int k = 1; k < 9; ++k) printf(" %d:%d", k, hist[k]); printf("\n");
            printf("  TRACE within a CU: end-time spread avg %.1f us; (last-started minus first-started) end avg %.1f us; mean end by start rank:", spread / ncu, first_last / ncu);
            for (int k = 0; k < 8 && rank_n[k]; ++k) printf(" %.1f", rank_end[k] / rank_n[k]); printf("\n");
        }
        sa.trace = nullptr;
    }
#endif
    printf("  TIMING %s %s: one-per-tile median %.2f us (min %.2f) | persistent k-balanced median %.2f us (min %.2f)\n", name, STORE ? "store" : "update",
           ta[ta.size() / 2], ta[0], tb[tb.size() / 2], tb[0]);
    printf("  RESULT %s %s: %s\n", name, STORE ? "store" : "update", bad_total ? "MISMATCH" : "ok");
    return bad_total ? 1 : 0;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "c2";
    const int grid = argc > 2 ? atoi(argv[2]) : 0;
    const int c2p[4] = {2880, 2048, 2048, 2048}, c2c[4] = {2048, 2048, 2048, 320};
    const int c3p[4] = {3136, 2048, 2048, 2048};
    int rc = 0;
    if (!strcmp(which, "c3")) rc |= run_case<false>("C3", c3p, c2c, 4, grid);
    else if (!strcmp(which, "store")) rc |= run_case<true>("C2", c2p, c2c, 4, grid);
    else if (!strcmp(which, "rest")) rc |= run_case<true>("C2 layers 2-4", c2p + 1, c2c + 1, 3, grid);
    else rc |= run_case<false>("C2", c2p, c2c, 4, grid);
    return rc;
}
