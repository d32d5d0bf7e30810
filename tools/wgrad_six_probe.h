// tools/wgrad_six_probe.h -- development probe (NOT part of the library; measured and not adopted, profiles/r04_merged_dgrad_wgrad.txt): the six-wave
// form of the fp32 wgrad + update (two waves own the W / delta tile on their own vmcnt, as bp_wgrad_dma_bf16.h WgradDmaBf6 does for bf16).  It was
// wired into run_wgrads() behind an environment switch for the measurement; parity tests green under it.
#pragma once
#include "../dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h"
template <int KTOT, int MINWG>
struct WgradDma6 {
    using M = WgradDma<16, 4, MINWG, KTOT, true>;
    static constexpr int GLD = 68;
    static __device__ __forceinline__ void run(const GemmArgs &g, const EpiArgs &e, int b, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (b >= g.tiles_m * g.tiles_n) return;
        int tile_m, tile_n;
        if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j % per; tile_m = j / per; }
        else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
        const int m0 = tile_m * 64, n0 = tile_n * 64;
        const bool do_bias = tile_m == 0;
        if (wave >= 4) {
            const int u = wave - 4;
            float4 w4[8], d4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i, row = 32 * u + (c >> 4), col = (c & 15) * 4;
                const size_t o = (size_t)(m0 + row) * e.ldc + n0 + col;
                w4[i] = *reinterpret_cast<const float4 *>(e.C + o);
                d4[i] = *reinterpret_cast<const float4 *>(e.aux2 + o);
            }
#pragma unroll 1
            for (int t = 0; t < M::NT; ++t) __builtin_amdgcn_s_barrier();
            __syncthreads();
            if (do_bias) { __syncthreads(); __syncthreads(); }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i, row = 32 * u + (c >> 4), col = (c & 15) * 4;
                const float4 g4 = *reinterpret_cast<const float4 *>(smem + row * GLD + col);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, wv[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w}, dv[4] = {d4[i].x, d4[i].y, d4[i].z, d4[i].w};
                float dn[4], wn_[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { dn[j] = e.mom * dv[j] - e.c1 * (gv[j] / e.ndiv + e.wc * wv[j]); wn_[j] = dn[j] + 1.0f * wv[j]; }
                const size_t o = (size_t)(m0 + row) * e.ldc + n0 + col;
                *reinterpret_cast<float4 *>(e.aux2 + o) = make_float4(dn[0], dn[1], dn[2], dn[3]);
                *reinterpret_cast<float4 *>(e.C + o) = make_float4(wn_[0], wn_[1], wn_[2], wn_[3]);
            }
            return;
        }
        const int wm = wave >> 1, wn = wave & 1, mb = m0 + wm * 32, nb = n0 + wn * 32;
        const int a_off = wm * 32 + (lane & 31), b_off = wn * 32 + (lane & 31), kh = lane >> 5;
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        float bsum = 0.f;
        EpiPre pre;
#pragma unroll
        for (int t = 0; t < M::D; ++t) M::issue_tile(g, m0, n0, t * M::BK, smem, t, wave, lane);
        M::template iter<0>(g, e, m0, n0, smem, wave, lane, tid, a_off, b_off, kh, mb, nb, do_bias, bsum, acc, pre);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
        __syncthreads();
        if (do_bias) {
            float *red = smem;
            red[(tid >> 6) * 64 + (tid & 63)] = bsum;
            __syncthreads();
            if (tid < 64 && n0 + tid < e.n_limit) {
                const float s = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
                const int n = n0 + tid;
                const float d = e.mom * e.bias_d[n] - e.c1 * (s / e.ndiv + 0.0f * e.bias_w[n]);
                e.bias_d[n] = d;
                e.bias_w[n] = d + 1.0f * e.bias_w[n];
            }
            __syncthreads();
        }
        {
            const int nl = wn * 32 + (lane & 31), ml = wm * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[(ml + (r & 3) + 8 * (r >> 2)) * GLD + nl] = acc[0][r];
        }
        __syncthreads();
    }
};
template <int KTOT, int MINWG>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(MINWG == 4 ? 6 : 5, MINWG == 4 ? 6 : 5))) void bp_wgrad_dma_six(const MultiArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[WgradDma<16, 4, MINWG, KTOT, true>::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    WgradDma6<KTOT, MINWG>::run(a.g[p], a.e[p], b - a.first_tile[p], smem);
}
