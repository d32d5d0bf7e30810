// tools/fwd_glds_probe.h -- DEVELOPMENT PROBE: M = 256 forward / dgrad GEMM with LDS-DMA staging of BOTH operands and an
// un-transposed, XOR-swizzled LDS image of the k-contiguous operand(s), read as ds_read_b128 (4 consecutive k per lane).
//   fwd   (B_KC = false): A = y_prev [m][k] (k contiguous), B = W [k][n] (n contiguous)
//   dgrad (B_KC = true) : A = dEdX  [m][k],                 B = W [n][k] (k contiguous)
// Tile 32 x 64 x 64, 4 waves = 2 (n halves) x 2 (k halves of every k-tile), 4-stage LDS ring (24 KB / stage), three
// k-tiles in flight, one raw s_barrier per k-tile with counted vmcnt.  k order inside a tile is permuted (lanes 0-31
// take k = 8t+j, lanes 32-63 k = 8t+4+j of MFMA step j) -- any pairing is a valid reduction order.
#pragma once
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"

template <bool B_KC, int EPI>
struct GemmDma {
    static constexpr int BM = 32, BN = 64, BK = 64, ST = 4, D = ST - 1, KS = 2;
    static constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK, STAGE = A_STAGE + B_STAGE;      // floats: 2048 + 4096
    static constexpr int SMEM = ST * STAGE;                                                     // 96 KB
    static constexpr int NA = A_STAGE / 1024, NB = B_STAGE / 1024, NDMA = NA + NB;              // per wave and k-tile: 2 + 4
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    static __device__ __forceinline__ void issue_tile(const GemmArgs &g, int m0, int n0, int k0, float *smem, int st, int wave, int lane)
    {
        const int r4 = lane >> 4, p = lane & 15;
#pragma unroll
        for (int i = 0; i < NA; ++i) {                       // A rows are k-contiguous: chunk p of row r holds global chunk p ^ (r & 15)
            const int q = wave * NA + i, r = q * 4 + r4;
            const float *ga = g.A + (size_t)(m0 + r) * g.lda + k0 + ((p ^ (r & 15)) << 2);
            __builtin_amdgcn_global_load_lds((glb_ptr)ga, (lds_ptr)(smem + st * STAGE + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = wave * NB + i, r = q * 4 + r4;
            const float *gb = B_KC ? g.B + (size_t)(n0 + r) * g.ldb + k0 + ((p ^ (r & 15)) << 2)      // rows = n, swizzled like A
                                   : g.B + (size_t)(k0 + r) * g.ldb + n0 + (p << 2);                  // rows = k, linear
            __builtin_amdgcn_global_load_lds((glb_ptr)gb, (lds_ptr)(smem + st * STAGE + A_STAGE + q * 256), 16, 0, 0);
        }
    }

    static __device__ __forceinline__ void multiply(const float *smem, int st, int ks, int wn, int lane, f32x16 (&acc)[2])
    {
        const int i = lane & 31, kh = lane >> 5;
        const float *As = smem + st * STAGE + i * BK, *Bs = smem + st * STAGE + A_STAGE;
        float4 a4[4], b4[4];
        float bv[4][4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cidx = ks * 8 + 2 * t + kh;
            a4[t] = *reinterpret_cast<const float4 *>(As + ((cidx ^ (i & 15)) << 2));
            if constexpr (B_KC) {
                const int n = wn * 32 + i;
                b4[t] = *reinterpret_cast<const float4 *>(Bs + n * BK + ((cidx ^ (n & 15)) << 2));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[t][j] = Bs[(ks * 32 + 8 * t + 4 * kh + j) * BN + wn * 32 + i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float av[4] = {a4[t].x, a4[t].y, a4[t].z, a4[t].w};
            float bw[4];
            if constexpr (B_KC) { bw[0] = b4[t].x; bw[1] = b4[t].y; bw[2] = b4[t].z; bw[3] = b4[t].w; }
            else { bw[0] = bv[t][0]; bw[1] = bv[t][1]; bw[2] = bv[t][2]; bw[3] = bv[t][3]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bw[j], acc[j & 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    static __device__ __forceinline__ void run(const GemmArgs &g, const EpiArgs &e, int b, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wn = wave & 1, ks = wave >> 1;
        int tile_m, tile_n;
        if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j / g.tiles_m; tile_m = j % g.tiles_m; }
        else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
        const int m0 = tile_m * BM, n0 = tile_n * BN, mb = m0, nb = n0 + wn * 32;
        const int nt = (g.K + BK - 1) / BK;
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        EpiPre pre;
        if (ks == 0) epilogue_fetch<EPI, 0, 8>(e, mb, nb, lane, pre); else epilogue_fetch<EPI, 8, 8>(e, mb, nb, lane, pre);
#pragma unroll
        for (int t = 0; t < D; ++t) issue_tile(g, m0, n0, t * BK, smem, t, wave, lane);
        int t = 0, st = 0;
        for (; t + D < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            issue_tile(g, m0, n0, (t + D) * BK, smem, st == 0 ? ST - 1 : st - 1, wave, lane);
            multiply(smem, st, ks, wn, lane, acc);
            st = st == ST - 1 ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        multiply(smem, st, ks, wn, lane, acc); st = st == ST - 1 ? 0 : st + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        multiply(smem, st, ks, wn, lane, acc); st = st == ST - 1 ? 0 : st + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        multiply(smem, st, ks, wn, lane, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
        __syncthreads();
        // in-workgroup k-split: wave group ks finishes registers [ks*8, ks*8+8) of the block (bp_kernels.h ksplit_give/take)
        if (ks == 0) ksplit_give<2, 0>(acc[0], smem, wn, lane); else ksplit_give<2, 1>(acc[0], smem, wn, lane);
        __syncthreads();
        if (ks == 0) { ksplit_take<2, 0>(acc[0], smem, wn, lane); epilogue_block<EPI, 0, 8>(e, mb, nb, acc[0], lane, pre); }
        else { ksplit_take<2, 1>(acc[0], smem, wn, lane); epilogue_block<EPI, 8, 8>(e, mb, nb, acc[0], lane, pre); }
    }
};

template <bool B_KC, int EPI>
__global__ __launch_bounds__(256, 1) void bp_gemm_dma(const GemmArgs g, const EpiArgs e)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmDma<B_KC, EPI>::SMEM];
    GemmDma<B_KC, EPI>::run(g, e, blockIdx.x, smem);
}
