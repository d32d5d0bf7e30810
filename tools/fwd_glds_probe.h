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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the same tile and ring with the loop of the bf16 LDS-DMA GEMM (bp_bf16.h): every LDS read is inline asm, every wait
// counted by hand, the fragments of MFMA group G+2 are in flight while group G runs (a group = 4 MFMAs = one 16-byte A chunk
// per lane), the barrier of tile T+1 sits in the MIDDLE of tile T (behind group 1) so that groups 2 and 3 carry the six DMA
// pieces of tile T+3 and the first two fragment fetches of tile T+1: no MFMA ever waits behind a barrier + LDS round trip.
typedef float gd_f4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void gd_wait_f(gd_f4 &a, float &b0, float &b1, float &b2, float &b3)
{ asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N)); }
template <int N> __device__ __forceinline__ void gd_wait_d(gd_f4 &a, gd_f4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

template <bool B_KC, int EPI, int ABL = 0>
struct GemmDma2 {
    static constexpr int BM = 32, BN = 64, BK = 64, ST = 4, KS = 2;
    static constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK, STAGE = A_STAGE + B_STAGE;      // floats
    static constexpr int SMEM = ST * STAGE;                                                     // 96 KB
    static constexpr int NRD = B_KC ? 2 : 5;                                                    // LDS reads per group
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    static __device__ __forceinline__ void run(const GemmArgs &g, const EpiArgs &e, int b, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wn = wave & 1, ks = wave >> 1, i = lane & 31, kh = lane >> 5;
        int tile_m, tile_n;
        if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j / g.tiles_m; tile_m = j % g.tiles_m; }
        else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
        const int m0 = tile_m * BM, n0 = tile_n * BN, mb = m0, nb = n0 + wn * 32;
        const int nt = g.K / BK;
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        EpiPre pre;
        if (ks == 0) epilogue_fetch<EPI, 0, 8>(e, mb, nb, lane, pre); else epilogue_fetch<EPI, 8, 8>(e, mb, nb, lane, pre);
        // this wave's six pieces (1 KiB = 4 rows of 64 floats) of a tile: A pieces 2*wave + {0,1}, B pieces 4*wave + {0..3}
        const float *psrc[6]; size_t pstep[6];
        const int r4 = lane >> 4, p = lane & 15;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q < 2) { const int r = (wave * 2 + q) * 4 + r4; psrc[q] = g.A + (size_t)(m0 + r) * g.lda + ((p ^ (r & 15)) << 2); pstep[q] = BK; }
            else {
                const int r = (wave * 4 + q - 2) * 4 + r4;
                if constexpr (B_KC) { psrc[q] = g.B + (size_t)(n0 + r) * g.ldb + ((p ^ (r & 15)) << 2); pstep[q] = BK; }
                else { psrc[q] = g.B + (size_t)r * g.ldb + n0 + (p << 2); pstep[q] = (size_t)BK * g.ldb; }
            }
        }
#define GD_PIECE(q, T, st)                                                                                                        \
        do { if ((ABL & 1) && (T) >= 3) break; const int tt_ = (T) < nt ? (T) : nt - 1;                                            \
             __builtin_amdgcn_global_load_lds((glb_ptr)(psrc[q] + (size_t)tt_ * pstep[q]),                                        \
                 (lds_ptr)(smem + (st) * STAGE + ((q) < 2 ? (wave * 2 + (q)) * 256 : A_STAGE + (wave * 4 + (q) - 2) * 256)), 16, 0, 0); } while (0)
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)smem;
        unsigned aoff[4], boff[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cidx = ks * 8 + 2 * t + kh;
            aoff[t] = lds0 + 4u * (unsigned)(i * BK + ((cidx ^ (i & 15)) << 2));
            const int n = wn * 32 + i;
            boff[t] = B_KC ? lds0 + 4u * (unsigned)(A_STAGE + n * BK + ((cidx ^ (n & 15)) << 2))
                           : lds0 + 4u * (unsigned)(A_STAGE + (ks * 32 + 8 * t + 4 * kh) * BN + n);
        }
        gd_f4 fa[2], fb4[2]; float fb[2][4];
#define GD_READS(so, t, s)                                                                                                        \
        do { asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s]) : "v"(aoff[t] + (so)));                                             \
             if constexpr (B_KC) asm volatile("ds_read_b128 %0, %1" : "=v"(fb4[s]) : "v"(boff[t] + (so)));                        \
             else { const unsigned ba_ = boff[t] + (so);                                                                          \
                    asm volatile("ds_read_b32 %0, %1" : "=v"(fb[s][0]) : "v"(ba_));                                               \
                    asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(fb[s][1]) : "v"(ba_));                                    \
                    asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(fb[s][2]) : "v"(ba_));                                    \
                    asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(fb[s][3]) : "v"(ba_)); } } while (0)
#define GD_GROUP(s, N, P0, P1, P2, T, stn, RD)                                                                                    \
        do { if constexpr (B_KC) gd_wait_d<(N)>(fa[s], fb4[s]); else gd_wait_f<(N)>(fa[s], fb[s][0], fb[s][1], fb[s][2], fb[s][3]); \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             float bw_[4];                                                                                                        \
             if constexpr (B_KC) { bw_[0] = fb4[s][0]; bw_[1] = fb4[s][1]; bw_[2] = fb4[s][2]; bw_[3] = fb4[s][3]; }               \
             else { bw_[0] = fb[s][0]; bw_[1] = fb[s][1]; bw_[2] = fb[s][2]; bw_[3] = fb[s][3]; }                                  \
             acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], bw_[0], acc[0], 0, 0, 0);                                    \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             if ((P0) >= 0) GD_PIECE((P0) < 0 ? 0 : (P0), (T) + 3, stn);                                                          \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], bw_[1], acc[1], 0, 0, 0);                                    \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             if ((P1) >= 0) GD_PIECE((P1) < 0 ? 0 : (P1), (T) + 3, stn);                                                          \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][2], bw_[2], acc[0], 0, 0, 0);                                    \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             if ((P2) >= 0) GD_PIECE((P2) < 0 ? 0 : (P2), (T) + 3, stn);                                                          \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][3], bw_[3], acc[1], 0, 0, 0);                                    \
             __builtin_amdgcn_sched_barrier(0);                                                                                   \
             RD;                                                                                                                  \
             __builtin_amdgcn_sched_barrier(0); } while (0)
#pragma unroll
        for (int T = 0; T < ST - 1; ++T)
#pragma unroll
            for (int q = 0; q < 6; ++q) GD_PIECE(q, T, T);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GD_READS(0u, 0, 0); GD_READS(0u, 1, 1);
        for (int T = 0; T < nt; ++T) {
            const unsigned so = (unsigned)((T & 3) * STAGE * 4), so1 = (unsigned)(((T + 1) & 3) * STAGE * 4);
            const int stn = (T + 3) & 3;
            GD_GROUP(0, NRD, -1, -1, -1, T, stn, GD_READS(so, 2, 0));
            GD_GROUP(1, NRD, -1, -1, -1, T, stn, GD_READS(so, 3, 1));
            if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // tile T+1 has landed (tile T+2 may be in flight, T+3 is issued below)
            if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();                           // ... for every wave; every wave is done with the stage of tile T-1
            __builtin_amdgcn_sched_barrier(0);
            GD_GROUP(0, NRD, 0, 1, 2, T, stn, GD_READS(so1, 0, 0));
            GD_GROUP(1, NRD, 3, 4, 5, T, stn, GD_READS(so1, 1, 1));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef GD_PIECE
#undef GD_READS
#undef GD_GROUP
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
        __syncthreads();
        if (ks == 0) ksplit_give<2, 0>(acc[0], smem, wn, lane); else ksplit_give<2, 1>(acc[0], smem, wn, lane);
        __syncthreads();
        if (ks == 0) { ksplit_take<2, 0>(acc[0], smem, wn, lane); epilogue_block<EPI, 0, 8>(e, mb, nb, acc[0], lane, pre); }
        else { ksplit_take<2, 1>(acc[0], smem, wn, lane); epilogue_block<EPI, 8, 8>(e, mb, nb, acc[0], lane, pre); }
    }
};

template <bool B_KC, int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 1) void bp_gemm_dma2(const GemmArgs g, const EpiArgs e)
{
    __shared__ __attribute__((aligned(1024))) float smem[GemmDma2<B_KC, EPI, ABL>::SMEM];
    GemmDma2<B_KC, EPI, ABL>::run(g, e, blockIdx.x, smem);
}
