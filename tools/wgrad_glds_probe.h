// tools/wgrad_glds_probe.h -- DEVELOPMENT PROBE, not part of the product (the library uses GemmKernel<64,64,32,...> of
// bp_kernels.h): weight gradient + fused momentum update with LDS-DMA operand staging.  Bit-identical to the product
// kernel and equally fast (results and the ablation table it produced: DESIGN.md section 7).
//
//   G = Y_prev^T . dEdX  (SgemmNT, DevFunc.h:57-67; BP_GPU.cu:642), then kernUpdatedelta + kernAccSum
//   (DevFunc.cu:313-318, 270-277) on the tile while it is still in registers, bias gradient by m-tile 0
//   (kernAccSumrow, DevFunc.cu:224-242).
//
// Both operands of this GEMM are k-major in memory ([frame][unit]: the reduction runs over frames), which is
// exactly the LDS image the MFMA fragment fetch wants ([k][m] / [k][n], conflict-free ds_read_b32).  So the
// tiles can go global -> LDS by `global_load_lds_dwordx4` (1 KiB per wave instruction = 4 k-rows of a 64-wide
// tile) with no register staging and no ds_write at all: the MFMA waves only issue 4 DMA instructions per k-tile.
// 64x64x32 tiles, 3-stage LDS ring (48 KB => 3 workgroups per CU), two k-tiles in flight ahead of the one being
// multiplied, ONE raw s_barrier per k-tile with counted s_waitcnt vmcnt (a __syncthreads() would drain the DMA
// queue: it fences with vmcnt(0)).  The W / delta tile of the fused update is fetched by plain loads issued
// behind the LAST operand tile, so no operand wait ever includes it (vmcnt retires in order).
#pragma once
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"

#if defined(GLDS_ABLATE) && (GLDS_ABLATE & 64)
#define BP_VMCNT(n) ((void)0)
#else
#define BP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

// TM = 32x32 blocks per wave along m: workgroup tile (TM*64) x 64, k-tile 32/TM rows, so that every k-tile is 16 MFMAs
// per wave and 16 KB / 12 KB of DMA per workgroup either way.  TM = 2 halves the activation-panel traffic per FLOP, reads
// each dEdX fragment once for two MFMAs and doubles the work a workgroup does per prologue + epilogue.
template <int TM>
struct WgradGlds {
    static constexpr int BM = 64 * TM, BN = 64, BK = 32 / TM, ST = 3;
    static constexpr int A_STAGE = BK * BM, B_STAGE = BK * BN, STAGE = A_STAGE + B_STAGE;   // floats
    static constexpr int SMEM = ST * STAGE;                                                   // 48 KB (TM 1) / 36 KB (TM 2)
    static constexpr int MIN_WG = 3;
    static constexpr int NA = A_STAGE / 1024, NB = B_STAGE / 1024;       // DMA instructions per wave and k-tile (1 KiB each, 4 waves)
    static constexpr int NDMA = NA + NB;
    static_assert(A_STAGE % 1024 == 0 && B_STAGE % 1024 == 0, "tile rows must fill whole DMA instructions");

    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    // this wave's DMA instructions for k-tile at frame k0 into LDS stage `st`: instruction q of the operand covers
    // 1 KiB = (256 / width) consecutive k-rows of the tile
    static __device__ __forceinline__ void issue_tile(const GemmArgs &g, int m0, int n0, int k0, float *smem, int st, int wave, int lane)
    {
#if defined(GLDS_ABLATE) && (GLDS_ABLATE & 64)     // timing experiments only: no operand loads at all
        return;
#endif
        constexpr int RA = 256 / BM, RB = 256 / BN;                     // k-rows per DMA instruction (BM = 64: 4, 128: 2)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = wave * NA + i, r = q * RA + lane / (BM / 4), c = (lane % (BM / 4)) * 4;
            const float *ga = g.A + (size_t)(k0 + r) * g.lda + m0 + c;
            __builtin_amdgcn_global_load_lds((glb_ptr)ga, (lds_ptr)(smem + st * STAGE + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = wave * NB + i, r = q * RB + lane / (BN / 4), c = (lane % (BN / 4)) * 4;
            const float *gb = g.B + (size_t)(k0 + r) * g.ldb + n0 + c;
            __builtin_amdgcn_global_load_lds((glb_ptr)gb, (lds_ptr)(smem + st * STAGE + A_STAGE + q * 256), 16, 0, 0);
        }
    }

    // one k-tile on LDS stage `st`: BK/2 k-steps x TM blocks = 16 MFMAs, fragments fetched RD steps ahead
    static __device__ __forceinline__ void multiply(const float *smem, int st, int a_off, int b_off, int kh, f32x16 (&acc)[2])
    {
        constexpr int NK = BK / 2, RD = NK < 4 ? NK : 4;
        const float *ap = smem + st * STAGE + kh * BM + a_off, *bp = smem + st * STAGE + A_STAGE + kh * BN + b_off;
        float av[NK][TM], bv[NK];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < RD; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[s][i] = ap[2 * s * BM + i * 32];
            bv[s] = bp[2 * s * BN];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if constexpr (TM == 1) {
                acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0], bv[s], acc[s & 1], 0, 0, 0);     // two chains on one block
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0], bv[s], acc[0], 0, 0, 0);             // two blocks = two chains
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1], bv[s], acc[1], 0, 0, 0);
            }
            if (s + RD < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[s + RD][i] = ap[2 * (s + RD) * BM + i * 32];
                bv[s + RD] = bp[2 * (s + RD) * BN];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    static __device__ __forceinline__ void bias_rows(const float *smem, int st, int tid, float &bsum)
    {
        constexpr int RPT = BK / 4;                                            // k-rows per thread group (4 groups of 64 columns)
        const float *bs = smem + st * STAGE + A_STAGE + (tid >> 6) * RPT * BN + (tid & 63);
#pragma unroll
        for (int k = 0; k < RPT; ++k) bsum += bs[k * BN];
    }

    // One problem; workgroups first_block, first_block + stride, ... walk its tiles (same XCD-aware map as GemmKernel).
    // K must be a multiple of BK with at least 2 k-tiles (the host pads the bunch to 64 rows; rows past the bunch are 0).
    static __device__ __forceinline__ void run(const GemmArgs &g, const EpiArgs &e, int first_block, int stride, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
        const int nt = g.K / BK;
        for (int b = first_block; b < g.tiles_m * g.tiles_n; b += stride) {
            int tile_m, tile_n;
            if ((g.tiles_n & 7) == 0) {
                const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3;
                tile_n = xcd * per + j % per; tile_m = j / per;                 // n-panels fastest inside an XCD (BP_WGRAD_NFAST)
            } else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
            const int m0 = tile_m * BM, n0 = tile_n * BN;
            const int mb = m0 + wm * TM * 32, nb = n0 + wn * 32;
            const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * 32 + (lane & 31), kh = lane >> 5;
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            const bool do_bias = tile_m == 0;
            float bsum = 0.f;
            EpiPre pre[TM];

            issue_tile(g, m0, n0, 0, smem, 0, wave, lane);
            issue_tile(g, m0, n0, BK, smem, 1, wave, lane);
            // steady state: at the top of iteration t tiles t and t+1 are in flight (NDMA instructions each per wave)
            int t = 0, st = 0;
            for (; t + 2 < nt; ++t) {
                if constexpr (NDMA == 4) BP_VMCNT(4); else BP_VMCNT(3);        // this wave's pieces of tile t have landed
#if !(defined(GLDS_ABLATE) && (GLDS_ABLATE & 128))   // (128: timing experiments only: no barrier in the steady-state loop)
                __builtin_amdgcn_s_barrier();                                    // ... everybody's; and stage (t+2)%3 is no longer read
#endif
                issue_tile(g, m0, n0, (t + 2) * BK, smem, st == 0 ? 2 : st - 1, wave, lane);
                if (do_bias) bias_rows(smem, st, tid, bsum);
                multiply(smem, st, a_off, b_off, kh, acc);
                st = st == 2 ? 0 : st + 1;
            }
            // W / delta of this wave's blocks: 32 plain loads each, younger than every operand tile
#pragma unroll
            for (int i = 0; i < TM; ++i) epilogue_fetch<EPI_WGRAD_UPDATE, 0, 16>(e, mb + i * 32, nb, lane, pre[i]);
            if constexpr (TM == 1) BP_VMCNT(36); else BP_VMCNT(63);              // tile nt-2 landed (tile nt-1 + W/delta may be in flight; vmcnt saturates at 63)
            __builtin_amdgcn_s_barrier();
            if (do_bias) bias_rows(smem, st, tid, bsum);
            multiply(smem, st, a_off, b_off, kh, acc);
            st = st == 2 ? 0 : st + 1;
            if constexpr (TM == 1) BP_VMCNT(32); else BP_VMCNT(63);              // tile nt-1 landed
            __builtin_amdgcn_s_barrier();
            if (do_bias) bias_rows(smem, st, tid, bsum);
            multiply(smem, st, a_off, b_off, kh, acc);
            if constexpr (TM == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
            }
            __syncthreads();                                                     // (drains everything: the ring is free again)
            // ---- bias gradient: column sums of the dEdX panel (kernAccSumrow), update of b by m-tile 0
            if (do_bias) {
                float *red = smem;                                               // [4][64]
                red[(tid >> 6) * BN + (tid & 63)] = bsum;
                __syncthreads();
                if (tid < BN && n0 + tid < e.n_limit) {
                    const float s = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
                    const int n = n0 + tid;
                    const float d = e.mom * e.bias_d[n] - e.c1 * (s / e.ndiv + 0.0f * e.bias_w[n]);
                    e.bias_d[n] = d;
                    e.bias_w[n] = d + 1.0f * e.bias_w[n];
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) epilogue_block<EPI_WGRAD_UPDATE, 0, 16>(e, mb + i * 32, nb, acc[i], lane, pre[i]);
        }
    }
};

template <int TM>
__global__ __launch_bounds__(256, WgradGlds<TM>::MIN_WG) void bp_wgrad_glds_multi(const MultiArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[WgradGlds<TM>::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    WgradGlds<TM>::run(a.g[p], a.e[p], b - a.first_tile[p], a.first_tile[p + 1] - a.first_tile[p], smem);
}

// (the adopted static-K form of this design is the product kernel: dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h)
