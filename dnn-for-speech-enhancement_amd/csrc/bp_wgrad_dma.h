// bp_wgrad_dma.h -- weight gradient + fused momentum update with LDS-DMA operand staging (static bunch sizes).
//
//   G = Y_prev^T . dEdX  (SgemmNT, DevFunc.h:57-67; BP_GPU.cu:642), then kernUpdatedelta + kernAccSum
//   (DevFunc.cu:313-318, 270-277) on the tile while it is still in registers, bias gradient by m-tile 0
//   (kernAccSumrow, DevFunc.cu:224-242).
//
// Both operands of this GEMM are k-major in memory ([frame][unit]: the reduction runs over the frames of the bunch),
// which is exactly the LDS image the MFMA fragment fetch wants ([k][m] / [k][n], conflict-free ds_read_b32).  So the
// tiles go global -> LDS by `global_load_lds_dwordx4` (1 KiB per wave instruction = 4 k-rows of a 64-wide tile): no
// register staging, no ds_write, 116 VGPRs => 4 workgroups per CU instead of 3.  64x64 tiles, 16-row k-tiles in a
// 4-stage LDS ring (32 KB), three k-tiles in flight ahead of the one being multiplied, ONE raw s_barrier per k-tile with
// an exact, counted s_waitcnt vmcnt (a __syncthreads() would fence with vmcnt(0) and drain the DMA queue); the k-loop is
// fully unrolled for the bunch size, so every count is an immediate.  The W / delta tile of the fused update is
// fetched by plain loads issued right behind the LAST operand tile: no operand wait ever includes it (vmcnt retires in
// order) and it has three k-tiles of MFMA work to land.  Measured on C2 (grouped launch of all four layers, in-step):
// 80.4 us vs 84.7 us for the register-staged GemmKernel<64,64,32,...> it replaces; results are bit-identical to it
// for W and delta (same k-order, same two accumulator chains).  Other shapes of this design that were measured and
// not adopted (128x64 tiles, dynamic k-loop, 3/6-stage rings) live in tools/wgrad_glds_probe.h.
#pragma once
#include "bp_kernels.h"

// template <BK, ST, MINWG, K>: K = frames of the bunch (static).  Prefetch distance D = ST-1 tiles.
template <int N> struct VmWait { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); } };

template <int BKX, int STX, int MINWG, int KTOT, bool STORE = false>
struct WgradDma {
    // STORE: data-parallel form -- the gradient tile and the bias gradient go to the flat gradient buffer (e.C / e.bias_g),
    // no W / delta traffic; otherwise the fused momentum update
    static constexpr int EPI = STORE ? EPI_WGRAD_STORE : EPI_WGRAD_UPDATE;
    static constexpr int NWD = STORE ? 0 : 32;                    // plain loads of the W / delta tile per lane
    static constexpr int BM = 64, BN = 64, BK = BKX, ST = STX, NT = KTOT / BKX, D = STX - 1;
    static constexpr int A_STAGE = BK * BM, B_STAGE = BK * BN, STAGE = A_STAGE + B_STAGE;
    static constexpr int SMEM = ST * STAGE;
    static constexpr int MIN_WG = MINWG;
    static constexpr int NA = A_STAGE / 1024, NB = B_STAGE / 1024, NDMA = NA + NB;
    static_assert(NA >= 1 && NB >= 1 && NT > D, "shape");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    static __device__ __forceinline__ void issue_tile(const GemmArgs &g, int m0, int n0, int k0, float *smem, int st, int wave, int lane)
    {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = wave * NA + i, r = q * 4 + (lane >> 4), c = (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)(g.A + (size_t)(k0 + r) * g.lda + m0 + c), (lds_ptr)(smem + st * STAGE + q * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = wave * NB + i, r = q * 4 + (lane >> 4), c = (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)(g.B + (size_t)(k0 + r) * g.ldb + n0 + c), (lds_ptr)(smem + st * STAGE + A_STAGE + q * 256), 16, 0, 0);
        }
    }
    static __device__ __forceinline__ void multiply(const float *smem, int st, int a_off, int b_off, int kh, f32x16 (&acc)[2])
    {
        constexpr int NK = BK / 2, RD = NK < 4 ? NK : 4;
        const float *ap = smem + st * STAGE + kh * BM + a_off, *bp = smem + st * STAGE + A_STAGE + kh * BN + b_off;
        float av[NK], bv[NK];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < RD; ++s) { av[s] = ap[2 * s * BM]; bv[s] = bp[2 * s * BN]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[s & 1], 0, 0, 0);
            if (s + RD < NK) { av[s + RD] = ap[2 * (s + RD) * BM]; bv[s + RD] = bp[2 * (s + RD) * BN]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <int T>
    static __device__ __forceinline__ void iter(const GemmArgs &g, const EpiArgs &e, int m0, int n0, float *smem, int wave, int lane, int tid,
                                                int a_off, int b_off, int kh, int mb, int nb, bool do_bias, float &bsum, f32x16 (&acc)[2], EpiPre &pre)
    {
        if constexpr (T < NT) {
            // in flight at this point: tiles T .. min(T+D, NT)-1, plus the 32 W/delta loads once the last tile has been issued
            constexpr int tiles_after = (T + D < NT ? D : NT - T) - 1;
            constexpr bool wd_out = T + D > NT;                       // W/delta were issued in an earlier iteration (right after tile NT-1)
            VmWait<tiles_after * NDMA + (wd_out ? NWD : 0)>::go();
            __builtin_amdgcn_s_barrier();
            if constexpr (T + D < NT) issue_tile(g, m0, n0, (T + D) * BK, smem, (T + D) % ST, wave, lane);
            if constexpr (T + D == NT && !STORE) epilogue_fetch<EPI_WGRAD_UPDATE, 0, 16>(e, mb, nb, lane, pre);   // behind the last tile (issued at T-1)
            if (do_bias) {
                constexpr int RPT = BK / 4;
                const float *bs = smem + (T % ST) * STAGE + A_STAGE + (tid >> 6) * RPT * BN + (tid & 63);
#pragma unroll
                for (int k = 0; k < RPT; ++k) bsum += bs[k * BN];
            }
            multiply(smem, T % ST, a_off, b_off, kh, acc);
            iter<T + 1>(g, e, m0, n0, smem, wave, lane, tid, a_off, b_off, kh, mb, nb, do_bias, bsum, acc, pre);
        }
    }
    static __device__ __forceinline__ void run(const GemmArgs &g, const EpiArgs &e, int first_block, int stride, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
        for (int b = first_block; b < g.tiles_m * g.tiles_n; b += stride) {
            int tile_m, tile_n;
            if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j % per; tile_m = j / per; }
            else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
            const int m0 = tile_m * BM, n0 = tile_n * BN, mb = m0 + wm * 32, nb = n0 + wn * 32;
            const int a_off = wm * 32 + (lane & 31), b_off = wn * 32 + (lane & 31), kh = lane >> 5;
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            const bool do_bias = tile_m == 0;
            float bsum = 0.f;
            EpiPre pre;
#pragma unroll
            for (int t = 0; t < D; ++t) issue_tile(g, m0, n0, t * BK, smem, t, wave, lane);
            iter<0>(g, e, m0, n0, smem, wave, lane, tid, a_off, b_off, kh, mb, nb, do_bias, bsum, acc, pre);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
            __syncthreads();
            if (do_bias) {
                float *red = smem;
                red[(tid >> 6) * BN + (tid & 63)] = bsum;
                __syncthreads();
                if (tid < BN && n0 + tid < e.n_limit) {
                    const float s = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
                    const int n = n0 + tid;
                    if constexpr (STORE) {
                        __hip_atomic_store(e.bias_g + n, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (write-through, like the tile below)
                    } else {
                        const float d = e.mom * e.bias_d[n] - e.c1 * (s / e.ndiv + 0.0f * e.bias_w[n]);
                        e.bias_d[n] = d;
                        e.bias_w[n] = d + 1.0f * e.bias_w[n];
                    }
                }
                __syncthreads();
            }
            epilogue_block<EPI, 0, 16>(e, mb, nb, acc[0], lane, pre);
            if constexpr (STORE) {
                // data-parallel step: tell the exchange stream that this tile of the layer's gradient segment is complete, WITHOUT a
                // kernel boundary (one grouped launch for all layers; the exchange of layer 1 starts while the other layers' tiles
                // still run).  The tile and the bias gradient were stored with system-scope write-through stores (sc0 sc1,
                // epilogue_block): every storing wave drains vmcnt -- an acknowledged write-through store is in memory --, the
                // workgroup meets at a barrier, one lane counts the tile (relaxed, system scope).  That is the store half of the memory
                // model's own code sequence for system-scope atomics; it needs no L2 write-back because nothing was left dirty.
                // Measured alternatives (C2 through the exchange path at world 1, profiles/r05_dp_world1.txt): a system-scope RELEASE on
                // the count (ADVICE r4's first proposal: one buffer_wbl2 per tile, 3648 per step) 0.3445 ms against 0.2533; one release
                // per layer on every XCD inside bp_dp_sync 0.2750 (16 waves) / 0.2640 (8) against 0.2551.
                if (e.done) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_fetch_add(e.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
};

template <int BKX, int STX, int MINWG, int KTOT, bool STORE = false>
__global__ __launch_bounds__(256, MINWG) void bp_wgrad_dma(const MultiArgs a)
{
    using K = WgradDma<BKX, STX, MINWG, KTOT, STORE>;
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    K::run(a.g[p], a.e[p], b - a.first_tile[p], a.first_tile[p + 1] - a.first_tile[p], smem);
}
