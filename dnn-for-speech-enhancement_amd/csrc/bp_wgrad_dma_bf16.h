// bp_wgrad_dma_bf16.h -- bf16 compute mode (BASELINE.json configs[4]): weight gradient + fused fp32 momentum update +
// bf16 shadow refresh + bias gradient of EVERY layer in one grouped launch, LDS-DMA staged.
//
//   G_l = Y_{l-1}^T . dEdX_l   (SgemmNT, DevFunc.h:57-67; BP_GPU.cu:642) with bf16 operands and fp32 accumulation
//   (v_mfma_f32_32x32x16_bf16), then kernUpdatedelta + kernAccSum (DevFunc.cu:313-318, 270-277) on the fp32 master
//   W / delta while the tile is in registers, the refreshed bf16 shadow of W, and the bias
//   gradient (kernAccSumrow, DevFunc.cu:224-242) by m-tile 0 from the dEdX panel it streams anyway.
//
// This is the structure of bp_wgrad_dma.h carried over to the bf16 operands.  Both operands are the TRANSPOSED
// bf16 copies the producing epilogues already write (yT [unit][frame], dxT [unit][frame]): k (= frames) contiguous, so
// a 64-row x 32-frame tile is 64 rows x 64 bytes and goes global -> LDS by `global_load_lds_dwordx4`, one 1 KiB wave
// instruction per 16 rows.  The LDS image is lane-linear (DMA writes base + lane*16), so the bank swizzle lives in the
// SOURCE address: 16-byte slot s of row r holds k-chunk s ^ ((r >> 2) & 3); an MFMA fragment read (ds_read_b128 of one
// chunk per lane, 32 rows per half-wave) then touches all 64 banks once per 16-lane group.  64x64 tiles, 4 waves of
// one 32x32 block, 32-frame k-tiles in a 4-stage ring (32 KB => 4 workgroups per CU), three tiles in flight, ONE raw
// s_barrier per k-tile with an exact counted vmcnt.  Two kernels share that loop (WgradDmaBf): the data-parallel gradient
// STORE (four waves; tile and bias gradient go to the flat gradient buffer) and the fused update of the single-device step, the
// SIX-wave form at the end of this file (WgradDmaBf6: two more waves own W / delta on their own vmcnt; 64-frame k-tiles in a
// ring of 3 for bunches of 256 frames and more).  The step is HBM-bound here (18 bytes per parameter: fp32 W and delta read +
// written, one bf16 shadow written); what this buys over bp_gemm_bf16<BEPI_WGRAD_UPDATE> is occupancy (80 VGPRs), no register
// staging, one launch for all layers instead of one GEMM + one bias kernel per layer.
#pragma once
#include "bp_kernels.h"
#include "bp_bf16.h"
#include "bp_wgrad_dma.h"

struct BfWgradProblem {
    const bf16_t *A, *B;               // yT_{l-1} [pad64(prev)][ldk], dxT_l [pad64(cur)][ldk]; k = frame, contiguous
    int ldk;                           // halfs per operand row (bunch rows rounded up to 64)
    int tiles_m, tiles_n;
    EpiArgs e;                         // fp32 side exactly as bp_wgrad_dma: C = W (or G), aux2 = delta, bias_w/d/g, mom, c1, wc, ndiv
    bf16_t *Wb; int ldwb;              // refreshed bf16 shadow [prev][cur] (fused update only)
};
enum { BF_WGRAD_MAXP = 8 };
struct BfWgradMulti { BfWgradProblem p[BF_WGRAD_MAXP]; int first_tile[BF_WGRAD_MAXP + 1]; int n; };

// BKX = 64 (rows of 128 bytes = whole lines per DMA request instead of halves; slot s of row r holds chunk s ^ ((r>>1)&7)), STX = ring length
template <int KTOT, int BKX = 32, int STX = 4>
struct WgradDmaBf {
    static constexpr int BM = 64, BN = 64, BK = BKX, ST = STX, D = ST - 1, NT = KTOT / BK;
    static constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK, STAGE = A_STAGE + B_STAGE;     // halfs
    static constexpr int SMEM = ST * STAGE;                                                    // halfs (32 KB; 48 KB for 64-deep tiles x 3)
    static constexpr int CH = BK / 8, RPP = 512 / BK, NA = BM / RPP / 4;                       // 16-byte chunks per row, rows per 1 KiB piece, pieces per wave and operand
    static constexpr int NDMA = 2 * NA;
    static_assert((BK == 32 || BK == 64) && KTOT % BK == 0 && NT > D, "bunch rows");
    static __device__ __forceinline__ int swz(int r) { return BK == 32 ? (r >> 2) & 3 : (r >> 1) & 7; }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    static __device__ __forceinline__ void issue_tile(const BfWgradProblem &g, int m0, int n0, int k0, bf16_t *smem, int st, int wave, int lane)
    {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = wave * NA + i, r = q * RPP + lane / CH, c = (lane % CH) ^ swz(r);       // LDS slot lane % CH of row r <- k-chunk c
            __builtin_amdgcn_global_load_lds((glb_ptr)(g.A + (size_t)(m0 + r) * g.ldk + k0 + c * 8), (lds_ptr)(smem + st * STAGE + q * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(g.B + (size_t)(n0 + r) * g.ldk + k0 + c * 8), (lds_ptr)(smem + st * STAGE + A_STAGE + q * 512), 16, 0, 0);
        }
    }
    static __device__ __forceinline__ void multiply(const bf16_t *smem, int st, int ra, int rb, int kh, f32x16 &acc)
    {
        const bf16_t *ap = smem + st * STAGE + ra * BK, *bp = smem + st * STAGE + A_STAGE + rb * BK;
        const int sa = swz(ra), sb = swz(rb);
        constexpr int NQ = BK / 16;
        bf16x8_t a[NQ], b[NQ];
        // keep the fragment reads and their MFMAs inside this k-tile's barrier interval: the stage is refilled by DMA
        // right after the NEXT barrier, so every read of it must have completed (been consumed) before that barrier
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a[q] = *reinterpret_cast<const bf16x8_t *>(ap + (((2 * q + kh) ^ sa) * 8));
            b[q] = *reinterpret_cast<const bf16x8_t *>(bp + (((2 * q + kh) ^ sb) * 8));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[q], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int T>
    static __device__ __forceinline__ void iter(const BfWgradProblem &g, int m0, int n0, bf16_t *smem, int wave, int lane, int tid, int ra, int rb,
                                                int kh, bool do_bias, float &bsum, f32x16 &acc)
    {
        if constexpr (T < NT) {
            // in flight here: tiles T .. min(T+D, NT)-1
            constexpr int tiles_after = (T + D < NT ? D : NT - T) - 1;
            VmWait<tiles_after * NDMA>::go();
            __builtin_amdgcn_s_barrier();
            if constexpr (T + D < NT) issue_tile(g, m0, n0, (T + D) * BK, smem, (T + D) % ST, wave, lane);
            if (do_bias) {          // column sums of dEdX: row (tid >> 2) of the B tile, one 16-byte chunk per thread (any slot order)
                // (inline asm: for a plain LDS load next to in-flight LDS-DMA hipcc drains vmcnt(0) first, which would
                // serialise this workgroup's whole ring; the counted wait above already covers the stage read here)
#pragma unroll
                for (int cc = 0; cc < CH / 4; ++cc) {
                    uint4 u;
                    const unsigned la = (unsigned)(uintptr_t)(lds_ptr)(smem + (T % ST) * STAGE + A_STAGE + (tid >> 2) * BK + ((tid & 3) + 4 * cc) * 8);
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(u) : "v"(la) : "memory");
                    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bsum += __uint_as_float(w4[j] << 16); bsum += __uint_as_float(w4[j] & 0xFFFF0000u); }
                }
            }
            multiply(smem, T % ST, ra, rb, kh, acc);
            iter<T + 1>(g, m0, n0, smem, wave, lane, tid, ra, rb, kh, do_bias, bsum, acc);
        }
    }
    // the data-parallel gradient store: tile and bias gradient into the flat gradient buffer (e.C / e.bias_g)
    static __device__ __forceinline__ void run(const BfWgradProblem &g, int first_block, int stride, bf16_t *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
        const EpiArgs &e = g.e;
        for (int b = first_block; b < g.tiles_m * g.tiles_n; b += stride) {
            int tile_m, tile_n;
            if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j % per; tile_m = j / per; }
            else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
            const int m0 = tile_m * BM, n0 = tile_n * BN, mb = m0 + wm * 32, nb = n0 + wn * 32;
            const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), kh = lane >> 5;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bool do_bias = tile_m == 0;
            float bsum = 0.f;
#pragma unroll
            for (int t = 0; t < D; ++t) issue_tile(g, m0, n0, t * BK, smem, t, wave, lane);
            iter<0>(g, m0, n0, smem, wave, lane, tid, ra, rb, kh, do_bias, bsum, acc);
            if (do_bias) {
                bsum += __shfl_xor(bsum, 1);
                bsum += __shfl_xor(bsum, 2);
                const int n = n0 + (tid >> 2);
                if ((tid & 3) == 0 && n < e.n_limit) e.bias_g[n] = bsum;
            }
            // lane -> column n, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32x32 block.  Padded rows / columns
            // hold zeros in G: no predicates.
            const int n = nb + (lane & 31), rbase = mb + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) e.C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * e.ldc + n] = acc[r];
            if (b + stride < g.tiles_m * g.tiles_n) __syncthreads();      // the ring is refilled by the next tile's prologue
        }
    }
};

template <int KTOT>
__global__ __launch_bounds__(256, 4) void bp_wgrad_dma_bf16_store(const BfWgradMulti a)
{
    using K = WgradDmaBf<KTOT>;
    __shared__ __attribute__((aligned(16))) bf16_t smem[K::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    K::run(a.p[p], b - a.first_tile[p], a.first_tile[p + 1] - a.first_tile[p], smem);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Six waves per workgroup (fused update only): waves 0-3 run the operand ring and the MFMAs exactly as above, waves 4-5 own the fp32
// side of the tile.  Why: s_waitcnt vmcnt retires IN ORDER, so a wave that streams operand tiles cannot also keep its W / delta
// loads in flight through the k-loop (every tile wait would include them) -- they were issued behind the last operand tile, each
// workgroup had its 32 KB of HBM reads in flight for ~2.5 us of a ~24 us tile slot, and the launch was bound by that concurrency,
// not by bandwidth: forcing K to 128 / 256 / 512 frames gives 284 / 325 / 370 us, i.e. 263 us of streaming (6.2 TB/s) PLUS the
// k-loop.  An update wave has its own vmcnt: it issues the tile's W and delta loads (2 x 8 KB per wave, dwordx4) when the workgroup
// starts, sits through the k-loop's barriers, takes the gradient tile from LDS (the operand ring is free by then) and does
// kernUpdatedelta + kernAccSum + the shadow with the same expressions as above (results are bit-identical), storing whole
// 16-byte pieces of lines.  The fp32 W / delta stream (16 bytes per parameter, read and written once per step, 1.3 GB at
// configs[4]) goes through NONTEMPORAL loads and stores, the bf16 shadow through ordinary stores: the shadows of all layers
// (160 MB) then survive in the 256 MB Infinity Cache until the next step's forward reads them, instead of being flushed out
// by the master weights behind them -- forward GEMM 36.9 -> 28.5 us per launch, this launch 326 -> 305 us (same box A/B).
template <int KTOT, int BKX = 32, int STX = 4>
struct WgradDmaBf6 {
    using M = WgradDmaBf<KTOT, BKX, STX>;                            // the MFMA waves' loop (no W/delta traffic in it)
    static constexpr int GLD = 68;                         // floats per row of the staged gradient tile [64][68]
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    static_assert(64 * GLD * 4 <= M::SMEM * 2, "gradient tile fits the ring");

    static __device__ __forceinline__ void run(const BfWgradProblem &g, int b, bf16_t *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (b >= g.tiles_m * g.tiles_n) return;            // (padding workgroups of the grouped launch: all six waves leave)
        int tile_m, tile_n;
        if ((g.tiles_n & 7) == 0) { const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3; tile_n = xcd * per + j % per; tile_m = j / per; }
        else { tile_m = b % g.tiles_m; tile_n = b / g.tiles_m; }
        const int m0 = tile_m * 64, n0 = tile_n * 64;
        const EpiArgs &e = g.e;
        float *sg = reinterpret_cast<float *>(smem);
        if (wave >= 4) {
            // ---- update wave u: rows 32u .. 32u+31 of the tile; chunk c = lane + 64 i -> row c>>4, 4 floats at column 4*(c&15)
            const int u = wave - 4;
            float4 w4[8], d4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i, row = 32 * u + (c >> 4), col = (c & 15) * 4;
                const size_t o = (size_t)(m0 + row) * e.ldc + n0 + col;
                const nt_f4 a = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(e.C + o));
                const nt_f4 d = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(e.aux2 + o));
                w4[i] = make_float4(a[0], a[1], a[2], a[3]); d4[i] = make_float4(d[0], d[1], d[2], d[3]);
            }
#pragma unroll 1
            for (int t = 0; t < M::NT; ++t) __builtin_amdgcn_s_barrier();      // the k-loop's barriers (one per k-tile)
            __syncthreads();                                                    // every MFMA wave is past its last fragment read
            __syncthreads();                                                    // the gradient tile is in LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i, row = 32 * u + (c >> 4), col = (c & 15) * 4;
                const float4 g4 = *reinterpret_cast<const float4 *>(sg + row * GLD + col);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, wv[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w}, dv[4] = {d4[i].x, d4[i].y, d4[i].z, d4[i].w};
                float dn[4], wn_[4]; bf16_t hb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dn[j] = e.mom * dv[j] - e.c1 * (gv[j] / e.ndiv + e.wc * wv[j]);      // kernUpdatedelta
                    wn_[j] = dn[j] + 1.0f * wv[j];                                        // kernAccSum
                    hb[j] = f2bf(wn_[j]);
                }
                const size_t o = (size_t)(m0 + row) * e.ldc + n0 + col;
                const nt_f4 dst = {dn[0], dn[1], dn[2], dn[3]}, wst = {wn_[0], wn_[1], wn_[2], wn_[3]};
                __builtin_nontemporal_store(dst, reinterpret_cast<nt_f4 *>(e.aux2 + o));
                __builtin_nontemporal_store(wst, reinterpret_cast<nt_f4 *>(e.C + o));
                *reinterpret_cast<uint2 *>(g.Wb + (size_t)(m0 + row) * g.ldwb + n0 + col) =
                    make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
            }
            return;
        }
        // ---- MFMA waves
        const int wm = wave >> 1, wn = wave & 1;
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), kh = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool do_bias = tile_m == 0;
        float bsum = 0.f;
#pragma unroll
        for (int t = 0; t < M::D; ++t) M::issue_tile(g, m0, n0, t * M::BK, smem, t, wave, lane);
        M::template iter<0>(g, m0, n0, smem, wave, lane, tid, ra, rb, kh, do_bias, bsum, acc);
        __syncthreads();
        {   // gradient block -> LDS: lane -> column, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32x32 block
            const int nl = wn * 32 + (lane & 31), ml = wm * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) sg[(ml + (r & 3) + 8 * (r >> 2)) * GLD + nl] = acc[r];
        }
        __syncthreads();
        if (do_bias) {
            bsum += __shfl_xor(bsum, 1);
            bsum += __shfl_xor(bsum, 2);
            const int n = n0 + (tid >> 2);
            if ((tid & 3) == 0 && n < e.n_limit) {
                const float d = e.mom * e.bias_d[n] - e.c1 * (bsum / e.ndiv + 0.0f * e.bias_w[n]);
                e.bias_d[n] = d;
                e.bias_w[n] = d + 1.0f * e.bias_w[n];
            }
        }
    }
};

template <int KTOT, int BKX = 32, int STX = 4>
__global__ __launch_bounds__(384, 3) void bp_wgrad_dma_bf16_six(const BfWgradMulti a)
{
    __shared__ __attribute__((aligned(16))) bf16_t smem[WgradDmaBf<KTOT, BKX, STX>::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    WgradDmaBf6<KTOT, BKX, STX>::run(a.p[p], b - a.first_tile[p], smem);
}

