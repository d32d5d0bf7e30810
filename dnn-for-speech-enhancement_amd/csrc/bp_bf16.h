// bp_bf16.h -- the same training step with bf16 GEMM operands (BASELINE.json configs[4]: "bf16 forward/backward,
// fp32 master weights").  fp32 stays the type of: accumulation (v_mfma_f32_32x32x16_bf16), bias,
// the activation / loss arithmetic of the epilogues, the master weights W, the momentum state delta and the update.
// bf16 (round-to-nearest-even) is the storage type of everything a GEMM reads: activations y_l, back-propagated
// errors dEdX_l and a shadow copy of the weights that the update epilogue refreshes.
//
// Layout idea: every GEMM of the step is  C[m][n] = sum_k A[m][k] * B[n][k].  Activations and back-propagated errors are
// kept in two orientations, written together by the producing epilogue (a lane owns one column and 4-row groups of a 32x32
// MFMA block, so the transposed copy is written as 8-byte runs), which makes their operands k-contiguous everywhere:
//     fwd   l : y_l[f][c]      = act( y_{l-1}[f][p] . Wb_l[p][c] )           y  : [frames][units]   yT : [units][frames]
//     dgrad l : dx_{l-1}[f][p] = act'(y_{l-1}) * ( dx_l[f][c] . Wb_l[p][c] ) dx : [frames][units]   dxT: [units][frames]
//     wgrad l : G_l[p][c]      = yT_{l-1}[p][f] . dxT_l[c][f]                Wb : [prev][cur]  (ONE shadow of the weights)
// The weights have ONE bf16 shadow, in the reference's own layout [prev][cur].  For dgrad that is k-contiguous (k = cur).
// For the forward k = prev runs down the rows: its B tile goes into LDS as it lies in memory ([k][n], full 128-byte
// lines) and the MFMA fragments -- 8 consecutive k of one column per lane -- come out of it through the hardware
// transpose read ds_read_b64_tr_b16 (two per fragment).  Round 3 kept a second, transposed shadow instead, which the
// HBM-bound weight-gradient launch had to write: 2 bytes per parameter and step (160 MB at configs[4]).
// One kernel template (128/64/32 x 64 x 64 tiles, 4 waves) serves forward, dgrad and the per-layer wgrad fallback with different
// epilogues, in two staging forms: register-staged (global -> VGPR -> ds_write_b128 into padded LDS rows; every shape) and LDS-DMA
// (DMA = true; the 128-row tiles of the 4096-wide layers of configs[4]).  The grouped update launch is bp_wgrad_dma_bf16.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16_t f2bf(float f)          // round to nearest even; NaN stays NaN
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }

enum { BEPI_FWD_HIDDEN = 0, BEPI_FWD_OUT = 1, BEPI_DGRAD = 2, BEPI_WGRAD_UPDATE = 3, BEPI_WGRAD_STORE = 4 };

struct BfGemmArgs {
    const bf16_t *A, *B;       // A [M][lda], B [N][ldb], both k-contiguous; rows padded to 64, k padded to 64 (zeros)
    int lda, ldb, K;           // K: multiple of 64
    int tiles_m, tiles_n;
    int k_rot;                 // LDS-DMA form: k-tile offset between the m-tiles of one n-tile
    float *ks_slab; unsigned *ks_cnt;   // KS > 1 (k range split over KS workgroups per tile): partial tiles [tile][KS][reg][thread], one ticket word per tile
};
struct BfEpiArgs {
    int m_limit, n_limit, n_true;      // rows / cols that exist (padded extents), unpadded column count
    // outputs in both orientations (bf16): C [M][ldc] and CT [N][ldct]
    bf16_t *C, *CT; int ldc, ldct;
    const float *bias; float alpha; int act;                       // fwd
    const float *targ; int ldt; float *out; int ldo; float scale;  // fwd_out: targets, optional fp32 output, 2/n
    const bf16_t *yprev; int ldy;                                  // dgrad: y_{l-1} (post-dropout)
    float *W, *D; int ldw; float mom, c1, wc, ndiv;                // wgrad: fp32 master W / delta (update) or G (store, in W)
    uint32_t drop_thresh, seed_lo, seed_hi, step, layer; int frame_off;
};

static constexpr int BF_BN = 64, BF_BK = 64, BF_LDS = BF_BK + 8;   // LDS row stride in halfs (144 B: conflict-free b128)
static constexpr int BF_NPF = 3;                                     // k-tiles in flight in registers

// BM = 64: 256 threads, waves 2 x 2.  BM = 32: 128 threads, waves 1 x 2 -- twice the workgroups for the M = bunch GEMMs.
// The MFMA work per k-tile is tiny (4 x v_mfma_f32_32x32x16_bf16 per wave), so the loop is bound by the latency of the
// tile loads: BF_NPF tiles are kept in flight in registers, two LDS stages, one barrier per k-tile.
// BM = 128: 256 threads, waves 2 x 2, each wave TWO 32x32 blocks along m (64 x 32): a third less operand traffic per
// FLOP than 64 x 64 and every B fragment feeds two MFMAs -- used when 128-row tiles still give every CU a workgroup
// (the 4096-wide layers of configs[4]: these GEMMs are bound by operand delivery into the CUs, not by the bf16 MFMA rate).
// BKN: the B operand lies [k][n] in memory (n contiguous; the forward's weights Wb[prev][cur]) instead of [n][k]: its
// tile is staged as 64 k-rows of 64 n (BF_LDB halfs apart: 192 bytes, so that the 4 k-rows x 64 bytes a half-wave's
// transpose read touches fall into the four quarters of the 256-byte bank row) and read with ds_read_b64_tr_b16.
static constexpr int BF_LDB = 96;
typedef short bf_v4s __attribute__((ext_vector_type(4)));
typedef short bf_v8s __attribute__((ext_vector_type(8)));
typedef float bf_f32x4 __attribute__((ext_vector_type(4)));
// DMA (BM = 128 only; the 4096-wide GEMMs of configs[4]): the operand tiles go global -> LDS by global_load_lds_dwordx4 instead of
// through registers + ds_write_b128 (13 LDS cycles per wave instruction, 312 per k-tile: more than the tile's 256 MFMA cycles).
// One k-tile = 192 rows (128 of A, 64 of B) x 128 bytes = 24 pieces of 1 KiB (8 rows each), 6 per wave; ring of 4 stages, three
// tiles in flight, ONE raw s_barrier per k-tile behind a counted s_waitcnt vmcnt.  Bank swizzle in the SOURCE address of the DMA:
// chunk c (16 bytes) of row r sits in slot c ^ ((r>>1)&7) (ds_read_b128 fragments: the 16 lanes of a read group cover both row
// parities x 8 slots), and for the [k][n] tile of the forward chunk c of k-row k in slot c ^ (4*((k>>1)&1)) (ds_read_b64_tr_b16:
// the 4 k-rows x 64 bytes of a half-wave cover all 64 banks).  Every LDS read of the loop is inline asm and every wait is counted
// by hand (reads return in order): the fragments of k-step q+1 are in flight while the MFMAs of step q run, the six DMA pieces of
// tile t+3 are issued between them.  (Through the builtin, the compiler puts s_waitcnt vmcnt(0) in front of every transpose read
// -- it orders them behind ALL pending LDS-DMA -- which serialises the loop: 70.9 us.)  tools/bf16_gemm_probe.hip holds the
// measured steps from the register-staged loop (35.5-36.9 us per 512 x 4096 x 4096 GEMM) to this one (22.3-23.3 us).
template <int N> __device__ __forceinline__ void bf_lgkm3(bf_f32x4 &a, bf_f32x4 &b, bf_f32x4 &c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }
template <int N> __device__ __forceinline__ void bf_lgkm4(bf_f32x4 &a, bf_f32x4 &b, bf_v4s &c, bf_v4s &d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); }
// KS > 1 (the narrow output layer of configs[4]: 80 tiles of 32 x 64 over K = 4096 are 64 DEPENDENT k-tile round trips on a third of the
// CUs, 17.7 us): the k range of a tile is split over KS workgroups -- launched next to each other on ONE XCD (block index mod 8), so they
// meet in one L2 -- that write their fp32 partial tiles (write-through), drain, and take a ticket from the tile's word; the LAST arriver
// sums the KS partials in the fixed order 0..KS-1 (its own included, from memory: the same bits whoever arrives last) and runs the
// epilogue.  Nobody waits for anybody.  The ticket words only grow (KS per launch; 2^32 is a multiple of KS).
template <int EPI, int BM, bool BKN = false, bool DMA = false, int KS = 1>
__global__ __launch_bounds__(BM == 32 ? 128 : 256, DMA ? 1 : 2) void bp_gemm_bf16(const BfGemmArgs g, const BfEpiArgs e)
{
    static_assert(!DMA || BM == 128, "the LDS-DMA loop is written for 128 x 64 tiles");
    static_assert(KS == 1 || (!DMA && (KS & (KS - 1)) == 0), "split k: register-staged loop, power-of-two split");
    constexpr int NTHR = BM == 32 ? 128 : 256, ROWS = BM + BF_BN, NCHK = ROWS * 8 / NTHR;     // 16-byte chunks per thread and tile
    constexpr int TMB = BM == 128 ? 2 : 1;                                                      // 32x32 blocks per wave along m
    constexpr int NCHK_A = BM * 8 / NTHR;                                                       // chunks i < NCHK_A belong to A, the rest to B
    constexpr int STAGE_H = BKN ? BM * BF_LDS + BF_BK * BF_LDB : ROWS * BF_LDS;                 // halfs per LDS stage
    constexpr int DMA_STAGE = 192 * 128, DMA_ST = 4;                                            // bytes per stage, ring length
    __shared__ __attribute__((aligned(1024))) bf16_t smem[DMA ? DMA_ST * DMA_STAGE / 2 : 2 * STAGE_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, wm = (BM >= 64) ? wave >> 1 : 0, wn = wave & 1;
    // XCD-aware tile map (block b runs on XCD b % 8): the workgroups that share a B panel (same tile_n, all tile_m)
    // sit on one XCD, so the panel is fetched into that XCD's L2 once instead of eight times
    int tile_m, tile_n;
    int tile_lin = blockIdx.x, kz = 0;
    if constexpr (KS > 1) {                                   // groups of 8 tiles x KS slices: slice z of tile 8g+j is block (8*KS)g + 8z + j (the launch has tiles % 8 == 0)
        const int grp = blockIdx.x / (8 * KS), r = blockIdx.x % (8 * KS);
        kz = r >> 3; tile_lin = grp * 8 + (r & 7);
    }
    if ((g.tiles_n & 7) == 0) {
        const int b = tile_lin, xcd = b & 7, jj = b >> 3, per = g.tiles_n >> 3;
        tile_n = xcd * per + jj / g.tiles_m; tile_m = jj % g.tiles_m;
    } else {
        tile_m = tile_lin % g.tiles_m; tile_n = tile_lin / g.tiles_m;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BF_BN;
    // chunk c of a tile: row c>>3 (rows [0, BM) from A, then BF_BN rows from B), 8 halfs at column (c&7)*8
    const bf16_t *src[NCHK]; int dst[NCHK];
#pragma unroll
    for (int i = 0; i < NCHK; ++i) {
        const int c = tid + i * NTHR, row = c >> 3, kc = (c & 7) * 8;
        if (BKN && i >= NCHK_A) {                            // B tile [k][n]: row = k-row of the tile, 8 consecutive n per chunk
            src[i] = g.B + (size_t)(row - BM) * g.ldb + n0 + kc;
            dst[i] = BM * BF_LDS + (row - BM) * BF_LDB + kc;
        } else {
            src[i] = row < BM ? g.A + (size_t)(m0 + row) * g.lda + kc : g.B + (size_t)(n0 + row - BM) * g.ldb + kc;
            dst[i] = row * BF_LDS + kc;
        }
    }
    const size_t bstep = BKN ? (size_t)g.ldb : 1;            // halfs per k of the B operand's source
    const int nt_all = g.K / BF_BK, kt0 = KS > 1 ? kz * nt_all / KS : 0;
    if constexpr (KS > 1) {
#pragma unroll
        for (int i = 0; i < NCHK; ++i) src[i] += (size_t)kt0 * BF_BK * ((BKN && i >= NCHK_A) ? bstep : (size_t)1);
    }
    // ---- epilogue mapping: lane -> column n, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32x32 block
    const int n = n0 + wn * 32 + (lane & 31);
    const int rbase = m0 + wm * 32 * TMB + 4 * (lane >> 5);   // (wm = 0 for BM = 32); block i of the wave: + 32*i
    const bool live = n < e.n_true;
    // Everything the epilogue READS (targets | y_{l-1} | W, delta) is fetched here, BEFORE the k-loop, in one burst: its
    // latency hides under the loop (wgrad has only bunch/64 k-tiles, so a workgroup is otherwise one round trip for
    // the tiles plus one for W/delta), and no load has to wait behind the epilogue's stores, which go through
    // pointers the compiler must assume may alias the inputs.
    float in0[TMB][16], in1[TMB][16];
    float bn = 0.0f;
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { in0[b][r] = 0.0f; in1[b][r] = 0.0f; }
    // dgrad: y_{l-1} of this lane's 16 (32) elements -- UNCONDITIONAL loads from a clamped row and column, in straight-line code,
    // kept as the raw 16 bits until the epilogue widens them.  (Guarded by m < m_limit and widened on the spot, the compiler emitted
    // load, s_waitcnt vmcnt(0), shift 32 times in a row: 32 serial round trips in front of the k-loop -- the 5-k-tile output-layer
    // dgrad of configs[4] took 15.6 us.)
    if constexpr (EPI == BEPI_DGRAD) {
        const int nc = n < e.n_limit ? n : e.n_limit - 1;
#pragma unroll
        for (int b = 0; b < TMB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = rbase + 32 * b + (r & 3) + 8 * (r >> 2);
                m = m < e.m_limit ? m : e.m_limit - 1;
                in0[b][r] = __uint_as_float((uint32_t)e.yprev[(size_t)m * e.ldy + nc]);
            }
    }
    if (n < e.n_limit) {
        if constexpr (EPI == BEPI_FWD_HIDDEN || EPI == BEPI_FWD_OUT) bn = e.bias[n];
#pragma unroll
        for (int b = 0; b < TMB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + 32 * b + (r & 3) + 8 * (r >> 2);
                if (m < e.m_limit) {
                    if constexpr (EPI == BEPI_FWD_OUT) { if (e.C && live) in0[b][r] = e.targ[(size_t)m * e.ldt + n]; }
                    if constexpr (EPI == BEPI_WGRAD_UPDATE) { in0[b][r] = e.W[(size_t)m * e.ldw + n]; in1[b][r] = e.D[(size_t)m * e.ldw + n]; }
                }
            }
    }
    f32x16 accs[TMB];
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[b][r] = 0.0f;
    const int nt = KS > 1 ? (kz + 1) * nt_all / KS - kt0 : nt_all;
    if constexpr (DMA) {
        typedef __attribute__((address_space(3))) void *lds_ptr;
        typedef const __attribute__((address_space(1))) void *glb_ptr;
        char *sm = reinterpret_cast<char *>(smem);
        // this wave's six pieces of a tile: rows 8*(6*wave+i) .. +7 of the 192-row image; lane -> row + (lane>>3), slot lane&7
        const bf16_t *psrc[6]; size_t pstep[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = 8 * (wave * 6 + i) + (lane >> 3);
            if (r < BM || !BKN) {
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                psrc[i] = (r < BM ? g.A + (size_t)(m0 + r) * g.lda : g.B + (size_t)(n0 + r - BM) * g.ldb) + c * 8; pstep[i] = BF_BK;
            } else {
                const int k = r - BM, c = (lane & 7) ^ (4 * ((k >> 1) & 1));
                psrc[i] = g.B + (size_t)k * g.ldb + n0 + c * 8; pstep[i] = (size_t)BF_BK * g.ldb;
            }
        }
        // the workgroups that share a B panel (the m-tiles of one n-tile, same XCD) walk k out of phase, BF_ROT tiles apart: each
        // line of the panel is then pulled from HBM by ONE of them and found in L2 by the others, instead of four requests for a line
        // that is still on its way (cold weights: 30.3 -> 26.5 us in tools/bf16_gemm_probe.hip)
        const int rot = (tile_m * g.k_rot) % nt;
#define BF_PIECE(i, t, st)                                                                                          \
        do { int tt_ = ((t) < nt ? (t) : nt - 1) + rot;     /* past the end: a duplicate into a stage nobody reads */   \
             if (tt_ >= nt) tt_ -= nt;                                                                              \
             __builtin_amdgcn_global_load_lds((glb_ptr)(psrc[i] + (size_t)tt_ * pstep[i]),                          \
                                              (lds_ptr)(sm + (st) * DMA_STAGE + (wave * 6 + (i)) * 1024), 16, 0, 0); } while (0)
        // fragment addresses (bytes in LDS, stage 0): row r, chunk 2q+h -> r*128 + 16*(h ^ (s&1)) + 32*(q ^ (s>>1)), s = (r>>1)&7
        const int ra = wm * 64 + (lane & 31), rb = BM + wn * 32 + (lane & 31), hh = lane >> 5;
        const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)sm;
        const int jj4 = (lane & 15) >> 2, cb = 4 * wn + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
        unsigned aq[4], bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            aq[q] = lds0 + ra * 128 + 16 * (hh ^ (sa & 1)) + 32 * (q ^ (sa >> 1));
            bq[q] = BKN ? lds0 + BM * 128 + (16 * q + 8 * hh + jj4) * 128 + 16 * (cb ^ (4 * ((jj4 >> 1) & 1))) + 8 * (lane & 1)
                        : lds0 + rb * 128 + 16 * (hh ^ (sb & 1)) + 32 * (q ^ (sb >> 1));
        }
        bf_f32x4 fa0[4], fa1[4], fb[4]; bf_v4s flo[4], fhi[4];
        constexpr int NRD = BKN ? 4 : 3;                    // LDS reads per k-step
#define BF_READS(so, q)                                                                                             \
        do { asm volatile("ds_read_b128 %0, %1" : "=v"(fa0[q]) : "v"(aq[q] + (so)));                                \
             if constexpr (BKN) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(flo[q]) : "v"(bq[q] + (so)));    \
                                  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(fhi[q]) : "v"(bq[q] + (so))); } \
             else asm volatile("ds_read_b128 %0, %1" : "=v"(fb[q]) : "v"(bq[q] + (so)));                            \
             asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa1[q]) : "v"(aq[q] + (so))); } while (0)
#define BF_ARRIVED(q, N) do { if constexpr (BKN) bf_lgkm4<(N)>(fa0[q], fa1[q], flo[q], fhi[q]); else bf_lgkm3<(N)>(fa0[q], fa1[q], fb[q]); } while (0)
#pragma unroll
        for (int t = 0; t < DMA_ST - 1; ++t)
#pragma unroll
            for (int i = 0; i < 6; ++i) BF_PIECE(i, t, t);
        for (int t = 0; t < nt; ++t) {
            const unsigned so = (unsigned)((t % DMA_ST) * DMA_STAGE);
            const int stn = (t + DMA_ST - 1) % DMA_ST;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DMA_ST - 2) * 6) : "memory");       // tile t has landed (the DMA_ST-2 tiles behind it may be in flight)
            __builtin_amdgcn_s_barrier();                            // ... for every wave; and the stage of tile t-1 is free for tile t+DMA_ST-1
            __builtin_amdgcn_sched_barrier(0);
            BF_READS(so, 0); BF_READS(so, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) BF_ARRIVED(q, NRD); else BF_ARRIVED(q, 0);          // behind step q: the reads of step q+1
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t bf_;
                if constexpr (BKN) bf_ = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(flo[q], fhi[q], 0, 1, 2, 3, 4, 5, 6, 7));
                else bf_ = __builtin_bit_cast(bf16x8_t, fb[q]);
                accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa0[q]), bf_, accs[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < 2) BF_PIECE(2 * q, t + DMA_ST - 1, stn); else BF_PIECE(2 + q, t + DMA_ST - 1, stn);
                __builtin_amdgcn_sched_barrier(0);
                accs[TMB - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa1[q]), bf_, accs[TMB - 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < 2) BF_PIECE(2 * q + 1, t + DMA_ST - 1, stn);
                if (q + 2 < 4) BF_READS(so, q + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped duplicates of the last three iterations
#undef BF_PIECE
#undef BF_READS
#undef BF_ARRIVED
    } else {
    // three register images of a tile, always addressed by name (a runtime-indexed array would live in scratch)
    struct Img { uint4 v[NCHK]; };
    Img r0, r1, r2;
#define BF_LOAD(R, t)                                                                                   \
    do {                                                                                                \
        const int kk_ = ((t) < nt ? (t) : nt - 1) * BF_BK;      /* unconditional, clamped */             \
        _Pragma("unroll") for (int i = 0; i < NCHK; ++i)                                                \
            R.v[i] = *reinterpret_cast<const uint4 *>(src[i] + ((BKN && i >= NCHK_A) ? (size_t)kk_ * bstep : (size_t)kk_)); \
    } while (0)
#define BF_STORE(R, st)                                                                                 \
    do {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NCHK; ++i) { const uint4 v_ = R.v[i];                     \
            *reinterpret_cast<uint4 *>(smem + (st) * STAGE_H + dst[i]) = v_; }                          \
    } while (0)
    const int arow = wm * 32 * TMB + (lane & 31), brow = BM + wn * 32 + (lane & 31), kh = (lane >> 5) * 8;
    // BKN: this lane's piece of the transpose read (16 lanes fetch 4 k-rows x 16 n: lane i supplies row i>>2, 4 n at 4*(i&3),
    // and receives column i of that 4x16 block = 4 consecutive k of n = 16*((lane>>4)&1) + i; lanes 32..63 take k + 8)
    const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * BF_LDB + wn * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
// multiply tile t (LDS stage t&1); RN holds tile t+1: move it to the other stage and refill RN with tile t+1+NPF
#define BF_ITER(t, RN)                                                                                  \
    do {                                                                                                \
        const bf16_t *base_ = smem + ((t) & 1) * STAGE_H;                                               \
        const bf16_t *ap_ = base_ + arow * BF_LDS + kh, *bp_ = base_ + brow * BF_LDS + kh;              \
        bf16x8_t a_[TMB][4], b_[4];                                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                              \
            if constexpr (BKN) {                                                                        \
                typedef __attribute__((address_space(3))) bf_v4s *lds4_;                                \
                const bf16_t *tp_ = base_ + BM * BF_LDS + tr_off + 16 * q_ * BF_LDB;                    \
                const bf_v4s lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_)(tp_));               \
                const bf_v4s hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_)(tp_ + 4 * BF_LDB));  \
                b_[q_] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7)); \
            } else                                                                                      \
            b_[q_] = *reinterpret_cast<const bf16x8_t *>(bp_ + 16 * q_);                                \
            _Pragma("unroll") for (int i_ = 0; i_ < TMB; ++i_) a_[i_][q_] = *reinterpret_cast<const bf16x8_t *>(ap_ + i_ * 32 * BF_LDS + 16 * q_); \
        }                                                                                               \
        BF_STORE(RN, ((t) + 1) & 1);               /* (past the last tile: a clamped duplicate nobody reads) */ \
        BF_LOAD(RN, (t) + 1 + BF_NPF);                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < TMB; ++i_)                                          \
                accs[i_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[i_][q_], b_[q_], accs[i_], 0, 0, 0); \
        __syncthreads();                                                                                \
    } while (0)
    // tile t waits in image r(t % 3) until it is moved to LDS stage t & 1
    BF_LOAD(r0, 0); BF_LOAD(r1, 1); BF_LOAD(r2, 2);
    BF_STORE(r0, 0);
    BF_LOAD(r0, BF_NPF);
    __syncthreads();
    int t = 0;
    for (; t + 3 <= nt; t += 3) { BF_ITER(t, r1); BF_ITER(t + 1, r2); BF_ITER(t + 2, r0); }
    if (t < nt) { BF_ITER(t, r1); if (t + 1 < nt) BF_ITER(t + 1, r2); }
#undef BF_LOAD
#undef BF_STORE
#undef BF_ITER

    }

    // ---- epilogue
    // DMA form: the two bf16 copies of the tile leave through LDS as whole 128-byte lines (8 x 16-byte stores per thread instead of
    // 32 two-byte and 16 eight-byte ones straight from the accumulator layout: the store ISSUE was a third of the kernel).  The ring
    // is free: every wave is past its last fragment read once the barrier below is passed, and no DMA is in flight.
    constexpr int OC_LD = 72, OT_LD = 136;                    // halfs per row of the staged C [128][64] and CT [64][128] tiles
    bf16_t *oc = smem, *ot = smem + 128 * OC_LD;
    if constexpr (KS > 1) {
        constexpr int PART = NTHR * 16 * TMB;                 // floats per partial tile, [reg][thread]
        float *slab = g.ks_slab + (size_t)tile_lin * KS * PART, *mine = slab + kz * PART;
#pragma unroll
        for (int b = 0; b < TMB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + (b * 16 + r) * NTHR + tid, accs[b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *tk = reinterpret_cast<unsigned *>(smem);    // (the stages are free: the loop ended on a barrier)
        if (tid == 0) *tk = __hip_atomic_fetch_add(g.ks_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if ((*tk & (KS - 1)) != KS - 1) return;               // not the last of this launch's KS arrivals
        constexpr int ZG = KS < 4 ? KS : 4;                    // partials in flight per pass (4 x 16 x TMB registers)
#pragma unroll
        for (int z0 = 0; z0 < KS; z0 += ZG) {
            float pz[ZG][TMB][16];
#pragma unroll
            for (int z = 0; z < ZG; ++z)
#pragma unroll
                for (int b = 0; b < TMB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pz[z][b][r] = __hip_atomic_load(slab + (z0 + z) * PART + (b * 16 + r) * NTHR + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int b = 0; b < TMB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = z0 == 0 ? pz[0][b][r] : accs[b][r] + pz[0][b][r];
#pragma unroll
                    for (int z = 1; z < ZG; ++z) s += pz[z][b][r];
                    accs[b][r] = s;
                }
        }
    }
    if constexpr (DMA) __syncthreads();
    else if (n >= e.n_limit) return;
#pragma unroll
    for (int blk = 0; blk < TMB; ++blk) {
    const f32x16 &acc = accs[blk];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mq = rbase + 32 * blk + 8 * q;      // 4 consecutive rows mq..mq+3 (mq % 4 == 0)
        float v[4];
        if constexpr (EPI == BEPI_FWD_HIDDEN) {
            uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            // same counter layout as the fp32 path (bp_kernels.h)
            if (e.drop_thresh) drop_words4(w, mq, n, e.frame_off, (uint32_t)e.n_true, e.layer, e.step, e.seed_lo, e.seed_hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = act_fwd(e.act, e.alpha * acc[4 * q + j] + bn);
                if (!live || w[j] < e.drop_thresh || mq + j >= e.m_limit) y = 0.0f;
                v[j] = y;
            }
        } else if constexpr (EPI == BEPI_FWD_OUT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                float d = 0.0f;
                if (m < e.m_limit) {
                    const float o = live ? e.alpha * acc[4 * q + j] + bn : 0.0f;
                    if (e.out) e.out[(size_t)m * e.ldo + n] = o;
                    if (e.C && live) d = e.scale * (o - in0[blk][4 * q + j]);                     // kernSubClean
                }
                v[j] = d;
            }
            if (!e.C) continue;
        } else if constexpr (EPI == BEPI_DGRAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = (mq + j < e.m_limit && live) ? act_bwd(e.act, __uint_as_float(__float_as_uint(in0[blk][4 * q + j]) << 16)) * acc[4 * q + j] : 0.0f;
        } else {                                      // wgrad: rows = units of layer l-1, cols = units of layer l
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                if (m >= e.m_limit) { v[j] = 0.0f; continue; }
                const size_t i = (size_t)m * e.ldw + n;
                if constexpr (EPI == BEPI_WGRAD_UPDATE) {
                    const float w = in0[blk][4 * q + j];
                    const float d = e.mom * in1[blk][4 * q + j] - e.c1 * (acc[4 * q + j] / e.ndiv + e.wc * w);   // kernUpdatedelta
                    e.D[i] = d;
                    v[j] = d + 1.0f * w;                                                                    // kernAccSum
                    e.W[i] = v[j];
                } else {
                    e.W[i] = acc[4 * q + j];          // gradient into the flat buffer (data parallel)
                    v[j] = 0.0f;
                }
            }
            if constexpr (EPI == BEPI_WGRAD_STORE) continue;
        }
        // bf16 copies in both orientations (for wgrad: the refreshed shadow weights)
        bf16_t hb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[j] = f2bf(v[j]);
        const uint2 pk = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
        if constexpr (DMA) {
            const int ml = mq - m0, nl = n - n0;
#pragma unroll
            for (int j = 0; j < 4; ++j) oc[(ml + j) * OC_LD + nl] = hb[j];
            *reinterpret_cast<uint2 *>(ot + nl * OT_LD + ml) = pk;
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mq + j < e.m_limit) e.C[(size_t)(mq + j) * e.ldc + n] = hb[j];
        if (e.CT) *reinterpret_cast<uint2 *>(e.CT + (size_t)n * e.ldct + mq) = pk;     // (the weights keep ONE shadow: no transposed copy)
        }
    }
    }
    if constexpr (DMA) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // C: 128 rows x 8 chunks of 16 bytes
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            if (m0 + row < e.m_limit)
                *reinterpret_cast<uint4 *>(e.C + (size_t)(m0 + row) * e.ldc + n0 + ch * 8) = *reinterpret_cast<const uint4 *>(oc + row * OC_LD + ch * 8);
        }
        if (e.CT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                      // CT: 64 rows x 16 chunks
                const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
                *reinterpret_cast<uint4 *>(e.CT + (size_t)(n0 + row) * e.ldct + m0 + ch * 8) = *reinterpret_cast<const uint4 *>(ot + row * OT_LD + ch * 8);
            }
        }
    }
}

// fp32 rows -> bf16 in both orientations: out[r][c] and outT[c][r] (rows x cols, padded leading dims; pads written 0).
// Used for the (masked) input bunch and for the shadow weights (at creation and after a data-parallel update).
__global__ void bp_to_bf16_both(const float *src, int lds, int rows, int cols, bf16_t *out, int ldo, bf16_t *outT, int ldt,
                                int rows_pad, int cols_pad)
{
    __shared__ bf16_t tile[32][33];
    const int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j;
        bf16_t h = 0;
        if (r < rows && c < cols) h = f2bf(src[(size_t)r * lds + c]);
        if (r < rows_pad && c < cols_pad) out[(size_t)r * ldo + c] = h;
        tile[j][threadIdx.x] = h;
    }
    if (!outT) return;                               // (uniform: the weights' single shadow)
    __syncthreads();
    const int rt = r0 + threadIdx.x;                 // transposed: thread x walks rows of the source
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int ct = blockIdx.x * 32 + j;
        if (ct < cols_pad && rt < rows_pad) outT[(size_t)ct * ldt + rt] = tile[threadIdx.x][j];
    }
}

// bias gradient = column sums of dEdX_l (kernAccSumrow) from the bf16 copy, fp32 accumulation; fused update of the
// bias (kernUpdatedelta with wc = 0 + kernAccSum) or store into the gradient buffer.
__global__ void bp_bias_bf16(const bf16_t *dx, int ld, int rows, int n_true, float *bias, float *dbias, float *gout,
                             float mom, float c1, float ndiv)
{
    // block = 64 columns x 16 row groups (the kernel is pure load latency: many rows in flight); fixed summation
    // order (row groups folded 0..15) => deterministic
    __shared__ float part[16][64];
    const int n = blockIdx.x * 64 + threadIdx.x;
    float s = 0.0f;
    if (n < n_true) {
        int f = threadIdx.y;
        for (; f + 112 < rows; f += 128) {            // 8 independent loads in flight, summed in a fixed order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = bf2f(dx[(size_t)(f + 16 * u) * ld + n]);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; f < rows; f += 16) s += bf2f(dx[(size_t)f * ld + n]);
    }
    part[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y != 0 || n >= n_true) return;
    s = 0.0f;
#pragma unroll
    for (int y = 0; y < 16; ++y) s += part[y][threadIdx.x];
    if (gout) { gout[n] = s; return; }
    const float d = mom * dbias[n] - c1 * (s / ndiv + 0.0f * bias[n]);
    dbias[n] = d;
    bias[n] = d + 1.0f * bias[n];
}
