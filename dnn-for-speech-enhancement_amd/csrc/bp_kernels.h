// bp_kernels.h -- CDNA4 (gfx950) kernels of the frame-wise DNN step.
//
// One LDS-staged fp32 MFMA GEMM template (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered
// fmaf chain) with the step's elementwise work fused into its epilogues, replacing the
// reference's cuBLAS Sgemm wrappers (DevFunc.h:29-67) + 10 small kernels (DevFunc.cu).
//
//   fwd   X = Y_prev . W  (+bias, act, dropout of the OUTPUT)   A=[m][k]  B=[k][n]
//   dgrad dEdX_prev = act'(y_prev) * (dEdX . W^T)               A=[m][k]  B=[n][k]
//   wgrad G = Y_prev^T . dEdX  (+ momentum update of W, b)      A=[k][m]  B=[k][n]
//
// Operand tiles live in LDS k-major ([k][m] / [k][n]) so that the MFMA operand fetch
// (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) is a conflict-free ds_read_b32 of 32
// consecutive dwords per half-wave.  k-contiguous global operands are transposed on the way
// in (float4 global load -> 4 x ds_write_b32, odd row stride => conflict-free); m/n-contiguous
// operands go in with ds_write_b128.  Register-staged software pipeline: the global loads of k-tile
// t+2 are in flight while tile t is multiplied and tile t+1 moves into the other LDS stage; one barrier per k-tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(std::forward<F>(f));
    }
}

enum { EPI_FWD_HIDDEN = 0, EPI_FWD_OUT = 1, EPI_DGRAD = 2, EPI_WGRAD_UPDATE = 3, EPI_WGRAD_STORE = 4,
       EPI_PARTIAL = 5 /* raw k-slice partial sums into slab blockIdx.y (split-K) */,
       EPI_OUT_SPLIT = 6 /* split-K output layer in ONE launch: k-slice partials into the slabs, the tile's last arriver sums them and runs EPI_FWD_OUT */ };
static constexpr int OUT_SPLITS = 4;                       // k-slices of the narrow output layer (EPI_OUT_SPLIT)

// The k-loop loads carry NO predicates (a predicated load makes hipcc drain vmcnt at the top of
// every iteration, which serialises the prefetch).  Contract with the caller instead:
//   * every operand buffer is readable for whole tiles (rows rounded up to the tile, one extra
//     row of slack), so tile reads never leave the allocation;
//   * rows/columns past the true extents either hold zeros (layer-width padding; dEdX rows past
//     the bunch, which zero the k-tail of wgrad) or only feed accumulator rows that the
//     epilogue never stores (frames past the bunch in fwd/dgrad).
struct GemmArgs {
    const float *A, *B;
    int lda, ldb;            // leading dimensions (floats)
    int K;                   // reduction extent actually looped (rounded up to BK inside)
    int tiles_m, tiles_n;
#ifdef BP_TRACE              // development only: per-workgroup phase timestamps (tools/gemm_trace.hip)
    unsigned long long *trace;
#endif
    int k_split;             // split-K: workgroup row blockIdx.y handles k in [y*k_split, y*k_split + K)
    size_t slab_stride;      // split-K: floats between the partial-sum slabs of consecutive k-slices
    float *ks_slab; unsigned *ks_ticket;   // EPI_OUT_SPLIT: the slabs ([slice][m][n], rows ldc apart like the output) and one ticket word per tile
};

struct EpiArgs {
    float *C; int ldc;               // Y | dEdX_L | dEdX_prev | W | G
    int m_limit, n_limit;            // rows / cols of C that exist (padded extents)
    int n_true;                      // unpadded column count (pad columns are forced to 0)
    const float *bias;               // fwd
    float alpha;                     // fwd: x = alpha*acc + bias (alpha = keep in CV, BP_GPU.cu:726-746)
    int act;                         // 0 ReLU, 1 Sigmoid
    const float *aux; int ldaux;     // fwd_out: targ | dgrad: y_prev
    float *aux2; int ldaux2;         // fwd_out: out (may be null) | wgrad_update: delta_W
    float scale;                     // fwd_out: 2/n_frames (DevFunc.cu:263)
    // wgrad update (DevFunc.cu:313-318 + 270-277)
    float mom, c1, wc, ndiv;         // c1 = (1-m)*lr or lr ; ndiv = (float)n
    float *bias_w, *bias_d, *bias_g; // bias / delta_bias (update) or bias-gradient (store)
    // dropout of the produced activation (BP_GPU.cu:546-549 applied by the producer)
    uint32_t drop_thresh, seed_lo, seed_hi, step, layer;
    int frame_off;                   // global frame index of row 0 of this bunch
    const uint8_t *mask; int ldmask; // injected dropout mask of the produced activation ([row][unit] bytes, 1 = drop;
                                     // bp_train_resident_masked, parity tests only) -- replaces the Philox draw
    unsigned *done;                  // wgrad store (data parallel): +1 per finished tile, for the exchange stream (bp_dp.h); may be null
};

// ------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                              uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Dropout words of the 4 consecutive bunch rows r0..r0+3 (r0 % 4 == 0) of unit n: word (gf & 3) of the Philox block
// keyed by (gf >> 2, unit) with gf = global frame index = row + frame_off.  frame_off % 4 == 0 (the usual case) needs
// one block; otherwise the four rows straddle two (frame_off is a launch constant, so the branch is uniform).
__device__ __forceinline__ void drop_words4(uint32_t (&w)[4], int r0, int n, int frame_off, uint32_t n_true, uint32_t layer,
                                            uint32_t step, uint32_t seed_lo, uint32_t seed_hi)
{
    const uint64_t g0 = (uint64_t)(uint32_t)(r0 + frame_off);
    const uint64_t idx = (g0 >> 2) * (uint64_t)n_true + (uint32_t)n;
    uint32_t a[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), layer, step};
    philox4x32_10(a[0], a[1], a[2], a[3], seed_lo, seed_hi);
    const int sh = frame_off & 3;
    if (sh == 0) { w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; return; }
    const uint64_t idx2 = idx + (uint64_t)n_true;
    uint32_t b[4] = {(uint32_t)idx2, (uint32_t)(idx2 >> 32), layer, step};
    philox4x32_10(b[0], b[1], b[2], b[3], seed_lo, seed_hi);
    if (sh == 1) { w[0] = a[1]; w[1] = a[2]; w[2] = a[3]; w[3] = b[0]; }
    else if (sh == 2) { w[0] = a[2]; w[1] = a[3]; w[2] = b[0]; w[3] = b[1]; }
    else { w[0] = a[3]; w[1] = b[0]; w[2] = b[1]; w[3] = b[2]; }
}

__device__ __forceinline__ float act_fwd(int act, float x)
{
    // DevFunc.cu:67-79 (ReLU, strict > 0) | DevFunc.cu:47-54 (.bak: 1/(1+expf(-x)))
    return act == 0 ? (x > 0.0f ? x : 0.0f) : 1.0f / (1.0f + expf(-x));
}
__device__ __forceinline__ float act_bwd(int act, float y)
{
    // DevFunc.cu:81-97 (y>0 ? 1 : 0) | :56-64 (.bak: (1-y)*y), from the post-dropout output y
    return act == 0 ? (y > 0.0f ? 1.0f : 0.0f) : (1.0f - y) * y;
}

// ------------------------------------------------------------------ epilogue of one 32x32 block
// C/D layout of v_mfma_f32_32x32x2_f32: lane l, reg r -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
// Inputs the epilogue needs from memory, fetched BEFORE the k-loop so their latency hides under
// it: bias (fwd), targ (fwd_out) / y_prev (dgrad) / W (wgrad) in p0, delta_W (wgrad) in p1.
struct EpiPre { float bias; f32x16 p0, p1; };

// A pointer the compiler cannot prove wave-uniform, forced into SGPRs (so that loads/stores through it
// take the  saddr + 32-bit lane offset  form and need no per-row 64-bit address VGPRs).
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    typedef __attribute__((address_space(1))) T *GlobalPtr;      // keep it a GLOBAL pointer (not flat)
    return (T *)(GlobalPtr)(((unsigned long long)hi << 32) | lo);
}

// Pin a wave-uniform pointer in an SGPR pair and hide its derivation from the optimiser, so that
// `*(T*)((char*)sgpr_row_base(p) + lane_byte_offset)` is selected as  global_load v, v_off, s[base]
// instead of being reassociated into per-lane 64-bit addresses.
template <class T>
__device__ __forceinline__ T *sgpr_row_base(T *p)
{
    typedef __attribute__((address_space(1))) T *GlobalPtr;
    GlobalPtr q = (GlobalPtr)p;
    asm volatile("" : "+s"(q));
    return (T *)q;
}

// Registers [R0, R0+RN) of one 32x32 accumulator block (after an in-workgroup k-split every wave
// finishes 16/KS of the block's registers).
template <int EPI, int R0, int RN>
__device__ __forceinline__ void epilogue_fetch(const EpiArgs &e, int mb, int nb, int lane, EpiPre &p)
{
    if constexpr (EPI == EPI_WGRAD_UPDATE) {
        // natural MFMA layout (lane -> column n, registers -> rows): every W / delta load and store is two full 128-byte rows
        // uniform row base (SGPRs) + one per-lane 32-bit offset: no per-row address VGPRs.  m_limit is a
        // multiple of 32 (padded widths), so a 32-row block is wholly inside or wholly outside the matrix;
        // outside blocks read row 0 and are never stored.
        const int mbc = mb < e.m_limit ? mb : 0;
        const float *cw = uniform_ptr(e.C + (size_t)mbc * e.ldc + nb);
        const float *cd = uniform_ptr(e.aux2 + (size_t)mbc * e.ldc + nb);
        const unsigned ldc = __builtin_amdgcn_readfirstlane(e.ldc);
        const unsigned lob = 4u * ((unsigned)(4 * (lane >> 5)) * (unsigned)e.ldc + (unsigned)(lane & 31));   // bytes
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * ldc;
            p.p0[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sgpr_row_base(cw + ro)) + lob);
            p.p1[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sgpr_row_base(cd + ro)) + lob);
        }
        return;
    }
    const int n = nb + (lane & 31);
    const int rbase = mb + 4 * (lane >> 5);
    static_assert(EPI != EPI_OUT_SPLIT, "fetched as EPI_FWD_OUT");
    if constexpr (EPI == EPI_FWD_HIDDEN || EPI == EPI_FWD_OUT) p.bias = e.bias[n];
    if constexpr (EPI == EPI_FWD_OUT || EPI == EPI_DGRAD) {
        if (EPI == EPI_FWD_OUT && !e.C) return;
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            const int mc = m < e.m_limit ? m : e.m_limit - 1;      // (rows past the matrix are never stored)
            p.p0[r] = e.aux[(size_t)mc * e.ldaux + n];
        }
    }
}

template <int EPI, int R0, int RN>
__device__ __forceinline__ void epilogue_block(const EpiArgs &e, int mb, int nb, const f32x16 &acc, int lane,
                                               const EpiPre &p)
{
    static_assert(RN % 4 == 0 && R0 % 4 == 0, "register range must cover whole 4-row groups");
    if constexpr (EPI == EPI_WGRAD_UPDATE || EPI == EPI_WGRAD_STORE) {
        const int n = nb + (lane & 31);
        if (n >= e.n_limit || mb >= e.m_limit) return;              // (m_limit % 32 == 0: whole block in or out)
        float *cw = uniform_ptr(e.C + (size_t)mb * e.ldc + nb);
        const unsigned ldc = __builtin_amdgcn_readfirstlane(e.ldc);
        const unsigned lob = 4u * ((unsigned)(4 * (lane >> 5)) * (unsigned)e.ldc + (unsigned)(lane & 31));   // bytes
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * ldc;
            if constexpr (EPI == EPI_WGRAD_UPDATE) {
                float *cd = uniform_ptr(e.aux2 + (size_t)mb * e.ldc + nb);
                const float w = p.p0[r];
                const float d = e.mom * p.p1[r] - e.c1 * (acc[r] / e.ndiv + e.wc * w);   // kernUpdatedelta
                *reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cd + ro)) + lob) = d;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cw + ro)) + lob) = d + 1.0f * w;   // kernAccSum
            } else {
                // gradient tile of the data-parallel step: SYSTEM-scope write-through stores (sc0 sc1) -- once the wave has drained
                // vmcnt the tile is in memory whatever kind of allocation the gradient buffer is, which is what lets the tile count
                // of bp_wgrad_dma.h hand the segment to the peers without a kernel boundary or an L2 write-back (bp_dp.h)
                __hip_atomic_store(reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cw + ro)) + lob), acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    const int n = nb + (lane & 31);
    const int rbase = mb + 4 * (lane >> 5);
    if (n >= e.n_limit) return;
    if constexpr (EPI == EPI_FWD_HIDDEN) {
        const float bn = p.bias;
        const bool live = n < e.n_true;
#pragma unroll
        for (int q = R0 / 4; q < (R0 + RN) / 4; ++q) {
            uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            const int r0 = rbase + 8 * q;
            if (e.mask) {                                        // injected mask: word 0 = "drop", thresh 1 (set by the host)
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (r0 + j < e.m_limit && live && e.mask[(size_t)(r0 + j) * e.ldmask + n]) ? 0u : 0xFFFFFFFFu;
            } else if (e.drop_thresh) drop_words4(w, r0, n, e.frame_off, (uint32_t)e.n_true, e.layer, e.step, e.seed_lo, e.seed_hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = r0 + j;
                float y = act_fwd(e.act, e.alpha * acc[q * 4 + j] + bn);
                if (!live || w[j] < e.drop_thresh) y = 0.0f;
                if (m < e.m_limit) e.C[(size_t)m * e.ldc + n] = y;
            }
        }
    } else if constexpr (EPI == EPI_FWD_OUT) {
        const float bn = p.bias;
        const bool live = n < e.n_true;
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) {
                const float o = live ? e.alpha * acc[r] + bn : 0.0f;
                if (e.aux2) e.aux2[(size_t)m * e.ldaux2 + n] = o;
                if (e.C) e.C[(size_t)m * e.ldc + n] = live ? e.scale * (o - p.p0[r]) : 0.0f;   // kernSubClean
            }
        }
    } else if constexpr (EPI == EPI_DGRAD) {
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) e.C[(size_t)m * e.ldc + n] = act_bwd(e.act, p.p0[r]) * acc[r];   // kernDsigmoid*kernVecMul
        }
    } else {  // EPI_PARTIAL: plain store
#pragma unroll
        for (int r = R0; r < R0 + RN; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) e.C[(size_t)m * e.ldc + n] = acc[r];
        }
    }
}

// EPI_OUT_SPLIT, registers [R0, R0+4) of the wave's 32x32 block.  store: this k-slice's partial sums into its slab, agent-scope
// write-through (complete in memory once the wave has drained vmcnt).  sum: the OUT_SPLITS partials of the tile, added in slice order
// 0..3 whichever slice arrived last (the summation order of the two-launch form this replaces: the same bits), agent-scope loads.
template <int R0>
__device__ __forceinline__ void out_split_store(const EpiArgs &e, float *slab, int mb, int nb, const f32x16 &acc, int lane)
{
    const int n = nb + (lane & 31), rbase = mb + 4 * (lane >> 5);
    if (n >= e.n_limit) return;
#pragma unroll
    for (int r = R0; r < R0 + 4; ++r) {
        const int m = rbase + (r & 3) + 8 * (r >> 2);
        if (m < e.m_limit) __hip_atomic_store(slab + (size_t)m * e.ldc + n, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int R0>
__device__ __forceinline__ void out_split_sum(const EpiArgs &e, const float *slab0, size_t stride, int mb, int nb, f32x16 &acc, int lane)
{
    const int n = nb + (lane & 31), rbase = mb + 4 * (lane >> 5);
    if (n >= e.n_limit) return;
    float p[OUT_SPLITS][4];
#pragma unroll
    for (int z = 0; z < OUT_SPLITS; ++z)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = R0 + rr, m = rbase + (r & 3) + 8 * (r >> 2), mc = m < e.m_limit ? m : e.m_limit - 1;   // (clamped: all 16 loads unconditional)
            p[z][rr] = __hip_atomic_load(slab0 + z * stride + (size_t)mc * e.ldc + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        float s = p[0][rr];
#pragma unroll
        for (int z = 1; z < OUT_SPLITS; ++z) s += p[z][rr];
        acc[R0 + rr] = s;
    }
}

// In-workgroup k-split (KS wave groups hold partial sums of the same 32x32 block): wave KSI keeps
// registers [KSI*16/KS, (KSI+1)*16/KS), hands the others to their owners through LDS, and finishes
// its share -- so every wave runs 1/KS of the epilogue instead of wave group 0 running all of it.
template <int KS, int KSI>
__device__ __forceinline__ void ksplit_give(const f32x16 &acc, float *red, int wq, int lane)
{
    constexpr int RN = 16 / KS;
#pragma unroll
    for (int o = 0; o < KS; ++o) {
        if (o == KSI) continue;
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) red[(((wq * KS + o) * KS + KSI) * RN + rr) * 64 + lane] = acc[o * RN + rr];
    }
}
template <int KS, int KSI>
__device__ __forceinline__ void ksplit_take(f32x16 &acc, const float *red, int wq, int lane)
{
    constexpr int RN = 16 / KS;
#pragma unroll
    for (int src = 0; src < KS; ++src) {
        if (src == KSI) continue;
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) acc[KSI * RN + rr] += red[(((wq * KS + KSI) * KS + src) * RN + rr) * 64 + lane];
    }
}

// ------------------------------------------------------------------ the GEMM
// Register image of one k-tile of both operands (global -> registers -> LDS staging).
template <int NVA, int NVB>
struct TileRegs { float4 a[NVA]; float4 b[NVB]; };

// BM x BN x BK workgroup tile, 4 waves arranged WM x WN x KS (KS = 4/(WM*WN) splits each
// k-tile between wave groups; partial sums meet in LDS before the epilogue).
// A_KC: A is [m][k] in memory (k contiguous) else [k][m]; B_KC: B is [n][k] else [k][n].
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI>
struct GemmCfg {
    static constexpr int KS = 4 / (WM * WN);
    static constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static constexpr int LDA_S = A_KC ? BM + 1 : BM;
    static constexpr int LDB_S = B_KC ? BN + 1 : BN;
    static constexpr int A_STAGE = (BK * LDA_S + 3) & ~3, B_STAGE = (BK * LDB_S + 3) & ~3;
    static constexpr int NVA = BM * BK / 4 / 256, NVB = BN * BK / 4 / 256;
    static constexpr int LPS = (BK / 4 < 8) ? BK / 4 : 8;     // lanes per 128-byte row segment
    static constexpr bool BIASG = (EPI == EPI_WGRAD_UPDATE || EPI == EPI_WGRAD_STORE);
    static_assert(WM * WN * KS == 4 && TM >= 1 && TN >= 1, "wave layout");
    static_assert(BK % (2 * KS) == 0 && BK % 4 == 0, "BK");
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 threads");
    using Regs = TileRegs<NVA, NVB>;

    // Addressing: uniform tile base (SGPR) + per-thread 32-bit offset that never changes, so a
    // load is one instruction and the k-advance is scalar arithmetic.
    struct Offs { int a[NVA]; int b[NVB]; };
    static __device__ __forceinline__ void make_offs(Offs &o, const GemmArgs &g, int tid)
    {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int f = tid + i * 256;
            if constexpr (A_KC) {
                const int l8 = f % LPS, seg = f / LPS, row = seg % BM, k4 = (seg / BM) * LPS + l8;
                o.a[i] = row * g.lda + k4 * 4;
            } else {
                const int c4 = f % (BM / 4), k = f / (BM / 4);
                o.a[i] = k * g.lda + c4 * 4;
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int f = tid + i * 256;
            if constexpr (B_KC) {
                const int l8 = f % LPS, seg = f / LPS, row = seg % BN, k4 = (seg / BN) * LPS + l8;
                o.b[i] = row * g.ldb + k4 * 4;
            } else {
                const int c4 = f % (BN / 4), k = f / (BN / 4);
                o.b[i] = k * g.ldb + c4 * 4;
            }
        }
    }
    // uniform base of k-tile k0 for each operand
    static __device__ __forceinline__ const float *base_a(const GemmArgs &g, int m0, int k0)
    {
        return A_KC ? g.A + (size_t)m0 * g.lda + k0 : g.A + (size_t)k0 * g.lda + m0;
    }
    static __device__ __forceinline__ const float *base_b(const GemmArgs &g, int n0, int k0)
    {
        return B_KC ? g.B + (size_t)n0 * g.ldb + k0 : g.B + (size_t)k0 * g.ldb + n0;
    }
    static __device__ __forceinline__ void load_a(Regs &r, int i, const float *pa, const Offs &o)
    {
        r.a[i] = *reinterpret_cast<const float4 *>(pa + o.a[i]);
    }
    static __device__ __forceinline__ void load_b(Regs &r, int i, const float *pb, const Offs &o)
    {
        r.b[i] = *reinterpret_cast<const float4 *>(pb + o.b[i]);
    }
    static __device__ __forceinline__ void load(Regs &r, const float *pa, const float *pb, const Offs &o)
    {
#pragma unroll
        for (int i = 0; i < NVA; ++i) load_a(r, i, pa, o);
#pragma unroll
        for (int i = 0; i < NVB; ++i) load_b(r, i, pb, o);
    }

    // one float4 of the A / B register image -> LDS (k-major, transposing k-contiguous operands)
    static __device__ __forceinline__ void store_a(const Regs &r, int i, float *As, int tid)
    {
        const int f = tid + i * 256;
        const float4 v = r.a[i];          // (a local copy keeps the register set out of scratch)
        if constexpr (A_KC) {
            const int l8 = f % LPS, seg = f / LPS, row = seg % BM, k4 = (seg / BM) * LPS + l8;
            As[(k4 * 4 + 0) * LDA_S + row] = v.x; As[(k4 * 4 + 1) * LDA_S + row] = v.y;
            As[(k4 * 4 + 2) * LDA_S + row] = v.z; As[(k4 * 4 + 3) * LDA_S + row] = v.w;
        } else {
            const int c4 = f % (BM / 4), k = f / (BM / 4);
            *reinterpret_cast<float4 *>(As + k * LDA_S + c4 * 4) = v;
        }
    }
    static __device__ __forceinline__ void store_b(const Regs &r, int i, float *Bs, int tid, float4 &bsum)
    {
        const int f = tid + i * 256;
        const float4 v = r.b[i];
        if constexpr (B_KC) {
            const int l8 = f % LPS, seg = f / LPS, row = seg % BN, k4 = (seg / BN) * LPS + l8;
            Bs[(k4 * 4 + 0) * LDB_S + row] = v.x; Bs[(k4 * 4 + 1) * LDB_S + row] = v.y;
            Bs[(k4 * 4 + 2) * LDB_S + row] = v.z; Bs[(k4 * 4 + 3) * LDB_S + row] = v.w;
        } else {
            const int c4 = f % (BN / 4), k = f / (BN / 4);
            *reinterpret_cast<float4 *>(Bs + k * LDB_S + c4 * 4) = v;
            if constexpr (BIASG) {   // every thread keeps the same 4 columns across k-tiles
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;   // used by m-tile 0 only
            }
        }
    }
    static __device__ __forceinline__ void store(const Regs &r, float *As, float *Bs, int tid, float4 &bsum)
    {
#pragma unroll
        for (int i = 0; i < NVA; ++i) store_a(r, i, As, tid);
#pragma unroll
        for (int i = 0; i < NVB; ++i) store_b(r, i, Bs, tid, bsum);
    }

    // One k-tile: multiply the tile resident in LDS stage (As, Bs) into acc (this wave takes
    // k-range `ks`); between the MFMAs (a) fetch the operand fragments RD steps ahead, (b) issue
    // the global loads of a later k-tile into register image rl, (c) move the NEXT k-tile
    // (register image rs) into the other LDS stage.  Everything except the first RD fragment
    // reads issues while the matrix pipe is busy.  NCH independent accumulator chains
    // (a single dependent v_mfma_f32_32x32x2_f32 chain only reaches 90 % of the issue rate).
    static constexpr int NCH = (TM * TN == 1) ? 2 : 1;
    template <bool DO_STORE, bool DO_LOAD>
    static __device__ __forceinline__ void step(const float *As, const float *Bs, f32x16 (&acc)[NCH][TM][TN], int ks,
                                                int a_off, int b_off, int kh, const Regs &rs, float *AsN, float *BsN,
                                                Regs &rl, const float *pa, const float *pb, const Offs &o, int tid,
                                                float4 &bsum)
    {
        constexpr int NK = BK / KS / 2, NP = NVA + NVB, RD = NK < 4 ? NK : 4;
        const float *ap = As + (ks * (BK / KS) + kh) * LDA_S + a_off;
        const float *bp = Bs + (ks * (BK / KS) + kh) * LDB_S + b_off;
        float av[NK][TM], bv[NK][TN];
        // hipcc's scheduler otherwise sinks every ds_read next to its MFMA (one exposed LDS latency
        // per pair), gathers the ds_writes in front of the barrier and the global loads + their
        // address arithmetic in front of the first MFMA; pin the intended interleave.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < RD; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[s][i] = ap[2 * s * LDA_S + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[s][j] = bp[2 * s * LDB_S + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        int pl = 0, ps = 0;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[s % NCH][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][j], acc[s % NCH][i][j], 0, 0, 0);
            if (s + RD < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[s + RD][i] = ap[2 * (s + RD) * LDA_S + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[s + RD][j] = bp[2 * (s + RD) * LDB_S + j * 32];
            }
            if constexpr (DO_LOAD) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {               // load piece q goes after MFMA step q*NK/NP (spread evenly over the k-tile)
                    if (q == pl && (q * NK) / NP == s) {
                        if (q < NVA) load_a(rl, q, pa, o); else load_b(rl, q - NVA, pb, o);
                        ++pl;
                    }
                }
            }
            if constexpr (DO_STORE) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {               // store piece q: same slots
                    if (q == ps && (q * NK) / NP == s) {
                        if (q < NVA) store_a(rs, q, AsN, tid); else store_b(rs, q - NVA, BsN, tid, bsum);
                        ++ps;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// The workgroup program of one GEMM problem: workgroups first_block, first_block+stride, ... of
// the launch walk its tiles.  Wrapped by bp_gemm (one problem per launch) and bp_gemm_dual (two
// independent problems in one launch, see there).
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI>
struct GemmKernel {
    using Cfg = GemmCfg<BM, BN, BK, WM, WN, A_KC, B_KC, EPI>;
    static constexpr int KS = Cfg::KS, TM = Cfg::TM, TN = Cfg::TN;
    static constexpr int A_STAGE = Cfg::A_STAGE, B_STAGE = Cfg::B_STAGE, STAGE = A_STAGE + B_STAGE;
    static constexpr bool BIASG = Cfg::BIASG;
    static constexpr int RED = (KS > 1) ? KS * WM * WN * 16 * 64 : 0;     // k-split exchange area (floats)
    static constexpr int BIASRED = BIASG ? 256 / (BN / 4) * BN : 0;
    static constexpr int SMEM0 = (2 * STAGE > RED) ? 2 * STAGE : RED;
    static constexpr int SMEM = SMEM0 > BIASRED ? SMEM0 : BIASRED;       // floats of LDS
    // workgroups per CU the register allocator must leave room for (launch bounds)
    static constexpr int MIN_WG = (SMEM * 4 > 80 * 1024) ? 1 : 2;

static __device__ __forceinline__ void run(const GemmArgs &g_in, const EpiArgs &e_in, int first_block, int stride,
                                           int block_y, float *smem)
{
    using Regs = typename Cfg::Regs;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform => SGPR row bases
    const int ks = wave / (WM * WN), wq = wave % (WM * WN), wm = wq / WN, wn = wq % WN;
    GemmArgs g = g_in;
    EpiArgs e = e_in;
    constexpr int EPI_E = EPI == EPI_OUT_SPLIT ? EPI_FWD_OUT : EPI;      // the epilogue proper
    if constexpr (EPI == EPI_PARTIAL || EPI == EPI_OUT_SPLIT) {      // this workgroup row's k-slice and output slab
        const size_t kz = (size_t)block_y * g.k_split;
        g.A += A_KC ? kz : kz * g.lda;
        g.B += B_KC ? kz : kz * g.ldb;
        if constexpr (EPI == EPI_PARTIAL) e.C += (size_t)block_y * g.slab_stride;
    }

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous range of
    // n-tiles so the W / dEdX column panels it streams stay in its private L2.
    // Persistent over tiles (grid may be smaller than the tile count): the epilogue's stores of
    // one tile are still draining while the next tile's k-loop runs.
#ifdef BP_TRACE
#define TRACE(i) do { if (tid == 0 && g.trace) { g.trace[(size_t)first_block * 8 + (i)] = wall_clock64();                 \
        if ((i) == 0) g.trace[(size_t)first_block * 8 + 6] = clock64();   /* shader-clock counter: */              \
        if ((i) == 3) g.trace[(size_t)first_block * 8 + 7] = clock64();   /* effective MHz of the run */           \
    } } while (0)
#else
#define TRACE(i) ((void)0)
#endif
    TRACE(0);
    for (int b = first_block; b < g.tiles_m * g.tiles_n; b += stride) {
    int tile_m, tile_n;
    {
        if ((g.tiles_n & 7) == 0) {
            const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3;
            if constexpr (BIASG) {
                // wgrad: the XCD's few n-panels (dEdX columns) stay hot anyway; walk the m-panels (activation
                // columns) slowly so the `per` workgroups that share one run back to back and the panel is fetched
                // into this L2 once, instead of being evicted by the W/delta stream before its next use
                tile_n = xcd * per + j % per;
                tile_m = j / per;
            } else {
                tile_n = xcd * per + j / g.tiles_m;
                tile_m = j % g.tiles_m;
            }
        } else {
            tile_m = b % g.tiles_m;
            tile_n = b / g.tiles_m;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // bias-gradient partial sums (wgrad; used by m-tile 0)
    const bool do_bias = BIASG && tile_m == 0;

    constexpr int NCH = Cfg::NCH;
    f32x16 accs[NCH][TM][TN];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[c][i][j][r] = 0.0f;

    const int nt = (g.K + BK - 1) / BK;
    const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * TN * 32 + (lane & 31);
    const int kh = lane >> 5;

    // Epilogue inputs (bias / targ / y_prev / W, delta) are fetched up front so their HBM/L2 latency
    // hides under the k-loop (same lane->element map as the accumulator).
    EpiPre pre[TM][TN];
    const int mb0 = m0 + wm * TM * 32, nb0 = n0 + wn * TN * 32;
    if constexpr (KS == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) epilogue_fetch<EPI_E, 0, 16>(e, mb0 + i * 32, nb0 + j * 32, lane, pre[i][j]);
    } else {
        static_assert(KS == 1 || (TM == 1 && TN == 1), "in-workgroup k-split needs one block per wave");
        if (ks == 0) epilogue_fetch<EPI_E, 0, 16 / KS>(e, mb0, nb0, lane, pre[0][0]);
        if (ks == 1) epilogue_fetch<EPI_E, 16 / KS, 16 / KS>(e, mb0, nb0, lane, pre[0][0]);
        if constexpr (KS == 4) {
            if (ks == 2) epilogue_fetch<EPI_E, 8, 4>(e, mb0, nb0, lane, pre[0][0]);
            if (ks == 3) epilogue_fetch<EPI_E, 12, 4>(e, mb0, nb0, lane, pre[0][0]);
        }
    }

    // ---- software pipeline: LDS holds tile t (double buffered), two register images hold tiles
    // t+1 (landed, being staged) and t+2 (in flight).  One barrier per k-tile.  Every load in the steady-state loop is
    // unconditional (past the end the tile index is clamped and the data ignored): a
    // conditional load turns into a phi + copy and hipcc then waits for it right away.
    const int last_k0 = (nt - 1) * BK;
    typename Cfg::Offs offs;
    Cfg::make_offs(offs, g, tid);
#define K0_OF(t) (((t) * BK) < last_k0 ? ((t) * BK) : last_k0)
#define PA(t) Cfg::base_a(g, m0, K0_OF(t))
#define PB(t) Cfg::base_b(g, n0, K0_OF(t))
#define AS(buf) (smem + (buf) * STAGE)
#define BS(buf) (smem + (buf) * STAGE + A_STAGE)
// multiply stage `buf`; ST: store image RS into the other stage; LD: load tile TL into image RL
#define STEP(ST, LD, buf, RS, RL, TL)                                                                      \
    Cfg::template step<ST, LD>(AS(buf), BS(buf), accs, ks, a_off, b_off, kh, RS, AS((buf) ^ 1), BS((buf) ^ 1),    \
                               RL, PA(TL), PB(TL), offs, tid, bsum)
    Regs r0, r1;
    Cfg::load(r0, PA(0), PB(0), offs);
    Cfg::store(r0, AS(0), BS(0), tid, bsum);
    Cfg::load(r1, PA(1), PB(1), offs);
    __syncthreads();
    TRACE(1);
    // Invariant at the top of iteration t: LDS stage t&1 holds tile t; tile t+1 is in flight / landed in registers
    // (r1 for even t, r0 for odd t).  The iteration multiplies tile t and, between the MFMAs, issues the global loads of tile
    // t+2 and moves tile t+1 into the other stage.  The steady-state loop contains no conditionals (a conditional step makes
    // hipcc copy the accumulators between register ranges every iteration); the last one or two tiles run after.
    int t = 0, buf = 0;
    for (; t + 3 <= nt; t += 2) {
        STEP(true, true, 0, r1, r0, t + 2);
        __syncthreads();
        STEP(true, true, 1, r0, r1, t + 3);
        __syncthreads();
    }
    if (nt - t == 2) {
        STEP(true, false, 0, r1, r0, 0);
        __syncthreads();
        buf = 1;
    }
    STEP(false, false, buf, r0, r0, 0);    // last tile: nothing left to stage or fetch
    __syncthreads();
    TRACE(2);
#undef K0_OF
#undef PA
#undef PB
#undef AS
#undef BS
#undef STEP
    // fold the independent accumulator chains
    f32x16 (&acc)[TM][TN] = accs[0];
    if constexpr (NCH == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += accs[1][i][j][r];
    }

    // ---- in-workgroup k-split: exchange partial sums through LDS (smem is free after the last
    // barrier); afterwards wave group ks owns registers [ks*16/KS, (ks+1)*16/KS) of its block
    if constexpr (KS > 1) {
        if (ks == 0) ksplit_give<KS, 0>(acc[0][0], smem, wq, lane);
        if (ks == 1) ksplit_give<KS, 1>(acc[0][0], smem, wq, lane);
        if constexpr (KS == 4) {
            if (ks == 2) ksplit_give<KS, 2>(acc[0][0], smem, wq, lane);
            if (ks == 3) ksplit_give<KS, 3>(acc[0][0], smem, wq, lane);
        }
        __syncthreads();
        if (ks == 0) ksplit_take<KS, 0>(acc[0][0], smem, wq, lane);
        if (ks == 1) ksplit_take<KS, 1>(acc[0][0], smem, wq, lane);
        if constexpr (KS == 4) {
            if (ks == 2) ksplit_take<KS, 2>(acc[0][0], smem, wq, lane);
            if (ks == 3) ksplit_take<KS, 3>(acc[0][0], smem, wq, lane);
        }
        if constexpr (BIASG) __syncthreads();
    }

    // ---- bias gradient: column sums of the dEdX panel this workgroup streamed (kernAccSumrow)
    if constexpr (BIASG && !B_KC) {
        if (do_bias) {
            constexpr int CG = BN / 4, RG = 256 / CG;          // column groups x row groups
            float *red = smem;                                 // [RG][BN]
            const int c4 = tid % CG, rg = tid / CG;
            *reinterpret_cast<float4 *>(red + rg * BN + c4 * 4) = bsum;
            __syncthreads();
            if (tid < BN && (n0 + tid) < e.n_limit) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < RG; ++r) s += red[r * BN + tid];
                const int n = n0 + tid;
                if constexpr (EPI == EPI_WGRAD_UPDATE) {
                    const float d = e.mom * e.bias_d[n] - e.c1 * (s / e.ndiv + 0.0f * e.bias_w[n]);
                    e.bias_d[n] = d;
                    e.bias_w[n] = d + 1.0f * e.bias_w[n];
                } else {
                    e.bias_g[n] = s;
                }
            }
        }
    }

    // ---- split-K across workgroups in one launch (EPI_OUT_SPLIT): every slice writes its partial tile, drains, takes a ticket from the
    // tile's word; the last of the OUT_SPLITS arrivals of this launch sums the slices and runs the output epilogue, the others are
    // done.  Nobody waits for anybody.  (The ticket words only grow: OUT_SPLITS per launch; the slices of a tile are workgroups
    // b, b + tiles, ... of the launch -- on one XCD, block index mod 8, whenever the tile count is a multiple of 8.)
    bool finish = true;
    if constexpr (EPI == EPI_OUT_SPLIT) {
        static_assert(KS == 4 && OUT_SPLITS == 4, "one 32x32 block per workgroup, four registers of it per wave");
        float *mine = g.ks_slab + (size_t)block_y * g.slab_stride;
        if (ks == 0) out_split_store<0>(e, mine, mb0, nb0, acc[0][0], lane);
        if (ks == 1) out_split_store<4>(e, mine, mb0, nb0, acc[0][0], lane);
        if (ks == 2) out_split_store<8>(e, mine, mb0, nb0, acc[0][0], lane);
        if (ks == 3) out_split_store<12>(e, mine, mb0, nb0, acc[0][0], lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // (also: every wave is past its reads of the exchange area)
        unsigned *tk = reinterpret_cast<unsigned *>(smem);
        if (tid == 0) *tk = __hip_atomic_fetch_add(g.ks_ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = (*tk & (OUT_SPLITS - 1)) == OUT_SPLITS - 1;
        if (finish) {
            if (ks == 0) out_split_sum<0>(e, g.ks_slab, g.slab_stride, mb0, nb0, acc[0][0], lane);
            if (ks == 1) out_split_sum<4>(e, g.ks_slab, g.slab_stride, mb0, nb0, acc[0][0], lane);
            if (ks == 2) out_split_sum<8>(e, g.ks_slab, g.slab_stride, mb0, nb0, acc[0][0], lane);
            if (ks == 3) out_split_sum<12>(e, g.ks_slab, g.slab_stride, mb0, nb0, acc[0][0], lane);
        }
    }

    if constexpr (KS == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                epilogue_block<EPI_E, 0, 16>(e, mb0 + i * 32, nb0 + j * 32, acc[i][j], lane, pre[i][j]);
    } else if (finish) {
        if (ks == 0) epilogue_block<EPI_E, 0, 16 / KS>(e, mb0, nb0, acc[0][0], lane, pre[0][0]);
        if (ks == 1) epilogue_block<EPI_E, 16 / KS, 16 / KS>(e, mb0, nb0, acc[0][0], lane, pre[0][0]);
        if constexpr (KS == 4) {
            if (ks == 2) epilogue_block<EPI_E, 8, 4>(e, mb0, nb0, acc[0][0], lane, pre[0][0]);
            if (ks == 3) epilogue_block<EPI_E, 12, 4>(e, mb0, nb0, acc[0][0], lane, pre[0][0]);
        }
    }
    if (b + stride < g.tiles_m * g.tiles_n) __syncthreads();   // smem is reused by the next tile
    }   // tile loop
    TRACE(3);
#undef TRACE
}
};   // GemmKernel

// TAG only separates instantiations by name (profilers report per kernel name): 1 = input-layer forward.
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI, int TAG = 0>
__global__ __launch_bounds__(256, (GemmKernel<BM, BN, BK, WM, WN, A_KC, B_KC, EPI>::MIN_WG)) void bp_gemm(const GemmArgs g, const EpiArgs e)
{
    using K = GemmKernel<BM, BN, BK, WM, WN, A_KC, B_KC, EPI>;
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    K::run(g, e, blockIdx.x, gridDim.x, blockIdx.y, smem);
}

// Up to 4 INDEPENDENT problems of the same tile configuration in one launch (grouped GEMM): the
// wgrad+update of several layers after the last dgrad of the step.  Each problem alone is short
// (K = bunch = 256: 16 k-tiles per workgroup) and its workgroups run in lockstep -- prologue, MFMA
// phase and the W/delta write-back each hit the chip all at once.  In one launch of ~7 rounds of
// workgroups the rounds drift apart, so the HBM phases of some overlap the MFMA phases of others
// and only one launch boundary / cold start is paid.
struct MultiArgs {
    GemmArgs g[4];
    EpiArgs e[4];
    int first_tile[5];       // first_tile[p] .. first_tile[p+1]-1 = workgroups of problem p
    int n;
};
template <class K>
__global__ __launch_bounds__(256, K::MIN_WG) void bp_gemm_multi(const MultiArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
    K::run(a.g[p], a.e[p], b - a.first_tile[p], a.first_tile[p + 1] - a.first_tile[p], 0, smem);
}

// ------------------------------------------------------------------ small kernels
// On-device frame stacking, one bunch at a time (include/bp_c_api.h, bp_window_chunk; the host does it per chunk,
// Interface.cc:757-797).  Rows 0..rows-1 of the bunch (the caller passes the tables already offset to its first sample)
// become the input tile x [rows][ld]: `win` consecutive floats of the raw frame matrix starting at frame win_start[i]
// (a context window is contiguous in memory), then the sentence's noise-aware block, then zero padding up to ld -- with
// the visible-layer dropout of BP_GPU.cu:536-539 drawn on the way (thresh != 0; the Philox words of bp_mask_input:
// key (global frame >> 2, unit), layer 0) -- and the bunch's targets t [rows][ldt] = targ_frames[targ_frame[i]]
// (Interface.cc:792-797; t may be null).  The stacked chunk (n_samples x 2827 floats) and its masked copy are never
// materialised: what stays resident are the raw frames (11x fewer bytes) and this one L2-sized tile per bunch.
// One thread = one column x 4 consecutive rows (= one Philox block; coalesced 4-byte accesses: raw rows of 257 floats
// are not 16-byte aligned); blockIdx.y < yb_in sweeps the input columns, the rest the target columns.
struct StageArgs {
    float *x; int ld, width;                      // staged input tile [rows][ld], unpadded width
    const float *fea; int fea_dim, win;           // raw frames [n_frames][fea_dim]; win = context * fea_dim floats per window
    const float *nat; const int *win_start, *nat_row;   // noise-aware rows; per-sample tables (already offset to the bunch's first sample); win_start == null: stacked rows (stage_rows_block)
    int rows; uint32_t thresh; int frame_off; uint32_t seed_lo, seed_hi, step;
    float *t; int ldt, twidth; const float *targ_frames; const int *targ_frame;   // staged targets (t may be null)
    int yb_in;                                    // column blocks of the input part; blocks by >= yb_in sweep the target columns
    int nbx;                                      // row blocks (rows / 4, rounded up): block (bx, by) of the flat index e is bx = e % nbx, by = e / nbx
};
// The same for a STACKED chunk (win_start == null: the caller handed rows that are already stacked, BP_GPU.cu:127-130): rows
// r0..r0+3 of the bunch are copied as they lie, fea = first row of the bunch, fea_dim = its leading dimension (a multiple of 4
// floats, rows 16-byte aligned), with the visible-layer dropout of BP_GPU.cu:536-539 applied on the way instead of in place in
// the cached chunk.  One thread = 4 rows x 4 consecutive columns: four 16-byte loads, four Philox blocks (one per column, all
// four words used), four 16-byte stores.
__device__ __forceinline__ void stage_rows_block(const StageArgs &a, int bx, int by, int tx)
{
    const int r0 = bx * 4, c = (by * 256 + tx) * 4;
    if (c >= a.ld) return;
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        v[j] = *reinterpret_cast<const float4 *>(a.fea + (size_t)(r0 + j < a.rows ? r0 + j : a.rows - 1) * a.fea_dim + c);   // (clamped, unconditional: four loads in flight)
    if (a.thresh) {
        // one Philox block per column; the loop is NOT unrolled (four inlined copies of the ten rounds were 3 KB of a launch whose code
        // shares the instruction caches with the rest of the step): the decisions are collected as bits
        // (bit 4k + j: row j of column k) and applied with constant indices
        uint32_t drop = 0;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            if (c + k >= a.width) break;
            uint32_t w[4];
            drop_words4(w, r0, c + k, a.frame_off, (uint32_t)a.width, 0u, a.step, a.seed_lo, a.seed_hi);
            const uint32_t m = (w[0] < a.thresh ? 1u : 0u) | (w[1] < a.thresh ? 2u : 0u) | (w[2] < a.thresh ? 4u : 0u) | (w[3] < a.thresh ? 8u : 0u);
            drop |= m << (4 * k);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (drop & (1u << (0 + j))) v[j].x = 0.0f;
            if (drop & (1u << (4 + j))) v[j].y = 0.0f;
            if (drop & (1u << (8 + j))) v[j].z = 0.0f;
            if (drop & (1u << (12 + j))) v[j].w = 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (r0 + j < a.rows) *reinterpret_cast<float4 *>(a.x + (size_t)(r0 + j) * a.ld + c) = v[j];
}
__device__ __forceinline__ void stage_block(const StageArgs &a, int bx, int by, int tx)
{
    if (!a.win_start) { stage_rows_block(a, bx, by, tx); return; }
    const int r0 = bx * 4;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // The table entries of the four rows are block-uniform (scalar loads).  Rows past the bunch's end read the LAST row's entry and
    // data (unconditional, clamped) and are only kept from the stores: guarded per row, every row became s_load, s_waitcnt, load --
    // four dependent scalar round trips in front of the four row loads of a launch that is nothing but latency.
    int rr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rr[j] = r0 + j < a.rows ? r0 + j : a.rows - 1;
    if (by >= a.yb_in) {
        const int c = (by - a.yb_in) * 256 + tx;
        if (c >= a.ldt) return;
        if (c < a.twidth) {
            int tf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tf[j] = a.targ_frame[rr[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a.targ_frames[(size_t)tf[j] * a.twidth + c];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j < a.rows) a.t[(size_t)(r0 + j) * a.ldt + c] = v[j];
        return;
    }
    const int c = by * 256 + tx;
    if (c >= a.ld) return;
    // all four rows' loads first (independent), the Philox block meanwhile, then the four stores
    if (c < a.win) {
        int ws[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ws[j] = a.win_start[rr[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.fea[(size_t)ws[j] * a.fea_dim + c];
    } else if (c < a.width) {
        int nr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nr[j] = a.nat_row[rr[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.nat[(size_t)nr[j] * a.fea_dim + (c - a.win)];
    }
    uint32_t w[4] = {~0u, ~0u, ~0u, ~0u};
    if (a.thresh && c < a.width) drop_words4(w, r0, c, a.frame_off, (uint32_t)a.width, 0u, a.step, a.seed_lo, a.seed_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (r0 + j < a.rows) a.x[(size_t)(r0 + j) * a.ld + c] = w[j] < a.thresh ? 0.0f : v[j];
}
__global__ __launch_bounds__(256) void bp_stage_bunch(const StageArgs a)
{
    stage_block(a, blockIdx.x % a.nbx, blockIdx.x / a.nbx, threadIdx.x);
}

// Injected visible-layer mask (bp_train_resident_masked, parity tests): out[f][u] = mask[f][u] ? 0 : in[f][u]
__global__ void bp_apply_mask(const float *in, float *out, int ld, int width, const uint8_t *mask, int n_frames)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (u >= ld || f >= n_frames) return;
    float v = in[(size_t)f * ld + u];
    if (u < width && mask[(size_t)f * width + u]) v = 0.0f;
    out[(size_t)f * ld + u] = v;
}

// Synthetic N(0,1) fill of a padded [rows][ld] buffer (cols >= width stay 0): Philox + Box-Muller.
__global__ void bp_fill_normal(float *buf, int ld, int width, int rows, uint32_t seed_lo, uint32_t seed_hi,
                               uint32_t stream)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread = 4 elements
    const size_t per_row = (size_t)(ld / 4);
    if (i >= per_row * (size_t)rows) return;
    const size_t r = i / per_row; const int c = (int)(i % per_row) * 4;
    uint32_t w0 = (uint32_t)i, w1 = (uint32_t)(i >> 32), w2 = stream, w3 = 0x5EEDu;
    philox4x32_10(w0, w1, w2, w3, seed_lo, seed_hi);
    const float u0 = (w0 + 1.0f) * 2.3283064365386963e-10f, u1 = w1 * 2.3283064365386963e-10f;
    const float u2 = (w2 + 1.0f) * 2.3283064365386963e-10f, u3 = w3 * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(fminf(u0, 1.0f))), r1 = sqrtf(-2.0f * logf(fminf(u2, 1.0f)));
    float v[4] = { r0 * cosf(6.283185307179586f * u1), r0 * sinf(6.283185307179586f * u1),
                   r1 * cosf(6.283185307179586f * u3), r1 * sinf(6.283185307179586f * u3) };
    float4 o;
    o.x = (c + 0 < width) ? v[0] : 0.f; o.y = (c + 1 < width) ? v[1] : 0.f;
    o.z = (c + 2 < width) ? v[2] : 0.f; o.w = (c + 3 < width) ? v[3] : 0.f;
    *reinterpret_cast<float4 *>(buf + r * ld + c) = o;
}

// The narrow output layer's forward, split over OUT_SPLITS k-slices (EPI_OUT_SPLIT; workgroup b < n_gemm: slice b / tiles of tile
// b % tiles) AND, in the same launch, the stacking of the NEXT bunch into the other staged tile (workgroups >= n_gemm; none if
// st.nbx == 0): the layer is 320 workgroups of launch latency on a chip that is otherwise idle at that point of the step, so the next
// bunch's bp_stage_bunch (5.1 us as its own launch in front of every bunch, round 3) rides along for free.  (Rounds 3-5 ran the layer
// as two launches -- partial sums, then a reduce launch that carried the staging: 7.9 + 5.2 us at configs[1].)
template <class K>
__global__ __launch_bounds__(256, K::MIN_WG) void bp_out_split_stage(const GemmArgs g, const EpiArgs e, int n_gemm, const StageArgs st)
{
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    const int b = blockIdx.x;
    if (b < n_gemm) { const int tiles = g.tiles_m * g.tiles_n; K::run(g, e, b % tiles, tiles, b / tiles, smem); }
    else { const int i = b - n_gemm; stage_block(st, i % st.nbx, i / st.nbx, threadIdx.x); }
}

// Momentum update on a flat [W|b] gradient segment after the data-parallel sum
// (kernUpdatedelta + kernAccSum, DevFunc.cu:313-318, 270-277); wc applies to the W part only.
__global__ void bp_update_flat(float *w, float *d, const float *g, size_t n_w, float *bw, float *bd,
                               const float *bg, int n_b, float mom, float c1, float wc, float ndiv)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_w; i += stride) {
        const float wi = w[i];
        const float di = mom * d[i] - c1 * (g[i] / ndiv + wc * wi);
        d[i] = di; w[i] = di + 1.0f * wi;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n_b; i += stride) {
        const float wi = bw[i];
        const float di = mom * bd[i] - c1 * (bg[i] / ndiv + 0.0f * wi);
        bd[i] = di; bw[i] = di + 1.0f * wi;
    }
}
