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
// operands go in with ds_write_b128.  Register-staged double buffering: global loads of
// k-tile t+1 are in flight while tile t is multiplied; one barrier per k-tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_FWD_HIDDEN = 0, EPI_FWD_OUT = 1, EPI_DGRAD = 2, EPI_WGRAD_UPDATE = 3, EPI_WGRAD_STORE = 4 };

struct GemmArgs {
    const float *A, *B;
    int lda, ldb;            // leading dimensions (floats)
    int K;                   // reduction extent actually looped (rounded up to BK inside)
    int a_row_limit;         // rows of A's non-contiguous index that exist (m for [m][k], k for [k][m])
    int b_row_limit;         // same for B (n for [n][k], k for [k][n])
    int a_col_limit, b_col_limit; // extent of the contiguous index (padded leading extent)
    int tiles_m, tiles_n;
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
template <int EPI>
__device__ __forceinline__ void epilogue_block(const EpiArgs &e, int mb, int nb, const f32x16 &acc, int lane)
{
    const int n = nb + (lane & 31);
    const int rbase = mb + 4 * (lane >> 5);
    if (n >= e.n_limit) return;
    if constexpr (EPI == EPI_FWD_HIDDEN) {
        const float bn = e.bias[n];
        const bool live = n < e.n_true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            const int r0 = rbase + 8 * q;
            if (e.drop_thresh) {
                const uint64_t gf = (uint64_t)(uint32_t)(r0 + e.frame_off);
                const uint64_t idx = (gf >> 2) * (uint64_t)(uint32_t)e.n_true + (uint32_t)n;
                w[0] = (uint32_t)idx; w[1] = (uint32_t)(idx >> 32); w[2] = e.layer; w[3] = e.step;
                philox4x32_10(w[0], w[1], w[2], w[3], e.seed_lo, e.seed_hi);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = r0 + j;
                float y = act_fwd(e.act, e.alpha * acc[q * 4 + j] + bn);
                if (!live || w[j] < e.drop_thresh) y = 0.0f;
                if (m < e.m_limit) e.C[(size_t)m * e.ldc + n] = y;
            }
        }
    } else if constexpr (EPI == EPI_FWD_OUT) {
        const float bn = e.bias[n];
        const bool live = n < e.n_true;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) {
                const float o = live ? e.alpha * acc[r] + bn : 0.0f;
                if (e.aux2) e.aux2[(size_t)m * e.ldaux2 + n] = o;
                if (e.C) {
                    const float t = e.aux[(size_t)m * e.ldaux + n];
                    e.C[(size_t)m * e.ldc + n] = live ? e.scale * (o - t) : 0.0f;   // kernSubClean
                }
            }
        }
    } else if constexpr (EPI == EPI_DGRAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) {
                const float y = e.aux[(size_t)m * e.ldaux + n];
                e.C[(size_t)m * e.ldc + n] = act_bwd(e.act, y) * acc[r];        // kernDsigmoid*kernVecMul
            }
        }
    } else if constexpr (EPI == EPI_WGRAD_UPDATE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) {
                const size_t i = (size_t)m * e.ldc + n;
                const float w = e.C[i];
                const float d = e.mom * e.aux2[i] - e.c1 * (acc[r] / e.ndiv + e.wc * w);  // kernUpdatedelta
                e.aux2[i] = d;
                e.C[i] = d + 1.0f * w;                                                     // kernAccSum
            }
        }
    } else {  // EPI_WGRAD_STORE
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < e.m_limit) e.C[(size_t)m * e.ldc + n] = acc[r];
        }
    }
}

// ------------------------------------------------------------------ the GEMM
// BM x BN x BK workgroup tile, 4 waves arranged WM x WN x KS (KS = 4/(WM*WN) splits each
// k-tile between wave groups; partial sums meet in LDS before the epilogue).
// A_KC: A is [m][k] in memory (k contiguous) else [k][m]; B_KC: B is [n][k] else [k][n].
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void bp_gemm(const GemmArgs g, const EpiArgs e)
{
    constexpr int KS = 4 / (WM * WN);
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert(WM * WN * KS == 4 && TM >= 1 && TN >= 1, "wave layout");
    static_assert(BK % (2 * KS) == 0 && BK % 4 == 0, "BK");
    constexpr int LDA_S = A_KC ? BM + 1 : BM;
    constexpr int LDB_S = B_KC ? BN + 1 : BN;
    constexpr int A_STAGE = (BK * LDA_S + 3) & ~3, B_STAGE = (BK * LDB_S + 3) & ~3;
    constexpr int NVA = BM * BK / 4 / 256, NVB = BN * BK / 4 / 256;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 threads");
    constexpr int RED = (KS > 1) ? (KS - 1) * WM * WN * TM * TN * 16 * 64 : 0;
    constexpr int SMEM = (2 * (A_STAGE + B_STAGE) > RED) ? 2 * (A_STAGE + B_STAGE) : RED;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (WM * WN), wq = wave % (WM * WN), wm = wq / WN, wn = wq % WN;

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous range of
    // n-tiles so the W / dEdX column panels it streams stay in its private L2.
    int tile_m, tile_n;
    {
        const int b = blockIdx.x, T = g.tiles_m * g.tiles_n;
        if ((g.tiles_n & 7) == 0) {
            const int xcd = b & 7, j = b >> 3, per = g.tiles_n >> 3;
            tile_n = xcd * per + j / g.tiles_m;
            tile_m = j % g.tiles_m;
        } else {
            tile_m = b % g.tiles_m;
            tile_n = b / g.tiles_m;
        }
        (void)T;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    float4 ra[NVA], rb[NVB];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};   // bias-gradient partial sums (wgrad, m-tile 0 only)
    constexpr bool BIASG = (EPI == EPI_WGRAD_UPDATE || EPI == EPI_WGRAD_STORE);
    const bool do_bias = BIASG && tile_m == 0;

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int f = tid + i * 256;
            if constexpr (A_KC) {
                constexpr int LPS = (BK / 4 < 8) ? BK / 4 : 8;
                const int l8 = f % LPS, seg = f / LPS, r = seg % BM, k4 = (seg / BM) * LPS + l8;
                const bool ok = (m0 + r) < g.a_row_limit && (k0 + k4 * 4) < g.a_col_limit;
                ra[i] = ok ? *reinterpret_cast<const float4 *>(g.A + (size_t)(m0 + r) * g.lda + k0 + k4 * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int c4 = f % (BM / 4), k = f / (BM / 4);
                const bool ok = (k0 + k) < g.a_row_limit && (m0 + c4 * 4) < g.a_col_limit;
                ra[i] = ok ? *reinterpret_cast<const float4 *>(g.A + (size_t)(k0 + k) * g.lda + m0 + c4 * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int f = tid + i * 256;
            if constexpr (B_KC) {
                constexpr int LPS = (BK / 4 < 8) ? BK / 4 : 8;
                const int l8 = f % LPS, seg = f / LPS, r = seg % BN, k4 = (seg / BN) * LPS + l8;
                const bool ok = (n0 + r) < g.b_row_limit && (k0 + k4 * 4) < g.b_col_limit;
                rb[i] = ok ? *reinterpret_cast<const float4 *>(g.B + (size_t)(n0 + r) * g.ldb + k0 + k4 * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int c4 = f % (BN / 4), k = f / (BN / 4);
                const bool ok = (k0 + k) < g.b_row_limit && (n0 + c4 * 4) < g.b_col_limit;
                rb[i] = ok ? *reinterpret_cast<const float4 *>(g.B + (size_t)(k0 + k) * g.ldb + n0 + c4 * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float *As = smem + buf * (A_STAGE + B_STAGE), *Bs = As + A_STAGE;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int f = tid + i * 256;
            if constexpr (A_KC) {
                constexpr int LPS = (BK / 4 < 8) ? BK / 4 : 8;
                const int l8 = f % LPS, seg = f / LPS, r = seg % BM, k4 = (seg / BM) * LPS + l8;
                As[(k4 * 4 + 0) * LDA_S + r] = ra[i].x; As[(k4 * 4 + 1) * LDA_S + r] = ra[i].y;
                As[(k4 * 4 + 2) * LDA_S + r] = ra[i].z; As[(k4 * 4 + 3) * LDA_S + r] = ra[i].w;
            } else {
                const int c4 = f % (BM / 4), k = f / (BM / 4);
                *reinterpret_cast<float4 *>(As + k * LDA_S + c4 * 4) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int f = tid + i * 256;
            if constexpr (B_KC) {
                constexpr int LPS = (BK / 4 < 8) ? BK / 4 : 8;
                const int l8 = f % LPS, seg = f / LPS, r = seg % BN, k4 = (seg / BN) * LPS + l8;
                Bs[(k4 * 4 + 0) * LDB_S + r] = rb[i].x; Bs[(k4 * 4 + 1) * LDB_S + r] = rb[i].y;
                Bs[(k4 * 4 + 2) * LDB_S + r] = rb[i].z; Bs[(k4 * 4 + 3) * LDB_S + r] = rb[i].w;
            } else {
                const int c4 = f % (BN / 4), k = f / (BN / 4);
                *reinterpret_cast<float4 *>(Bs + k * LDB_S + c4 * 4) = rb[i];
                if constexpr (BIASG) {   // every thread keeps the same 4 columns across k-tiles
                    if (do_bias) { bsum[0] += rb[i].x; bsum[1] += rb[i].y; bsum[2] += rb[i].z; bsum[3] += rb[i].w; }
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nt = (g.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int a_off = wm * TM * 32 + (lane & 31), b_off = wn * TN * 32 + (lane & 31);
    const int kh = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tiles((t + 1) * BK);
        const float *As = smem + buf * (A_STAGE + B_STAGE), *Bs = As + A_STAGE;
#pragma unroll
        for (int kk = ks * (BK / KS); kk < (ks + 1) * (BK / KS); kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[(kk + kh) * LDA_S + a_off + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bs[(kk + kh) * LDB_S + b_off + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- meet the k-split partial sums in LDS (smem is free after the last barrier)
    if constexpr (KS > 1) {
        if (ks > 0) {
            float *red = smem + ((ks - 1) * WM * WN + wq) * (TM * TN * 16 * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int s = 1; s < KS; ++s) {
                const float *red = smem + ((s - 1) * WM * WN + wq) * (TM * TN * 16 * 64);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * 64 + lane];
            }
        }
        if constexpr (BIASG) __syncthreads();
    }

    // ---- bias gradient: column sums of the dEdX panel this workgroup streamed (kernAccSumrow)
    if constexpr (BIASG && !B_KC) {
        if (do_bias) {
            constexpr int CG = BN / 4, RG = 256 / CG;          // column groups x row groups
            float *red = smem;                                 // [RG][BN]
            const int c4 = tid % CG, rg = tid / CG;
            red[rg * BN + c4 * 4 + 0] = bsum[0]; red[rg * BN + c4 * 4 + 1] = bsum[1];
            red[rg * BN + c4 * 4 + 2] = bsum[2]; red[rg * BN + c4 * 4 + 3] = bsum[3];
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

    if (ks == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                epilogue_block<EPI>(e, m0 + wm * TM * 32 + i * 32, n0 + wn * TN * 32 + j * 32, acc[i][j], lane);
    }
}

// ------------------------------------------------------------------ small kernels
// Visible-layer dropout of the resident chunk (BP_GPU.cu:536-539 masks the device copy of the
// chunk in place; here the masked frames go to a second buffer so the chunk stays reusable).
// One thread = one unit x 4 consecutive chunk rows.
__global__ void bp_mask_input(const float *in, float *out, int ld, int width, int first_frame, int n_frames,
                              int bunch, int frame_off, uint32_t thresh, uint32_t seed_lo, uint32_t seed_hi,
                              uint32_t step0)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int g4 = blockIdx.y;
    if (u >= ld) return;
    uint32_t w[4]; uint64_t cur_blk = ~0ull; uint32_t cur_step = 0;
    for (int j = 0; j < 4; ++j) {
        const int rel = g4 * 4 + j;
        if (rel >= n_frames) break;
        const int f = first_frame + rel;
        float v = in[(size_t)f * ld + u];
        if (u < width) {
            const uint32_t step = step0 + (uint32_t)(rel / bunch);
            const uint64_t gf = (uint64_t)(uint32_t)(rel % bunch + frame_off);
            const uint64_t blk = gf >> 2;
            if (blk != cur_blk || step != cur_step) {
                const uint64_t idx = blk * (uint64_t)(uint32_t)width + (uint32_t)u;
                w[0] = (uint32_t)idx; w[1] = (uint32_t)(idx >> 32); w[2] = 0u; w[3] = step;
                philox4x32_10(w[0], w[1], w[2], w[3], seed_lo, seed_hi);
                cur_blk = blk; cur_step = step;
            }
            if (w[gf & 3] < thresh) v = 0.0f;
        }
        out[(size_t)f * ld + u] = v;
    }
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
