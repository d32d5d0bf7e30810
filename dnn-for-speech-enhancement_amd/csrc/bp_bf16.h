// bp_bf16.h -- the same training step with bf16 GEMM operands (BASELINE.json configs[4]: "bf16 forward/backward,
// fp32 master weights").  fp32 stays the type of: accumulation (v_mfma_f32_32x32x16_bf16), bias,
// the activation / loss arithmetic of the epilogues, the master weights W, the momentum state delta and the update.
// bf16 (round-to-nearest-even) is the storage type of everything a GEMM reads: activations y_l, back-propagated
// errors dEdX_l and a shadow copy of the weights that the update epilogue refreshes.
//
// Layout idea: every GEMM of the step becomes  C[m][n] = sum_k A[m][k] * B[n][k]  with BOTH operands k-contiguous,
// because each bf16 array is kept in two orientations, written together by the producing epilogue (a lane owns one
// column and 4-row groups of a 32x32 MFMA block, so the transposed copy is written as 8-byte runs):
//     fwd   l : y_l[f][c]      = act( y_{l-1}[f][p] . WbT_l[c][p] )          y  : [frames][units]   yT : [units][frames]
//     dgrad l : dx_{l-1}[f][p] = act'(y_{l-1}) * ( dx_l[f][c] . Wb_l[p][c] ) dx : [frames][units]   dxT: [units][frames]
//     wgrad l : G_l[p][c]      = yT_{l-1}[p][f] . dxT_l[c][f]                Wb : [prev][cur]       WbT: [cur][prev]
// so one kernel (64x64x64 tiles, 4 waves, ds_read_b128 fragment reads from padded k-contiguous LDS rows) serves all
// three with different epilogues.  This path is a parity configuration, not the benchmarked one: it is written for
// clarity, not tuned to the bf16 MFMA peak (a 32x32 block per wave is LDS-read bound at ~half of it).
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

static constexpr int BF_BM = 64, BF_BN = 64, BF_BK = 64, BF_LDS = BF_BK + 8;   // LDS row stride in halfs (144 B: conflict-free b128)

template <int EPI>
__global__ __launch_bounds__(256, 2) void bp_gemm_bf16(const BfGemmArgs g, const BfEpiArgs e)
{
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BF_BM + BF_BN) * BF_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x % g.tiles_m, tile_n = blockIdx.x / g.tiles_m;
    const int m0 = tile_m * BF_BM, n0 = tile_n * BF_BN;
    // this thread's two 16-byte chunks of each operand tile: chunk c -> row c>>3, 8 halfs at column (c&7)*8
    const int c0 = tid, c1 = tid + 256;
    const int r0 = c0 >> 3, k0c = (c0 & 7) * 8, r1 = c1 >> 3, k1c = (c1 & 7) * 8;
    const bf16_t *pa0 = g.A + (size_t)(m0 + r0) * g.lda + k0c, *pa1 = g.A + (size_t)(m0 + r1) * g.lda + k1c;
    const bf16_t *pb0 = g.B + (size_t)(n0 + r0) * g.ldb + k0c, *pb1 = g.B + (size_t)(n0 + r1) * g.ldb + k1c;
    auto As = [&](int st) { return smem + st * (BF_BM + BF_BN) * BF_LDS; };
    auto Bs = [&](int st) { return smem + st * (BF_BM + BF_BN) * BF_LDS + BF_BM * BF_LDS; };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nt = g.K / BF_BK;
    uint4 ra0, ra1, rb0, rb1;
#define BF_LOAD(t)                                                                                      \
    do {                                                                                                \
        const int kk_ = ((t) < nt ? (t) : nt - 1) * BF_BK;                                              \
        ra0 = *reinterpret_cast<const uint4 *>(pa0 + kk_); ra1 = *reinterpret_cast<const uint4 *>(pa1 + kk_); \
        rb0 = *reinterpret_cast<const uint4 *>(pb0 + kk_); rb1 = *reinterpret_cast<const uint4 *>(pb1 + kk_); \
    } while (0)
#define BF_STORE(st)                                                                                    \
    do {                                                                                                \
        *reinterpret_cast<uint4 *>(As(st) + r0 * BF_LDS + k0c) = ra0; *reinterpret_cast<uint4 *>(As(st) + r1 * BF_LDS + k1c) = ra1; \
        *reinterpret_cast<uint4 *>(Bs(st) + r0 * BF_LDS + k0c) = rb0; *reinterpret_cast<uint4 *>(Bs(st) + r1 * BF_LDS + k1c) = rb1; \
    } while (0)
    BF_LOAD(0);
    BF_STORE(0);
    BF_LOAD(1);
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), kh = (lane >> 5) * 8;
    for (int t = 0; t < nt; ++t) {
        const int st = t & 1;
        const bf16_t *ap = As(st) + arow * BF_LDS + kh, *bp = Bs(st) + brow * BF_LDS + kh;
        bf16x8_t av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av[kk] = *reinterpret_cast<const bf16x8_t *>(ap + kk * 16);
            bv[kk] = *reinterpret_cast<const bf16x8_t *>(bp + kk * 16);
        }
        if (t + 1 < nt) BF_STORE(st ^ 1);          // tile t+1 (landed) -> other stage
        BF_LOAD(t + 2);                            // unconditional (clamped) prefetch
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kk], bv[kk], acc, 0, 0, 0);
        __syncthreads();
    }
#undef BF_LOAD
#undef BF_STORE

    // ---- epilogue: lane -> column n, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32x32 block
    const int n = n0 + wn * 32 + (lane & 31);
    const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
    if (n >= e.n_limit) return;
    const bool live = n < e.n_true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mq = rbase + 8 * q;                 // 4 consecutive rows mq..mq+3 (mq % 4 == 0)
        float v[4];
        if constexpr (EPI == BEPI_FWD_HIDDEN) {
            uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (e.drop_thresh) {                      // same counter layout as the fp32 path (bp_kernels.h)
                const uint64_t gf = (uint64_t)(uint32_t)(mq + e.frame_off);
                const uint64_t idx = (gf >> 2) * (uint64_t)(uint32_t)e.n_true + (uint32_t)n;
                w[0] = (uint32_t)idx; w[1] = (uint32_t)(idx >> 32); w[2] = e.layer; w[3] = e.step;
                philox4x32_10(w[0], w[1], w[2], w[3], e.seed_lo, e.seed_hi);
            }
            const float bn = e.bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = act_fwd(e.act, e.alpha * acc[4 * q + j] + bn);
                if (!live || w[j] < e.drop_thresh || mq + j >= e.m_limit) y = 0.0f;
                v[j] = y;
            }
        } else if constexpr (EPI == BEPI_FWD_OUT) {
            const float bn = e.bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                float d = 0.0f;
                if (m < e.m_limit) {
                    const float o = live ? e.alpha * acc[4 * q + j] + bn : 0.0f;
                    if (e.out) e.out[(size_t)m * e.ldo + n] = o;
                    if (e.C && live) d = e.scale * (o - e.targ[(size_t)m * e.ldt + n]);      // kernSubClean
                }
                v[j] = d;
            }
            if (!e.C) continue;
        } else if constexpr (EPI == BEPI_DGRAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                v[j] = (m < e.m_limit && live) ? act_bwd(e.act, bf2f(e.yprev[(size_t)m * e.ldy + n])) * acc[4 * q + j] : 0.0f;
            }
        } else {                                      // wgrad: rows = units of layer l-1, cols = units of layer l
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mq + j;
                if (m >= e.m_limit) { v[j] = 0.0f; continue; }
                const size_t i = (size_t)m * e.ldw + n;
                if constexpr (EPI == BEPI_WGRAD_UPDATE) {
                    const float w = e.W[i];
                    const float d = e.mom * e.D[i] - e.c1 * (acc[4 * q + j] / e.ndiv + e.wc * w);   // kernUpdatedelta
                    e.D[i] = d;
                    v[j] = d + 1.0f * w;                                                            // kernAccSum
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
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mq + j < e.m_limit) e.C[(size_t)(mq + j) * e.ldc + n] = hb[j];
        const uint2 pk = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
        *reinterpret_cast<uint2 *>(e.CT + (size_t)n * e.ldct + mq) = pk;
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
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_true) return;
    float s = 0.0f;
    for (int f = 0; f < rows; ++f) s += bf2f(dx[(size_t)f * ld + n]);
    if (gout) { gout[n] = s; return; }
    const float d = mom * dbias[n] - c1 * (s / ndiv + 0.0f * bias[n]);
    dbias[n] = d;
    bias[n] = d + 1.0f * bias[n];
}
