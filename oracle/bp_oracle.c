/*
 * oracle/bp_oracle.c -- CPU restatement of the reference's frame-wise DNN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library under
 * dnn-for-speech-enhancement_amd/csrc, the python host mirror, include/BP_GPU.h) may call,
 * link or import this file.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py.
 *
 * PARITY UNPINNED for the device path: the reference tree holds no golden vectors, known-answer
 * tests or fixtures and BP_GPU.cu / DevFunc.cu cannot be built here
 * (nvcc + cuBLAS + cuRAND are absent; no stand-ins are written).  What pins this oracle instead:
 * (1) an independent numpy fp64 restatement (oracle/bp_numpy.py) and committed fixtures generated
 * from it (the .npz files under tests/golden), (2) torch float64 autograd -- of one gradient on a small net AND of
 * multi-step trajectories on 1024-wide layers with momentum, weight cost, both momentum rules, both
 * activations, injected dropout masks and the keep-scaled CV forward (tests/test_oracle.py).
 * The reference's HOST code (Interface.cc, BPtrain.cc) is a different matter: it compiles in the
 * build container against include/BP_GPU.h (oracle/Makefile, target `ref` -> oracle/_ref/), and
 * pins this repo's reader / planner / weight-file code byte for byte (tests/test_ref_pins.py).
 *
 * What it follows (all paths relative to /root/reference):
 *   BP_GPU.cu:484-673   train_bunch_single : forward, MSE backward, momentum update order
 *   BP_GPU.cu:676-773   cv_bunch_single    : forward with keep-scaled weights
 *   BP_GPU.cu:408-479   CrossValid         : bunch loop incl. partial bunch, fp32 host sum
 *   BP_GPU.cu:241-331   train              : bunch loop, partial last bunch dropped
 *   DevFunc.h:29-67     Sgemm{NN,TN,NT} operand orientation / alpha-beta swap
 *   DevFunc.cu:34-45    kernDropout  (in[i]=0 if rand[i]<p, no rescale)
 *   DevFunc.cu:67-97    kernSigmoid/kernDsigmoid (live bodies = ReLU; .bak = logistic)
 *   DevFunc.cu:166-182  kernMultiCopy (bias broadcast)
 *   DevFunc.cu:224-242  kernAccSumrow (bias gradient)
 *   DevFunc.cu:253-268  kernSubClean  ((2/rows)*(out-targ))
 *   DevFunc.cu:270-277  kernAccSum    (w = delta + w)
 *   DevFunc.cu:313-318  kernUpdatedelta (live: (1-m) rule; :306-311 commented = classic)
 *
 * Layout (same as the reference): activations [frame][unit] row-major, weights
 * [prev][cur] row-major, all fp32.  ACC selects the accumulator type of the dot products
 * (float = the reference's arithmetic class; double = tighter pin for goldens).
 *
 * Dropout: the reference draws cuRAND XORWOW uniforms seeded from time(NULL)
 * (BP_GPU.cu:69-78), so its masks are not reproducible; parity is defined as "same result
 * given the same mask".  Masks here come either from caller-supplied byte arrays or from
 * the same counter-based Philox4x32-10 stream the HIP kernels use (see bp_philox_drop).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BP_MAXLAYER 10

typedef struct {
    int   numlayers;               /* number of layer SIZES (weight layers = numlayers-1) */
    int   layersizes[BP_MAXLAYER];
    float lrate, momentum, weightcost;
    int   dropoutflag;
    float visible_omit, hid_omit;
    int   activation;              /* 0 = ReLU (live DevFunc.cu), 1 = Sigmoid (.bak)      */
    int   momentum_rule;           /* 0 = live (1-m) rule, 1 = classic (.bak)              */
    int   acc_double;              /* 0 = fp32 accumulation, 1 = fp64 accumulation         */
    uint64_t seed;                 /* Philox key for generated dropout masks               */
    int   compute_dtype;           /* 0 = fp32 (the reference).  1 = bf16 GEMM operands: weights (a
                                      rounded copy), the masked input, every stored activation and
                                      every back-propagated error are rounded to bf16 (nearest even)
                                      where the HIP path stores them as bf16; accumulation, bias,
                                      loss, master weights and the update stay fp32
                                      (BASELINE.json configs[4]; bp_bf16.h)                        */
} oracle_cfg;

/* bf16 storage rounding (round to nearest even), value returned as float */
static inline float bf16_round(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return f;
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    memcpy(&f, &u, 4);
    return f;
}
static float *bf16_copy(const float *src, size_t n)
{
    float *d = (float *)malloc(sizeof(float) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) d[i] = bf16_round(src[i]);
    return d;
}

/* Threads the OpenMP loops use (the timing leg of bench.py picks the count that is fastest on the host: on a box whose
 * cgroup grants fewer cores than it shows, 256 spinning threads are 70x slower than 16). */
void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

uint32_t bp_drop_threshold(float p)
{
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

/* Same keying as the HIP kernels: one Philox block covers 4 consecutive GLOBAL frames of one
 * unit:  counter = {lo32(idx), hi32(idx), layer, step}, idx = (gframe/4)*width + unit,
 * word = gframe%4; key = seed.  layer = index of the layer whose OUTPUT is masked
 * (0 = the visible/input layer).  Drop iff word < threshold(p). */
int bp_philox_drop(uint64_t seed, uint32_t step, uint32_t layer, uint64_t gframe,
                   uint32_t unit, uint32_t width, uint32_t thresh)
{
    uint64_t idx = (gframe >> 2) * (uint64_t)width + unit;
    uint32_t c[4] = { (uint32_t)idx, (uint32_t)(idx >> 32), layer, step };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return c[gframe & 3] < thresh;
}

/* Fill a [B][width] byte mask (1 = dropped). */
void oracle_fill_mask(const oracle_cfg *cfg, uint32_t step, uint32_t layer, uint64_t gframe0,
                      int B, int width, uint8_t *mask)
{
    float p = layer == 0 ? cfg->visible_omit : cfg->hid_omit;
    uint32_t th = bp_drop_threshold(p);
    for (int f = 0; f < B; ++f)
        for (int u = 0; u < width; ++u)
            mask[(size_t)f * width + u] =
                (uint8_t)bp_philox_drop(cfg->seed, step, layer, gframe0 + f, u, width, th);
}

/* ------------------------------------------------------------------ dense pieces */
/* x[B][cur] = alpha * (y[B][prev] . W[prev][cur]) + bias[cur]
 * DevFunc.cu:166-182 (bias broadcast) + DevFunc.h:45-55 (SgemmNN, C = A.B + C).
 * alpha = keep for the CV path (BP_GPU.cu:726-746 scales W by keep, runs the GEMM, scales
 * back), 1 for training.  In the reference keep multiplies W before the products; here it
 * multiplies each weight too so the fp32 rounding matches that order. */
/* Cache- and register-blocked tiles of the three GEMMs (fp32 mode).  Blocking only changes WHICH output elements are
 * worked on together: every output element is still one fused-multiply-add chain over its reduction index in ascending
 * order, starting from the same initial value, so the results are bit-identical to the plain loops they replace
 * (kept below for fp64 accumulation).  target_clones: AVX-512 where the host has it, the baseline ISA otherwise. */
#define TILE_F 4
#define TILE_C 64
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define BP_CLONES __attribute__((target_clones("avx512f", "default")))
#else
#define BP_CLONES
#endif

/* x[f][c0..c0+nc) for nf <= TILE_F frames: bias + sum_k y[f][k] * (W[k][c] * keep) */
static BP_CLONES void affine_tile(int prev, int cur, const float *y, int nf, const float *W, const float *bias,
                                  float keep, int c0, int nc, float *x)
{
    float acc[TILE_F][TILE_C];
    for (int i = 0; i < nf; ++i)
        for (int c = 0; c < nc; ++c) acc[i][c] = bias[c0 + c];
    if (nf == TILE_F && nc == TILE_C && keep == 1.0f) {            /* the common full tile: fixed trip counts */
        for (int k = 0; k < prev; ++k) {
            const float *wr = W + (size_t)k * cur + c0;
            for (int i = 0; i < TILE_F; ++i) {
                const float a = y[(size_t)i * prev + k];
                for (int c = 0; c < TILE_C; ++c) acc[i][c] += a * wr[c];
            }
        }
    } else {
        for (int k = 0; k < prev; ++k) {
            const float *wr = W + (size_t)k * cur + c0;
            for (int i = 0; i < nf; ++i) {
                const float a = y[(size_t)i * prev + k];
                if (keep == 1.0f) for (int c = 0; c < nc; ++c) acc[i][c] += a * wr[c];
                else              for (int c = 0; c < nc; ++c) acc[i][c] += a * (wr[c] * keep);
            }
        }
    }
    for (int i = 0; i < nf; ++i)
        for (int c = 0; c < nc; ++c) x[(size_t)i * cur + c0 + c] = acc[i][c];
}

static void affine(int B, int prev, int cur, const float *y, const float *W, const float *bias,
                   float keep, int acc_double, float *x)
{
    if (!acc_double) {
        const int nfb = (B + TILE_F - 1) / TILE_F, ncb = (cur + TILE_C - 1) / TILE_C;
        /* column panels outermost: the threads that share a weight panel (prev x 64 floats) run back to back */
#pragma omp parallel for collapse(2) schedule(static)
        for (int cb = 0; cb < ncb; ++cb)
            for (int fb = 0; fb < nfb; ++fb) {
                const int c0 = cb * TILE_C, f0 = fb * TILE_F;
                affine_tile(prev, cur, y + (size_t)f0 * prev, B - f0 < TILE_F ? B - f0 : TILE_F, W, bias, keep, c0,
                            cur - c0 < TILE_C ? cur - c0 : TILE_C, x + (size_t)f0 * cur);
            }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int f = 0; f < B; ++f) {
        const float *yr = y + (size_t)f * prev;
        float *xr = x + (size_t)f * cur;
        double *acc = (double *)malloc(sizeof(double) * cur);
        for (int c = 0; c < cur; ++c) acc[c] = bias[c];
        for (int k = 0; k < prev; ++k) {
            const double a = yr[k];
            const float *wr = W + (size_t)k * cur;
            if (keep == 1.0f) for (int c = 0; c < cur; ++c) acc[c] += a * (double)wr[c];
            else              for (int c = 0; c < cur; ++c) acc[c] += a * (double)(wr[c] * keep);
        }
        for (int c = 0; c < cur; ++c) xr[c] = (float)acc[c];
        free(acc);
    }
}

/* DevFunc.cu:67-79 live body (ReLU, strict >0) | DevFunc.cu:47-54 / .bak (logistic, expf). */
static inline float act_fwd(int activation, float x)
{
    if (activation == 0) return x > 0.0f ? x : 0.0f;
    return 1.0f / (1.0f + expf(-x));
}
/* DevFunc.cu:81-97 live body (y>0 ? 1 : 0) | :56-64 / .bak ((1-y)*y); from the OUTPUT y. */
static inline float act_bwd(int activation, float y)
{
    if (activation == 0) return y > 0.0f ? 1.0f : 0.0f;
    return (1.0f - y) * y;
}

/* dEdY_prev[B][prev] = dEdX[B][cur] . W^T   (DevFunc.h:29-43 SgemmTN, BP_GPU.cu:636) */
#define TILE_P 8
#define TILE_FV 32
/* out[f0..f0+nf)[p0..p0+np) from the transposed error panel dT[c][f]: acc[p][f] += W[p][c] * dT[c][f], c ascending */
static BP_CLONES void dgrad_tile(int B, int prev, int cur, const float *dT, const float *W, int p0, int np, int f0, int nf, float *out)
{
    float acc[TILE_P][TILE_FV];
    for (int i = 0; i < np; ++i)
        for (int f = 0; f < nf; ++f) acc[i][f] = 0.0f;
    if (np == TILE_P && nf == TILE_FV) {
        for (int c = 0; c < cur; ++c) {
            const float *dr = dT + (size_t)c * B + f0;
            for (int i = 0; i < TILE_P; ++i) {
                const float w = W[(size_t)(p0 + i) * cur + c];
                for (int f = 0; f < TILE_FV; ++f) acc[i][f] += dr[f] * w;
            }
        }
    } else {
        for (int c = 0; c < cur; ++c) {
            const float *dr = dT + (size_t)c * B + f0;
            for (int i = 0; i < np; ++i) {
                const float w = W[(size_t)(p0 + i) * cur + c];
                for (int f = 0; f < nf; ++f) acc[i][f] += dr[f] * w;
            }
        }
    }
    for (int f = 0; f < nf; ++f)
        for (int i = 0; i < np; ++i) out[(size_t)(f0 + f) * prev + p0 + i] = acc[i][f];
}

static void dgrad(int B, int prev, int cur, const float *dedx, const float *W, int acc_double,
                  float *dedy_prev)
{
    if (!acc_double) {
        float *dT = (float *)malloc(sizeof(float) * (size_t)B * cur);        /* [cur][B] */
#pragma omp parallel for schedule(static)
        for (int c = 0; c < cur; ++c)
            for (int f = 0; f < B; ++f) dT[(size_t)c * B + f] = dedx[(size_t)f * cur + c];
        const int npb = (prev + TILE_P - 1) / TILE_P, nfb = (B + TILE_FV - 1) / TILE_FV;
#pragma omp parallel for collapse(2) schedule(static)
        for (int pb = 0; pb < npb; ++pb)
            for (int fb = 0; fb < nfb; ++fb) {
                const int p0 = pb * TILE_P, f0 = fb * TILE_FV;
                dgrad_tile(B, prev, cur, dT, W, p0, prev - p0 < TILE_P ? prev - p0 : TILE_P, f0, B - f0 < TILE_FV ? B - f0 : TILE_FV, dedy_prev);
            }
        free(dT);
        return;
    }
#pragma omp parallel for schedule(static)
    for (int f = 0; f < B; ++f) {
        const float *dr = dedx + (size_t)f * cur;
        float *o = dedy_prev + (size_t)f * prev;
        for (int p = 0; p < prev; ++p) {
            const float *wr = W + (size_t)p * cur;
            double s = 0.0;
            for (int c = 0; c < cur; ++c) s += (double)dr[c] * (double)wr[c];
            o[p] = (float)s;
        }
    }
}

/* G[prev][cur] = y_prev^T . dEdX  (DevFunc.h:57-67 SgemmNT, BP_GPU.cu:642);
 * gb[cur] = sum_f dEdX[f][:]    (DevFunc.cu:224-242 kernAccSumrow with alpha=0,beta=1). */
static BP_CLONES void wgrad_tile(int B, int prev, int cur, const float *y_prev, const float *dedx, int p0, int np, int c0, int nc, float *G)
{
    float acc[TILE_F][TILE_C];
    for (int i = 0; i < np; ++i)
        for (int c = 0; c < nc; ++c) acc[i][c] = 0.0f;
    if (np == TILE_F && nc == TILE_C) {
        for (int f = 0; f < B; ++f) {
            const float *dr = dedx + (size_t)f * cur + c0, *yr = y_prev + (size_t)f * prev + p0;
            for (int i = 0; i < TILE_F; ++i) {
                const float a = yr[i];
                for (int c = 0; c < TILE_C; ++c) acc[i][c] += a * dr[c];
            }
        }
    } else {
        for (int f = 0; f < B; ++f) {
            const float *dr = dedx + (size_t)f * cur + c0, *yr = y_prev + (size_t)f * prev + p0;
            for (int i = 0; i < np; ++i) {
                const float a = yr[i];
                for (int c = 0; c < nc; ++c) acc[i][c] += a * dr[c];
            }
        }
    }
    for (int i = 0; i < np; ++i)
        for (int c = 0; c < nc; ++c) G[(size_t)(p0 + i) * cur + c0 + c] = acc[i][c];
}

static void wgrad(int B, int prev, int cur, const float *y_prev, const float *dedx,
                  int acc_double, float *G, float *gb)
{
    if (!acc_double) {
        const int npb = (prev + TILE_F - 1) / TILE_F, ncb = (cur + TILE_C - 1) / TILE_C;
#pragma omp parallel for collapse(2) schedule(static)
        for (int cb = 0; cb < ncb; ++cb)
            for (int pb = 0; pb < npb; ++pb) {
                const int p0 = pb * TILE_F, c0 = cb * TILE_C;
                wgrad_tile(B, prev, cur, y_prev, dedx, p0, prev - p0 < TILE_F ? prev - p0 : TILE_F, c0, cur - c0 < TILE_C ? cur - c0 : TILE_C, G);
            }
        for (int c = 0; c < cur; ++c) gb[c] = 0.0f;
        for (int f = 0; f < B; ++f) {                 /* kernAccSumrow: sequential over frames for every column */
            const float *dr = dedx + (size_t)f * cur;
            for (int c = 0; c < cur; ++c) gb[c] += dr[c];
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < prev; ++p) {
        float *gr = G + (size_t)p * cur;
        double *acc = (double *)calloc(cur, sizeof(double));
        for (int f = 0; f < B; ++f) {
            const double a = y_prev[(size_t)f * prev + p];
            const float *dr = dedx + (size_t)f * cur;
            for (int c = 0; c < cur; ++c) acc[c] += a * (double)dr[c];
        }
        for (int c = 0; c < cur; ++c) gr[c] = (float)acc[c];
        free(acc);
    }
    for (int c = 0; c < cur; ++c) {
        double s = 0.0;
        for (int f = 0; f < B; ++f) s += dedx[(size_t)f * cur + c];
        gb[c] = (float)s;
    }
}

/* DevFunc.cu:313-318 (live) / :306-311 (classic) then DevFunc.cu:270-277 (w = delta + w).
 * Exactly the reference's fp32 association; n is an int as in the kernel signature. */
static void update(size_t size, float *delta, float *w, const float *g, int n, float m,
                   float lr, float wc, int classic)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < size; ++i) {
        float d;
        if (classic) d = m * delta[i] - lr * (g[i] / n + wc * w[i]);
        else         d = m * delta[i] - (1 - m) * lr * (g[i] / n + wc * w[i]);
        delta[i] = d;
        w[i] = d + 1.0f * w[i];
    }
}

/* ------------------------------------------------------------------ public entry points */
static size_t act_offset(const oracle_cfg *cfg, int layer, int B)
{
    size_t o = 0;
    for (int l = 1; l < layer; ++l) o += (size_t)B * cfg->layersizes[l];
    return o;
}
size_t oracle_act_floats(const oracle_cfg *cfg, int B)
{
    return act_offset(cfg, cfg->numlayers, B);
}

/* CV / inference forward (BP_GPU.cu:676-773): no masks; if dropoutflag, weights of layer 1
 * are scaled by 1-visible_omit and of deeper layers by 1-hid_omit (bias not scaled).
 * weights[l], bias[l] for l = 1..numlayers-1 (index 0 unused, as in the reference). */
void oracle_forward(const oracle_cfg *cfg, float *const *weights, float *const *bias, int B,
                    const float *in, float *out)
{
    const int L = cfg->numlayers;
    int maxw = 0;
    for (int l = 0; l < L; ++l) if (cfg->layersizes[l] > maxw) maxw = cfg->layersizes[l];
    float *a = (float *)malloc(sizeof(float) * (size_t)B * maxw);
    float *b = (float *)malloc(sizeof(float) * (size_t)B * maxw);
    const int bf = cfg->compute_dtype == 1;
    float *inb = bf ? bf16_copy(in, (size_t)B * cfg->layersizes[0]) : NULL;
    const float *y = bf ? inb : in;
    for (int l = 1; l < L; ++l) {
        const int prev = cfg->layersizes[l - 1], cur = cfg->layersizes[l];
        float keep = 1.0f;
        if (cfg->dropoutflag == 1) keep = (l == 1) ? 1.0f - cfg->visible_omit : 1.0f - cfg->hid_omit;
        float *x = (l == L - 1) ? out : ((l & 1) ? a : b);
        float *wb = bf ? bf16_copy(weights[l], (size_t)prev * cur) : NULL;
        affine(B, prev, cur, y, bf ? wb : weights[l], bias[l], keep, cfg->acc_double, x);
        free(wb);
        if (l != L - 1) {
            const size_t n = (size_t)B * cur;
            for (size_t i = 0; i < n; ++i) x[i] = act_fwd(cfg->activation, x[i]);
            if (bf) for (size_t i = 0; i < n; ++i) x[i] = bf16_round(x[i]);
        }
        y = x;
    }
    free(a); free(b); free(inb);
}

/* CrossValid (BP_GPU.cu:408-479): bunch loop (partial bunch processed), squared error summed
 * in a host float in frame-major, bin-minor order (:458-467); returns the SUM. */
float oracle_crossvalid(const oracle_cfg *cfg, float *const *weights, float *const *bias,
                        int bunchsize, int n_frames, const float *in, const float *targ)
{
    const int s0 = cfg->layersizes[0], sL = cfg->layersizes[cfg->numlayers - 1];
    float *out = (float *)malloc(sizeof(float) * (size_t)bunchsize * sL);
    float squared_err = 0.0f;
    for (int i = 0; i < n_frames; i += bunchsize) {
        const int fb = bunchsize > n_frames - i ? n_frames - i : bunchsize;
        oracle_forward(cfg, weights, bias, fb, in + (size_t)i * s0, out);
        const float *t = targ + (size_t)i * sL;
        for (int j = 0; j < fb; ++j)
            for (int d = 0; d < sL; ++d) {
                const float e = out[(size_t)j * sL + d] - t[(size_t)j * sL + d];
                squared_err = squared_err + e * e;
            }
    }
    free(out);
    return squared_err;
}

/* Forward + backward of one bunch of B frames WITHOUT the update: fills grads_w[l]
 * ([prev][cur]) and grads_b[l].  scale_frames is the n in (2/n)*(out-targ): B for the
 * reference's single-GPU step (DevFunc.cu:263), the GLOBAL bunch size when the bunch is a
 * data-parallel shard.  masks[l] (l = 0..numlayers-2): optional [B][s_l] byte
 * arrays, 1 = drop the OUTPUT of layer l before it feeds layer l+1 (BP_GPU.cu:534-551);
 * NULL entry / NULL array = no dropout on that layer.  `in` is modified in place when
 * masks[0] is given, exactly as the reference mutates its device copy (BP_GPU.cu:539).
 * acts (optional, oracle_act_floats(cfg,B) floats): post-dropout y_l for l=1..L-2 then the
 * linear output, packed back to back.  out (optional): [B][sL]. */
void oracle_grads(const oracle_cfg *cfg, float *const *weights, float *const *bias, int B,
                  float *in, const float *targ, const uint8_t *const *masks, int scale_frames,
                  float *const *grads_w, float *const *grads_b, float *acts, float *out)
{
    const int L = cfg->numlayers;
    const int bf = cfg->compute_dtype == 1;
    float *own_acts = NULL;
    if (!acts) acts = own_acts = (float *)malloc(sizeof(float) * oracle_act_floats(cfg, B));
    float *wb[BP_MAXLAYER] = {0};              /* bf16 mode: rounded copies of the (pre-update) weights */
    float *inb = NULL;                         /* bf16 mode: rounded copy of the masked input bunch     */
    if (bf) for (int l = 1; l < L; ++l) wb[l] = bf16_copy(weights[l], (size_t)cfg->layersizes[l - 1] * cfg->layersizes[l]);
    /* ---- forward, BP_GPU.cu:518-585 */
    for (int l = 1; l < L; ++l) {
        const int prev = cfg->layersizes[l - 1], cur = cfg->layersizes[l];
        float *yprev = (l == 1) ? in : acts + act_offset(cfg, l - 1, B);
        if (masks && masks[l - 1]) {
            const uint8_t *mk = masks[l - 1];
            const size_t n = (size_t)B * prev;
            for (size_t i = 0; i < n; ++i) if (mk[i]) yprev[i] = 0.0f;  /* kernDropout */
        }
        if (bf && l == 1) { inb = bf16_copy(in, (size_t)B * prev); yprev = inb; }
        float *x = acts + act_offset(cfg, l, B);
        affine(B, prev, cur, yprev, bf ? wb[l] : weights[l], bias[l], 1.0f, cfg->acc_double, x);
        if (l != L - 1) {
            const size_t n = (size_t)B * cur;
            for (size_t i = 0; i < n; ++i) x[i] = act_fwd(cfg->activation, x[i]);
            if (bf) for (size_t i = 0; i < n; ++i) x[i] = bf16_round(x[i]);   /* stored as bf16 (zeros of the next mask stay zeros) */
        }
    }
    const int sL = cfg->layersizes[L - 1];
    const float *o = acts + act_offset(cfg, L - 1, B);
    if (out) memcpy(out, o, sizeof(float) * (size_t)B * sL);

    /* ---- backward, BP_GPU.cu:588-671 (update applied by the caller) */
    int maxw = 0;
    for (int l = 0; l < L; ++l) if (cfg->layersizes[l] > maxw) maxw = cfg->layersizes[l];
    float *dedx = (float *)malloc(sizeof(float) * (size_t)B * maxw);
    float *dedy = (float *)malloc(sizeof(float) * (size_t)B * maxw);
    for (int l = L - 1; l > 0; --l) {
        const int prev = cfg->layersizes[l - 1], cur = cfg->layersizes[l];
        const size_t n = (size_t)B * cur;
        if (l == L - 1) {
            const float s = 2.0f / scale_frames;                 /* kernSubClean */
            for (size_t i = 0; i < n; ++i) dedx[i] = s * (o[i] - targ[i]);
        } else {
            const float *y = acts + act_offset(cfg, l, B);       /* post-dropout y_l */
            for (size_t i = 0; i < n; ++i)
                dedx[i] = act_bwd(cfg->activation, y[i]) * dedy[i]; /* kernDsigmoid*kernVecMul */
        }
        if (bf) for (size_t i = 0; i < n; ++i) dedx[i] = bf16_round(dedx[i]);
        if (l != 1) dgrad(B, prev, cur, dedx, bf ? wb[l] : weights[l], cfg->acc_double, dedy);
        const float *yprev = (l == 1) ? (bf ? inb : in) : acts + act_offset(cfg, l - 1, B);
        wgrad(B, prev, cur, yprev, dedx, cfg->acc_double, grads_w[l], grads_b[l]);
    }
    free(dedx); free(dedy); free(inb);
    for (int l = 1; l < L; ++l) free(wb[l]);
    if (own_acts) free(own_acts);
}

/* kernUpdatedelta + kernAccSum for every layer; n = divisor (B, or global B under DP). */
void oracle_update(const oracle_cfg *cfg, float *const *weights, float *const *bias,
                   float *const *delta_w, float *const *delta_b, float *const *grads_w,
                   float *const *grads_b, int n)
{
    for (int l = cfg->numlayers - 1; l > 0; --l) {
        const size_t prev = cfg->layersizes[l - 1], cur = cfg->layersizes[l];
        update(prev * cur, delta_w[l], weights[l], grads_w[l], n, cfg->momentum, cfg->lrate,
               cfg->weightcost, cfg->momentum_rule);
        update(cur, delta_b[l], bias[l], grads_b[l], n, cfg->momentum, cfg->lrate, 0.0f,
               cfg->momentum_rule);
    }
}

/* train_bunch_single (BP_GPU.cu:484-673).  All dgrads of a step use pre-update weights
 * (each layer's update follows its own dgrad, :636 then :643-652), so computing every
 * gradient first and updating afterwards is arithmetically identical.
 * gen_masks != 0: dropout masks come from the Philox stream keyed (seed, step, layer,
 * gframe0+f, unit) when cfg->dropoutflag == 1. */
void oracle_train_bunch(const oracle_cfg *cfg, float *const *weights, float *const *bias,
                        float *const *delta_w, float *const *delta_b, int B, float *in,
                        const float *targ, const uint8_t *const *masks, int gen_masks,
                        uint32_t step, uint64_t gframe0)
{
    const int L = cfg->numlayers;
    float *gw[BP_MAXLAYER] = {0}, *gb[BP_MAXLAYER] = {0};
    uint8_t *own[BP_MAXLAYER] = {0};
    const uint8_t *mk[BP_MAXLAYER] = {0};
    for (int l = 1; l < L; ++l) {
        gw[l] = (float *)malloc(sizeof(float) * (size_t)cfg->layersizes[l - 1] * cfg->layersizes[l]);
        gb[l] = (float *)malloc(sizeof(float) * cfg->layersizes[l]);
    }
    if (masks) for (int l = 0; l < L - 1; ++l) mk[l] = masks[l];
    else if (gen_masks && cfg->dropoutflag == 1)
        for (int l = 0; l < L - 1; ++l) {
            own[l] = (uint8_t *)malloc((size_t)B * cfg->layersizes[l]);
            oracle_fill_mask(cfg, step, (uint32_t)l, gframe0, B, cfg->layersizes[l], own[l]);
            mk[l] = own[l];
        }
    oracle_grads(cfg, weights, bias, B, in, targ, mk, B, gw, gb, NULL, NULL);
    oracle_update(cfg, weights, bias, delta_w, delta_b, gw, gb, B);
    for (int l = 1; l < L; ++l) { free(gw[l]); free(gb[l]); }
    for (int l = 0; l < L - 1; ++l) free(own[l]);
}

/* BP_GPU::train (BP_GPU.cu:241-331): consecutive full bunches; the partial last bunch is
 * dropped (:315-318).  Returns the number of bunches trained; *step is advanced per bunch. */
int oracle_train_chunk(const oracle_cfg *cfg, float *const *weights, float *const *bias,
                       float *const *delta_w, float *const *delta_b, int bunchsize, int n_frames,
                       float *in, const float *targ, int gen_masks, uint32_t *step)
{
    const int s0 = cfg->layersizes[0], sL = cfg->layersizes[cfg->numlayers - 1];
    int n = 0;
    for (int i = 0; i + bunchsize <= n_frames; i += bunchsize, ++n) {
        oracle_train_bunch(cfg, weights, bias, delta_w, delta_b, bunchsize, in + (size_t)i * s0,
                           targ + (size_t)i * sL, NULL, gen_masks, *step, 0);
        ++*step;
    }
    return n;
}
