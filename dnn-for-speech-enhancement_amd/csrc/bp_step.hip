// bp_step.hip -- C-ABI implementation (include/bp_c_api.h), part 1 of 3: the device state of one BP_GPU replacement
// object, the chunk interface and the per-bunch launch sequence (training, CV, forward).  gfx950 only.  The handle and
// what the other two translation units use of this one: bp_handle.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "bp_handle.h"
#include "bp_kernels.h"
#include "bp_bf16.h"
#include "bp_wgrad_dma.h"
#include "bp_wgrad_dma_bf16.h"

thread_local std::string g_bp_err;

hipError_t prof_mark(bp_handle *h, int kind)
{
    StepProf *p = h->prof;
    if (!p) return hipSuccess;
    if (p->used == p->ev.size()) {
        hipEvent_t e; hipError_t er = hipEventCreate(&e);
        if (er != hipSuccess) return er;
        p->ev.push_back(e); p->kind.push_back(kind);
    }
    p->kind[p->used] = kind;
    return hipEventRecord(p->ev[p->used++], h->stream);
}

static uint32_t drop_threshold(float p)
{
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

extern "C" const char *bp_last_error(void) { return g_bp_err.c_str(); }
extern "C" int bp_abi_version(void) { return 5; }   // 4: host-driven DP split removed; bp_rdv_*, bp_dp_attach_ex (RCCL transport), bp_dp_peer_info; 5: bp_dp_handoff
extern "C" const char *bp_build_target(void) { return "gfx950"; }
extern "C" int bp_device_count(int *n)
{
    if (!n) return fail(BP_ERR_ARG, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return BP_OK;
}

int dev_alloc(bp_handle *h, float **p, size_t n_floats)
{
    n_floats += SLACK;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n_floats * sizeof(float));
    if (e != hipSuccess) return fail(BP_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(q, 0, n_floats * sizeof(float), h->stream);
    if (e != hipSuccess) return fail(BP_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (float *)q;
    return BP_OK;
}

extern "C" int bp_dp_detach(bp_handle *h);
extern "C" int bp_destroy(bp_handle *h)
{
    if (!h) return BP_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    if (h->dp) (void)bp_dp_detach(h);
    for (void *p : h->allocs) (void)hipFree(p);
    for (auto &ws : h->wset) for (auto &r : ws.r) if (r.p) (void)hipFree(r.p);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->ev_copy) (void)hipEventDestroy(h->ev_copy);
    if (h->ev_retired) (void)hipEventDestroy(h->ev_retired);
    if (h->ev_wretired) (void)hipEventDestroy(h->ev_wretired);
    if (h->host_out) (void)hipHostFree(h->host_out);
    if (h->out_chunk) (void)hipFree(h->out_chunk);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return BP_OK;
}

static int check_hyper(float, float, float, int, float, float, const char *);
static int ensure_stacked(bp_handle *h);
static int ensure_stage_tiles(bp_handle *h);
static hipError_t stage_bunch(bp_handle *h, int first, int rows, bool train);
static StageArgs stage_args(bp_handle *h, int first, int rows, bool train, int tile, uint32_t step);
static int stage_blocks(const bp_handle *h, const StageArgs &a);
static int bf_alloc(bp_handle *h, bf16_t **p, size_t n_halfs);
static hipError_t bf_shadow(bp_handle *h, int l);

// k slices of the bf16 output forward (bf_out_splits).  Measured at configs[4], us per launch: unsplit 17.7, 4 slices 13.3, 8 slices 13.8,
// 16 slices 33.3 -- the exchange of the partial tiles grows with the slice count (profiles/r06_bf16_out_split.txt)
enum { BF_OUT_KS = 4 };
static bool bf_out_splits(const bp_handle *h);
extern "C" int bp_create(const bp_config *cfg, const float *const *weights, const float *const *bias,
                         bp_handle **out)
{
    if (!cfg || !weights || !bias || !out) return fail(BP_ERR_ARG, "bp_create: null argument");
    // WorkPara holds weights[MAXLAYER-1] indexed 1..numlayers-1 (Interface.h:44-45) => L <= 9
    if (cfg->numlayers < 2 || cfg->numlayers > BP_MAXLAYER - 1)
        return fail(BP_ERR_ARG, "bp_create: numlayers must be in 2..9");
    if (cfg->bunchsize < 1) return fail(BP_ERR_ARG, "bp_create: bunchsize must be >= 1");
    if (cfg->gpu_used < 1) return fail(BP_ERR_ARG, "bp_create: gpu_used must be >= 1");  // BP_GPU.cu:20-24
    if (cfg->compute_dtype != 0 && cfg->compute_dtype != 1) return fail(BP_ERR_ARG, "bp_create: compute_dtype must be 0 (fp32) or 1 (bf16)");
    {
        const int r = check_hyper(cfg->lrate, cfg->momentum, cfg->weightcost, cfg->dropoutflag, cfg->visible_omit, cfg->hid_omit, "bp_create");
        if (r != BP_OK) return r;
    }
    for (int l = 0; l < cfg->numlayers; ++l)
        if (cfg->layersizes[l] < 1) return fail(BP_ERR_ARG, "bp_create: layer size must be >= 1");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(BP_ERR_ARG, "bp_create: device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));

    bp_handle *h = new bp_handle();
    memset((void *)&h->cfg, 0, sizeof(h->cfg));
    h->cfg = *cfg;
    h->L = cfg->numlayers;
    h->B = cfg->bunchsize;
    h->Bg = cfg->global_bunchsize > 0 ? cfg->global_bunchsize : cfg->bunchsize;
    h->cap = cfg->max_chunk_frames > 0 ? cfg->max_chunk_frames : BP_MAXCACHEFRAME;
    if (h->cap < h->B) h->cap = h->B;
    h->chunk_frames = 0;
    h->step = 0;
    h->th_vis = cfg->dropoutflag == 1 ? drop_threshold(cfg->visible_omit) : 0u;
    h->th_hid = cfg->dropoutflag == 1 ? drop_threshold(cfg->hid_omit) : 0u;
    for (int l = 0; l < h->L; ++l) { h->s[l] = cfg->layersizes[l]; h->ld[l] = pad64(h->s[l]); }
    h->own_stream = nullptr; h->host_out = nullptr; h->ev0 = h->ev1 = nullptr;
    h->in = h->targ = h->out_dev = h->grad = nullptr; h->slabs = nullptr; h->out_splits = 1;
    h->last_ms = 0.f; h->last_bunches = 0; h->dp = nullptr; h->params = h->deltas = nullptr;
    h->next_first = -1; h->pre.valid = false; h->wgen = 0; h->stage_cur = 0;
    h->bf_ks_slab = nullptr; h->bf_ks_cnt = nullptr;

#define CK(x) do { int _r = (x); if (_r != BP_OK) { std::string m = g_bp_err; bp_destroy(h); g_bp_err = m; return _r; } } while (0)
#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::string m = std::string(#x) + ": " + hipGetErrorString(_e); bp_destroy(h); return fail(BP_ERR_DEVICE, m); } } while (0)
    HK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    HK(hipEventCreate(&h->ev0));
    HK(hipEventCreate(&h->ev1));
    HK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HK(hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
    HK(hipEventCreateWithFlags(&h->ev_retired, hipEventDisableTiming));
    HK(hipEventCreateWithFlags(&h->ev_wretired, hipEventDisableTiming));
    const int L = h->L;
    const size_t Bp = (size_t)((h->B + 63) & ~63);             // bunch rows rounded up to a whole tile
    // (the stacked chunk buffers in / targ are allocated by the first stacked upload, ensure_stacked():
    // a caller that only ever hands window chunks never pays for them)
    CK(dev_alloc(h, &h->out_dev, Bp * h->ld[L - 1]));
    // narrow output layer (e.g. 2048 -> 257): too few 32x32 tiles to fill 256 CUs, so its k range is
    // split over 4 workgroups per tile that meet through partial-sum slabs (EPI_OUT_SPLIT, bp_kernels.h)
    if (h->ld[L - 1] <= 512 && h->ld[L - 2] >= 1024 && h->ld[L - 2] % 256 == 0) {
        h->out_splits = OUT_SPLITS;
        // (measured round 3: 8 / 16 k-slices with 64x64 workgroup tiles -- half the operand bytes per FLOP -- are no faster:
        // C2 step 0.2216 ms with 4 slices, 0.2204 with 8, 0.2237 with 16; the layer is launch/latency-bound)
        h->slab_stride = Bp * h->ld[L - 1];
        CK(dev_alloc(h, &h->slabs, h->slab_stride * h->out_splits));
        float *tk = nullptr;
        CK(dev_alloc(h, &tk, (Bp / 32) * (size_t)((h->ld[L - 1] + 31) / 32)));      // one ticket word per 32 x 32 tile (zeroed)
        h->out_ticket = reinterpret_cast<unsigned *>(tk);
    }
    size_t goff = 0;
    for (int l = 1; l < L; ++l) {
        const size_t nw = (size_t)h->ld[l - 1] * h->ld[l];
        CK(dev_alloc(h, &h->y[l], Bp * h->ld[l]));       // rows >= B stay zero (never stored)
        CK(dev_alloc(h, &h->dx[l], Bp * h->ld[l]));      // rows >= B stay zero: k-tail of wgrad
        h->g_off[l] = goff; h->g_cnt[l] = nw + h->ld[l]; goff += h->g_cnt[l];
    }
    h->grad_floats = goff;
    CK(dev_alloc(h, &h->params, goff));
    CK(dev_alloc(h, &h->deltas, goff));
    for (int l = 1; l < L; ++l) {
        const size_t nw = (size_t)h->ld[l - 1] * h->ld[l];
        h->W[l] = h->params + h->g_off[l]; h->b[l] = h->W[l] + nw;
        h->dW[l] = h->deltas + h->g_off[l]; h->db[l] = h->dW[l] + nw;
    }
    if (h->Bg != h->B) CK(dev_alloc(h, &h->grad, goff));   // otherwise allocated on first bp_grads_resident
    for (int l = 1; l < L; ++l) {
        if (!weights[l] || !bias[l]) { bp_destroy(h); return fail(BP_ERR_ARG, "bp_create: weights[l]/bias[l] null"); }
        HK(hipMemcpy2DAsync(h->W[l], (size_t)h->ld[l] * 4, weights[l], (size_t)h->s[l] * 4, (size_t)h->s[l] * 4,
                            h->s[l - 1], hipMemcpyHostToDevice, h->stream));
        HK(hipMemcpyAsync(h->b[l], bias[l], (size_t)h->s[l] * 4, hipMemcpyHostToDevice, h->stream));
    }
    h->bf = cfg->compute_dtype == 1;
    h->Bp = (int)Bp;
    if (h->bf) {
        for (int l = 0; l < L; ++l) {
            const size_t act = Bp * (size_t)h->ld[l];
            if (l < L - 1) { CK(bf_alloc(h, &h->yb[l], act)); CK(bf_alloc(h, &h->ybT[l], act)); }
            if (l >= 1) {
                CK(bf_alloc(h, &h->dxb[l], act)); CK(bf_alloc(h, &h->dxbT[l], act));
                const size_t nw = (size_t)h->ld[l - 1] * h->ld[l];
                CK(bf_alloc(h, &h->Wb[l], nw));
                HK(bf_shadow(h, l));
            }
        }
        // narrow output layer behind a wide one: its forward splits k over BF_OUT_KS workgroups per 32 x 64 tile (bp_bf16.h)
        const int otiles = (int)(Bp / 32) * (h->ld[L - 1] / 64);
        if (bf_out_splits(h)) {
            CK(dev_alloc(h, &h->bf_ks_slab, (size_t)otiles * BF_OUT_KS * 128 * 16));
            float *cnt = nullptr;
            CK(dev_alloc(h, &cnt, (size_t)otiles));            // (dev_alloc zeroes)
            h->bf_ks_cnt = reinterpret_cast<unsigned *>(cnt);
        }
    }
    HK(hipStreamSynchronize(h->stream));
#undef CK
#undef HK
    *out = h;
    return BP_OK;
}

// lrate / momentum / weightcost must be finite; with dropout on, the omit rates must lie in [0, 1): a rate of 1 would
// make the CV keep-scale 0 and a rate outside the range wraps the 32-bit drop threshold.
static int check_hyper(float lrate, float momentum, float weightcost, int dropoutflag, float visible_omit, float hid_omit, const char *who)
{
    if (!(lrate == lrate && momentum == momentum && weightcost == weightcost) || std::isinf(lrate) || std::isinf(momentum) || std::isinf(weightcost))
        return fail(BP_ERR_ARG, std::string(who) + ": lrate / momentum / weightcost must be finite");
    if (dropoutflag == 1 && !(visible_omit >= 0.0f && visible_omit < 1.0f && hid_omit >= 0.0f && hid_omit < 1.0f))
        return fail(BP_ERR_ARG, std::string(who) + ": visible_omit and hid_omit must be in [0, 1) when dropoutflag is 1");
    return BP_OK;
}

extern "C" int bp_set_hyper(bp_handle *h, float lrate, float momentum, float weightcost, int dropoutflag, float visible_omit,
                            float hid_omit)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    int r = check_hyper(lrate, momentum, weightcost, dropoutflag, visible_omit, hid_omit, "bp_set_hyper");
    if (r != BP_OK) return r;
    h->cfg.lrate = lrate; h->cfg.momentum = momentum; h->cfg.weightcost = weightcost;
    if (dropoutflag != h->cfg.dropoutflag || visible_omit != h->cfg.visible_omit || hid_omit != h->cfg.hid_omit) {
        HIPCHK(hipSetDevice(h->cfg.device));
        const uint32_t th_vis = dropoutflag == 1 ? drop_threshold(visible_omit) : 0u;
        if (th_vis && h->in) {                                  // stacked chunk resident: its bunches are masked into the staged tile from now on
            r = ensure_stage_tiles(h);
            if (r != BP_OK) return r;
        }
        h->cfg.dropoutflag = dropoutflag; h->cfg.visible_omit = visible_omit; h->cfg.hid_omit = hid_omit;
        h->th_vis = th_vis;
        h->pre.valid = false;                                   // (a pre-staged bunch was masked with the old rate)
        h->th_hid = dropoutflag == 1 ? drop_threshold(hid_omit) : 0u;
    }
    return BP_OK;
}

extern "C" int bp_sync(bp_handle *h)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return dp_check(h);
}

// ------------------------------------------------------------------ launches
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI, int TAG = 0>
static hipError_t launch(hipStream_t st, GemmArgs g, const EpiArgs &e, int M, int N, int max_grid = 0)
{
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    int grid = g.tiles_m * g.tiles_n;
    if (max_grid > 0 && grid > max_grid) grid = max_grid;     // persistent: workgroups loop over tiles
    hipLaunchKernelGGL((bp_gemm<BM, BN, BK, WM, WN, A_KC, B_KC, EPI, TAG>), dim3(grid), dim3(256), 0, st, g, e);
    return hipGetLastError();
}

static EpiArgs epi_zero()
{
    EpiArgs e;
    memset(&e, 0, sizeof(e));
    e.alpha = 1.0f;
    return e;
}

// forward of weight layer l on M frames.  y_prev [M][ld_{l-1}].  train: hidden outputs get the
// hid_omit mask; output layer writes dEdX_L (and out when out != null).
hipError_t launch_fwd(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, const float *targ,
                      float *out, bool train, float alpha)
{
    const int L = h->L, prev = h->ld[l - 1], cur = h->ld[l];
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = y_prev; g.lda = prev; g.B = h->W[l]; g.ldb = cur; g.K = prev;
    EpiArgs e = epi_zero();
    e.m_limit = M; e.n_limit = cur; e.n_true = h->s[l]; e.bias = h->b[l]; e.alpha = alpha; e.act = h->cfg.activation;
    if (l != L - 1) {
        e.C = h->y[l]; e.ldc = cur;
        e.drop_thresh = train ? h->th_hid : 0u;
        e.seed_lo = (uint32_t)h->cfg.seed; e.seed_hi = (uint32_t)(h->cfg.seed >> 32);
        e.step = h->step; e.layer = (uint32_t)l; e.frame_off = h->cfg.rank_frame_offset;
        if (train && h->inj_mask[l]) { e.mask = h->inj_mask[l]; e.ldmask = h->s[l]; e.drop_thresh = 1u; }   // injected mask (tests)
        if (cur <= 512) return launch<32, 32, 64, 1, 1, true, false, EPI_FWD_HIDDEN>(st, g, e, M, cur);
        if (l == 1) return launch<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1>(st, g, e, M, cur);   // (TAG 1: own name in profiles)
        return launch<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN>(st, g, e, M, cur);
    }
    e.scale = 2.0f / (float)h->Bg;                       // kernSubClean: 2.0f/rows (global rows under DP)
    if (h->out_splits > 1) {
        g.K = prev / h->out_splits; g.k_split = g.K; g.slab_stride = h->slab_stride;
        g.ks_slab = h->slabs; g.ks_ticket = h->out_ticket;
        g.tiles_m = (M + 31) / 32; g.tiles_n = (cur + 31) / 32;
        e.C = train ? h->dx[l] : nullptr; e.ldc = cur;
        e.aux = targ; e.ldaux = cur; e.aux2 = out; e.ldaux2 = cur;
        using KOut = GemmKernel<32, 32, 64, 1, 1, true, false, EPI_OUT_SPLIT>;
        const int n_gemm = g.tiles_m * g.tiles_n * OUT_SPLITS;
        StageArgs sa; memset(&sa, 0, sizeof(sa));
        int n_stage = 0;
        if (train && h->next_first >= 0 && st == h->stream) {
            // another staged bunch behind this one (window chunk, or stacked chunk with visible dropout): stack / copy (and mask, with the NEXT step's Philox position) that bunch
            // into the other tile from the spare workgroups of this launch
            const int tile = 1 - h->stage_cur;
            sa = stage_args(h, h->next_first, h->B, true, tile, h->step + 1);
            n_stage = stage_blocks(h, sa);
            h->pre.valid = true; h->pre.first = h->next_first; h->pre.tile = tile; h->pre.step = h->step + 1; h->pre.gen = h->wgen;
        }
        hipLaunchKernelGGL(bp_out_split_stage<KOut>, dim3((unsigned)(n_gemm + n_stage)), dim3(256), 0, st, g, e, n_gemm, sa);
        return hipGetLastError();
    }
    e.C = train ? h->dx[l] : nullptr; e.ldc = cur;
    e.aux = targ; e.ldaux = cur; e.aux2 = out; e.ldaux2 = cur;
    if (cur <= 512) return launch<32, 32, 64, 1, 1, true, false, EPI_FWD_OUT>(st, g, e, M, cur);
    return launch<32, 64, 64, 1, 2, true, false, EPI_FWD_OUT>(st, g, e, M, cur);
}

// A prepared backward GEMM: arguments + which tile configuration it uses.
enum { CFG_DGRAD_WIDE, CFG_DGRAD_WIDE128, CFG_DGRAD_NARROW, CFG_WGRAD };
struct Prepared { GemmArgs g; EpiArgs e; int M, N, cfg; bool fused; };

using KDgradWide = GemmKernel<32, 64, 64, 1, 2, true, true, EPI_DGRAD>;
// K (= padded width of layer l) a multiple of 128: 128-deep k-tiles halve the barrier count; one
// 96 KB workgroup per CU (22.9 vs 25.4 us for 2048x2048, tools/gemm_probe.hip)
using KDgradWide128 = GemmKernel<32, 64, 128, 1, 2, true, true, EPI_DGRAD>;
using KDgradNarrow = GemmKernel<32, 32, 64, 1, 1, true, true, EPI_DGRAD>;
// wgrad: 64x64x32 tiles, 4 waves of one 32x32 block each.  168 VGPRs => 3 workgroups per CU, which is what
// lets the prologue / W,delta round trip of one workgroup hide behind the MFMA phase of the others
// (128x64 tiles move 25 % less through L2 but fit only 2 per CU: 0.242 vs 0.230 ms per C2 step).
template <int EPI> using KWgrad = GemmKernel<64, 64, 32, 2, 2, false, false, EPI>;
// bunch of 256 frames (the benchmark configuration): LDS-DMA staged, fully unrolled kernel of bp_wgrad_dma.h; the
// register-staged, fully unrolled form it replaced (84.7 vs 80.4 us) left the tree in round 6 (git ed4dac3 has it)
// data-parallel gradient store (no W/delta to carry): 128x64x16 tiles are 136 VGPRs and measured faster there
using KWgradStore = GemmKernel<128, 64, 16, 2, 2, false, false, EPI_WGRAD_STORE>;

// dEdX_{l-1} = act'(y_{l-1}) * (dEdX_l . W_l^T)     (BP_GPU.cu:611-637)
static Prepared prep_dgrad(bp_handle *h, int l, int M)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    Prepared p; memset(&p, 0, sizeof(p));
    p.g.A = h->dx[l]; p.g.lda = cur; p.g.B = h->W[l]; p.g.ldb = cur; p.g.K = cur;
    p.e = epi_zero();
    p.e.C = h->dx[l - 1]; p.e.ldc = prev; p.e.m_limit = M; p.e.n_limit = prev; p.e.n_true = h->s[l - 1];
    p.e.aux = h->y[l - 1]; p.e.ldaux = prev; p.e.act = h->cfg.activation;
    p.M = M; p.N = prev; p.cfg = prev <= 512 ? CFG_DGRAD_NARROW : (cur % 128 == 0 ? CFG_DGRAD_WIDE128 : CFG_DGRAD_WIDE);
    return p;
}

// G_l = y_{l-1}^T . dEdX_l, gb_l = colsum(dEdX_l); fused momentum update (single device) or
// store into the flat gradient buffer (data parallel).   (BP_GPU.cu:642-652)
static Prepared prep_wgrad(bp_handle *h, int l, int M, const float *y_prev, bool fused)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    Prepared p; memset(&p, 0, sizeof(p));
    p.g.A = y_prev; p.g.lda = prev; p.g.B = h->dx[l]; p.g.ldb = cur; p.g.K = M;
    p.e = epi_zero();
    p.e.ldc = cur; p.e.m_limit = prev; p.e.n_limit = cur; p.e.n_true = h->s[l];
    if (fused) {
        const float m = h->cfg.momentum, lr = h->cfg.lrate;
        p.e.C = h->W[l]; p.e.aux2 = h->dW[l]; p.e.ldaux2 = cur;
        p.e.mom = m; p.e.c1 = h->cfg.momentum_rule == 1 ? lr : (1 - m) * lr; p.e.wc = h->cfg.weightcost;
        p.e.ndiv = (float)h->Bg;
        p.e.bias_w = h->b[l]; p.e.bias_d = h->db[l];
    } else {
        p.e.C = h->grad + h->g_off[l];
        p.e.bias_g = h->grad + h->g_off[l] + (size_t)prev * cur;
    }
    p.M = prev; p.N = cur; p.cfg = CFG_WGRAD; p.fused = fused;
    return p;
}

// n (1..4) independent problems of one tile configuration in one launch (bp_gemm_multi).
template <class K, int BMT, int BNT>
static hipError_t run_multi(hipStream_t st, Prepared *ps, int n)
{
    MultiArgs a; memset(&a, 0, sizeof(a));
    int t = 0;
    for (int i = 0; i < n; ++i) {
        ps[i].g.tiles_m = (ps[i].M + BMT - 1) / BMT; ps[i].g.tiles_n = (ps[i].N + BNT - 1) / BNT;
        a.g[i] = ps[i].g; a.e[i] = ps[i].e; a.first_tile[i] = t;
        // every problem starts on a multiple of 8 workgroups: its problem-relative block index then has the same
        // low 3 bits as the hardware's blockIdx (= the XCD), which the XCD-aware tile map inside run() relies on
        // (the up-to-7 padding workgroups find no tile and exit)
        t += (ps[i].g.tiles_m * ps[i].g.tiles_n + 7) & ~7;
    }
    a.first_tile[n] = t; a.n = n;
    hipLaunchKernelGGL((bp_gemm_multi<K>), dim3(t), dim3(256), 0, st, a);
    return hipGetLastError();
}

// The same for a plain __global__ kernel taking MultiArgs with 64x64 tiles (bp_wgrad_dma.h).
template <void (*KERNEL)(const MultiArgs)>
static hipError_t run_multi_k(hipStream_t st, Prepared *ps, int n)
{
    MultiArgs a; memset(&a, 0, sizeof(a));
    int t = 0;
    for (int i = 0; i < n; ++i) {
        ps[i].g.tiles_m = (ps[i].M + 63) / 64; ps[i].g.tiles_n = (ps[i].N + 63) / 64;
        a.g[i] = ps[i].g; a.e[i] = ps[i].e; a.first_tile[i] = t;
        t += (ps[i].g.tiles_m * ps[i].g.tiles_n + 7) & ~7;      // (problem-relative block index keeps the XCD bits, see run_multi)
    }
    a.first_tile[n] = t; a.n = n;
    hipLaunchKernelGGL(KERNEL, dim3(t), dim3(256), 0, st, a);
    return hipGetLastError();
}

// The wgrad problems ps[0..n) (all fused or all store): grouped launches of up to 4 problems.
static hipError_t run_wgrads(hipStream_t st, Prepared *ps, int n)
{
    for (int i = 0; i < n;) {
        const int m = n - i < 4 ? n - i : 4;
        // bunches of 128 / 256 / 512 frames (the shipped .pl uses 128, BASELINE.json 256 and 512): LDS-DMA kernel, unrolled
        int kk = ps[i].g.K;
        for (int j = 0; j < m; ++j) if (ps[i + j].g.K != kk) kk = 0;
        hipError_t er;
        if (kk == 256 && ps[i].fused) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 256>>(st, ps + i, m);
        else if (kk == 128 && ps[i].fused) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 128>>(st, ps + i, m);
        else if (kk == 512 && ps[i].fused) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 512>>(st, ps + i, m);
        else if (kk == 256) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 256, true>>(st, ps + i, m);       // data-parallel gradient store
        else if (kk == 128) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 128, true>>(st, ps + i, m);
        else if (kk == 512) er = run_multi_k<bp_wgrad_dma<16, 4, 4, 512, true>>(st, ps + i, m);
        else if (ps[i].fused) er = run_multi<KWgrad<EPI_WGRAD_UPDATE>, 64, 64>(st, ps + i, m);
        else er = run_multi<KWgradStore, 128, 64>(st, ps + i, m);
        if (er != hipSuccess) return er;
        i += m;
    }
    return hipSuccess;
}

hipError_t launch_dgrad(bp_handle *h, hipStream_t st, int l, int M)
{
    Prepared p = prep_dgrad(h, l, M);
    if (p.cfg == CFG_DGRAD_WIDE128) return run_multi<KDgradWide128, 32, 64>(st, &p, 1);
    return p.cfg == CFG_DGRAD_WIDE ? run_multi<KDgradWide, 32, 64>(st, &p, 1) : run_multi<KDgradNarrow, 32, 32>(st, &p, 1);
}
hipError_t launch_wgrad(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, bool fused)
{
    Prepared p = prep_wgrad(h, l, M, y_prev, fused);
    return run_wgrads(st, &p, 1);
}

// Bunches whose input rows go through the staged tile (x0s2): every bunch of a window chunk (stacked on the device, SURVEY 8f
// N3) and, with visible-layer dropout on, every bunch of a STACKED chunk -- the mask of BP_GPU.cu:536-539 is applied while the
// bunch's rows are copied into the L2-sized tile that the layer-1 forward and the layer-1 weight gradient both read.  The chunk
// itself is never modified and no masked copy of it exists.
bool step_stages(const bp_handle *h) { return h->windows || (h->th_vis != 0u && !h->inj_x0); }

// ------------------------------------------------------------------ compute_dtype == 1 (bp_bf16.h)
static int bf_alloc(bp_handle *h, bf16_t **p, size_t n_halfs)
{
    float *q = nullptr;
    const int r = dev_alloc(h, &q, (n_halfs + 1) / 2);       // zero-filled, with slack
    *p = (bf16_t *)q;
    return r;
}
static hipError_t bf_convert(bp_handle *h, const float *src, int lds, int rows, int cols, bf16_t *out, int ldo, bf16_t *outT,
                             int ldt, int rows_pad, int cols_pad)
{
    hipLaunchKernelGGL(bp_to_bf16_both, dim3((cols_pad + 31) / 32, (rows_pad + 31) / 32), dim3(32, 8), 0, h->stream, src, lds,
                       rows, cols, out, ldo, outT, ldt, rows_pad, cols_pad);
    return hipGetLastError();
}
// fp32 master weights of layer l -> the bf16 shadow (creation, data-parallel update)
static hipError_t bf_shadow(bp_handle *h, int l)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    return bf_convert(h, h->W[l], cur, prev, cur, h->Wb[l], cur, nullptr, 0, prev, cur);
}
hipError_t step_shadow(bp_handle *h, int l) { return h->bf ? bf_shadow(h, l) : hipSuccess; }
// The output layer's forward runs split over k when its 32-row tiles are few (less than one per CU), whole groups of 8, and long.
static bool bf_out_splits(const bp_handle *h)
{
    const int L = h->L, tiles = (h->Bp / 32) * (h->ld[L - 1] / 64), nt = h->ld[L - 2] / 64;
    return h->Bp % 32 == 0 && tiles % 8 == 0 && tiles * BF_OUT_KS <= 2048 && tiles < 256 && nt >= 2 * BF_OUT_KS && nt % BF_OUT_KS == 0 && (h->Bp / 64) * (h->ld[L - 1] / 64) < 512
           && !(h->Bp % 128 == 0 && (h->Bp / 128) * (h->ld[L - 1] / 64) >= 256);
}
// One tile configuration per shape: 128-row tiles while they still give every CU a workgroup (LDS-DMA staged for the
// hidden forward / dgrad), else 64-row, else 32-row tiles (128-thread workgroups).
template <int EPI, bool BKN = false>
static hipError_t bf_launch(bp_handle *h, BfGemmArgs g, const BfEpiArgs &e, int M, int N)
{
    g.tiles_n = N / 64;
    if (M % 128 == 0 && (M / 128) * g.tiles_n >= 256) {
        g.tiles_m = M / 128;
        if constexpr (EPI == BEPI_FWD_HIDDEN || EPI == BEPI_DGRAD) {              // LDS-DMA staged loop (bp_bf16.h)
            if ((g.tiles_n & 7) == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && e.ldc % 8 == 0 && e.ldct % 8 == 0 && e.n_limit == N) {
                // k-tile offset between the m-tiles that share a weight panel (bp_bf16.h).  Measured in the step, configs[4], us per
                // launch: forward (weights cold behind the update launch) 38.1 in phase, 36.3 four tiles apart, 33.7 a quarter of K
                // apart; dgrad (weights read by the forward 0.3 ms earlier) 31.4 in phase, 30.4 two tiles apart, 32.3 a quarter apart.
                // A quarter of K apart the four sharers no longer meet in L2 and the forward fetches its panel FOUR times (164 MB
                // instead of 68 MB per launch at the L2's memory side, profiles/r04_bf16_gemm_probe.txt): 2.6 us per launch are not
                // worth 2.4x the fabric traffic, so both stay within reach of each other's lines.
                g.k_rot = EPI == BEPI_FWD_HIDDEN ? 4 : 2;
#ifdef BP_DEV
                if (EPI == BEPI_FWD_HIDDEN) g.k_rot = dev_int("BP_BF16_ROT_FWD", g.k_rot);   // (tools/run_bf_alias.sh reproduces the sweep)
#endif
                hipLaunchKernelGGL((bp_gemm_bf16<EPI, 128, BKN, true>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, h->stream, g, e);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((bp_gemm_bf16<EPI, 128, BKN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, h->stream, g, e);
    } else if ((M / 64) * g.tiles_n >= 512) {
        g.tiles_m = M / 64;
        hipLaunchKernelGGL((bp_gemm_bf16<EPI, 64, BKN>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, h->stream, g, e);
    } else {
        g.tiles_m = M / 32;
        if constexpr (EPI == BEPI_FWD_OUT) {
            if (h->bf_ks_slab && M == h->Bp && N == h->ld[h->L - 1]) {
                g.ks_slab = h->bf_ks_slab; g.ks_cnt = h->bf_ks_cnt;
                hipLaunchKernelGGL((bp_gemm_bf16<EPI, 32, BKN, false, BF_OUT_KS>), dim3(g.tiles_m * g.tiles_n * BF_OUT_KS), dim3(128), 0, h->stream, g, e);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((bp_gemm_bf16<EPI, 32, BKN>), dim3(g.tiles_m * g.tiles_n), dim3(128), 0, h->stream, g, e);
    }
    return hipGetLastError();
}
// forward of weight layer l on M frames (bf16 operands); train: hidden outputs get the hid_omit mask, the output
// layer emits dEdX_L; out (fp32, [M][ld_L]) optional
static hipError_t bf_fwd(bp_handle *h, int l, int M, const float *targ, float *out, bool train, float alpha)
{
    const int L = h->L, prev = h->ld[l - 1], cur = h->ld[l];
    BfGemmArgs g; memset(&g, 0, sizeof(g));
    g.A = h->yb[l - 1]; g.lda = prev; g.B = h->Wb[l]; g.ldb = cur; g.K = prev;       // B = Wb [k = prev][n = cur]: the BKN form of the kernel
    BfEpiArgs e; memset(&e, 0, sizeof(e));
    e.m_limit = M; e.n_limit = cur; e.n_true = h->s[l]; e.bias = h->b[l]; e.alpha = alpha; e.act = h->cfg.activation;
    e.ldc = cur; e.ldct = h->Bp;
    if (l != L - 1) {
        e.C = h->yb[l]; e.CT = h->ybT[l];
        e.drop_thresh = train ? h->th_hid : 0u;
        e.seed_lo = (uint32_t)h->cfg.seed; e.seed_hi = (uint32_t)(h->cfg.seed >> 32);
        e.step = h->step; e.layer = (uint32_t)l; e.frame_off = h->cfg.rank_frame_offset;
        return bf_launch<BEPI_FWD_HIDDEN, true>(h, g, e, h->Bp, cur);
    }
    e.scale = 2.0f / (float)h->Bg;
    e.targ = targ; e.ldt = cur; e.out = out; e.ldo = cur;
    if (train) { e.C = h->dxb[l]; e.CT = h->dxbT[l]; }
    return bf_launch<BEPI_FWD_OUT, true>(h, g, e, h->Bp, cur);
}
static hipError_t bf_input(bp_handle *h, const float *x0, int M)
{
    return bf_convert(h, x0, h->ld[0], M, h->ld[0], h->yb[0], h->ld[0], h->ybT[0], h->Bp, h->Bp, h->ld[0]);
}
// dEdX_{l-1} = act'(y_{l-1}) * (dEdX_l . W_l^T), pre-update weights
static hipError_t bf_dgrad(bp_handle *h, int l)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    BfGemmArgs g; memset(&g, 0, sizeof(g));
    g.A = h->dxb[l]; g.lda = cur; g.B = h->Wb[l]; g.ldb = cur; g.K = cur;
    BfEpiArgs e; memset(&e, 0, sizeof(e));
    e.m_limit = h->B; e.n_limit = prev; e.n_true = h->s[l - 1]; e.act = h->cfg.activation;
    e.C = h->dxb[l - 1]; e.ldc = prev; e.CT = h->dxbT[l - 1]; e.ldct = h->Bp; e.yprev = h->yb[l - 1]; e.ldy = prev;
    return bf_launch<BEPI_DGRAD>(h, g, e, h->Bp, prev);
}
// G_l = y_{l-1}^T . dEdX_l  (+ fused update and shadow refresh, or store into the flat buffer), bias gradient: the
// per-layer form for bunch sizes the unrolled LDS-DMA launch below is not built for
static hipError_t bf_wgrad(bp_handle *h, int l, bool fused)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    hipError_t er;
    BfGemmArgs g; memset(&g, 0, sizeof(g));
    g.A = h->ybT[l - 1]; g.lda = h->Bp; g.B = h->dxbT[l]; g.ldb = h->Bp; g.K = h->Bp;
    BfEpiArgs e; memset(&e, 0, sizeof(e));
    e.m_limit = prev; e.n_limit = cur; e.n_true = h->s[l]; e.ldw = cur;
    const float m = h->cfg.momentum, lr = h->cfg.lrate;
    const float c1 = h->cfg.momentum_rule == 1 ? lr : (1 - m) * lr;
    if (fused) {
        e.W = h->W[l]; e.D = h->dW[l]; e.mom = m; e.c1 = c1; e.wc = h->cfg.weightcost; e.ndiv = (float)h->Bg;
        e.C = h->Wb[l]; e.ldc = cur; e.CT = nullptr; e.ldct = 0;
        er = bf_launch<BEPI_WGRAD_UPDATE>(h, g, e, prev, cur);
    } else {
        e.W = h->grad + h->g_off[l];
        er = bf_launch<BEPI_WGRAD_STORE>(h, g, e, prev, cur);
    }
    if (er != hipSuccess) return er;
    hipLaunchKernelGGL(bp_bias_bf16, dim3((h->s[l] + 63) / 64), dim3(64, 16), 0, h->stream, h->dxb[l], cur, h->B, h->s[l],
                       h->b[l], h->db[l], fused ? (float *)nullptr : h->grad + h->g_off[l] + (size_t)prev * cur, m, c1,
                       (float)h->Bg);
    return hipGetLastError();
}
// The LDS-DMA wgrad of bp_wgrad_dma_bf16.h: static bunch sizes, layers ls[0..n) in one grouped launch (bias gradient
// fused); other bunch sizes keep bf_wgrad (GEMM kernel + bias kernel per layer).
static bool bf_dma_ok(const bp_handle *h) { return h->Bp == 128 || h->Bp == 256 || h->Bp == 512 || h->Bp == 1024; }
static hipError_t bf_wgrads_dma(bp_handle *h, const int *ls, int n, bool fused)
{
    const float m = h->cfg.momentum, lr = h->cfg.lrate;
    const float c1 = h->cfg.momentum_rule == 1 ? lr : (1 - m) * lr;
    for (int i0 = 0; i0 < n; i0 += BF_WGRAD_MAXP) {
        BfWgradMulti a; memset(&a, 0, sizeof(a));
        const int cnt = n - i0 < BF_WGRAD_MAXP ? n - i0 : BF_WGRAD_MAXP;
        int t = 0;
        for (int i = 0; i < cnt; ++i) {
            const int l = ls[i0 + i], prev = h->ld[l - 1], cur = h->ld[l];
            BfWgradProblem &p = a.p[i];
            p.A = h->ybT[l - 1]; p.B = h->dxbT[l]; p.ldk = h->Bp;
            p.tiles_m = prev / 64; p.tiles_n = cur / 64;
            p.e = epi_zero();
            p.e.ldc = cur; p.e.m_limit = prev; p.e.n_limit = cur; p.e.n_true = h->s[l];
            if (fused) {
                p.e.C = h->W[l]; p.e.aux2 = h->dW[l]; p.e.ldaux2 = cur;
                p.e.mom = m; p.e.c1 = c1; p.e.wc = h->cfg.weightcost; p.e.ndiv = (float)h->Bg;
                p.e.bias_w = h->b[l]; p.e.bias_d = h->db[l];
                p.Wb = h->Wb[l]; p.ldwb = cur;
            } else {
                p.e.C = h->grad + h->g_off[l];
                p.e.bias_g = h->grad + h->g_off[l] + (size_t)prev * cur;
            }
            a.first_tile[i] = t;
            t += (p.tiles_m * p.tiles_n + 7) & ~7;             // (problem-relative block index keeps the XCD bits, see run_multi)
        }
        a.first_tile[cnt] = t; a.n = cnt;
        // fused update: six waves (two of them own W / delta), 64-frame k-tiles in a ring of 3 when the bunch has at least 4;
        // data-parallel gradient store: the four-wave loop alone
#define BF_DMA_LAUNCH(K)                                                                                             \
        do { if (fused) hipLaunchKernelGGL((bp_wgrad_dma_bf16_six<K, (K >= 256 ? 64 : 32), (K >= 256 ? 3 : 4)>), dim3(t), dim3(384), 0, h->stream, a); \
             else hipLaunchKernelGGL((bp_wgrad_dma_bf16_store<K>), dim3(t), dim3(256), 0, h->stream, a); } while (0)
        switch (h->Bp) {
        case 128: BF_DMA_LAUNCH(128); break;
        case 256: BF_DMA_LAUNCH(256); break;
        case 512: BF_DMA_LAUNCH(512); break;
        default: BF_DMA_LAUNCH(1024); break;
        }
#undef BF_DMA_LAUNCH
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) return er;
    }
    return hipSuccess;
}
// (measured round 3: 128x64 workgroup tiles -- 25 % fewer operand bytes through L2 -> LDS, 3 workgroups per CU -- are no
// faster than these 64x64 ones on the configs[4] shape: 0.846 vs 0.835 ms per step)
static hipError_t bf_wgrads(bp_handle *h, const int *ls, int n, bool fused)
{
    if (bf_dma_ok(h)) return bf_wgrads_dma(h, ls, n, fused);
    for (int i = 0; i < n; ++i) { const hipError_t er = bf_wgrad(h, ls[i], fused); if (er != hipSuccess) return er; }
    return hipSuccess;
}

// ------------------------------------------------------------------ the pieces of one bunch
// Where the rows of the bunch starting at chunk frame `first` lie: window chunks are stacked (and masked) into the staging
// tile now, stacked chunks are read in place or from their masked copy.
hipError_t step_inputs(bp_handle *h, int first, const float **x0, const float **tg)
{
    const int L = h->L, B = h->B;
    if (h->windows) {
        const hipError_t er = stage_bunch(h, first, B, true);
        *x0 = h->x0s; *tg = h->tgs;
        return er;
    }
    *x0 = h->in + (size_t)first * h->ld[0];
    *tg = h->targ + (size_t)first * h->ld[L - 1];
    if (h->inj_x0) *x0 = h->inj_x0;                     // bp_train_resident_masked: input rows with the injected visible mask
    else if (step_stages(h)) {                          // visible-layer dropout: the bunch's rows, masked, in the staged tile
        const hipError_t er = stage_bunch(h, first, B, true);
        *x0 = h->x0s;
        return er;
    }
    return hipSuccess;
}
hipError_t step_forward(bp_handle *h, int l, const float *x0, const float *tg)
{
    if (h->bf) {
        if (l == 1) { const hipError_t er = bf_input(h, x0, h->B); if (er != hipSuccess) return er; }
        return bf_fwd(h, l, h->B, tg, nullptr, true, 1.0f);
    }
    return launch_fwd(h, h->stream, l, h->B, l == 1 ? x0 : h->y[l - 1], tg, nullptr, true, 1.0f);
}
hipError_t step_dgrad(bp_handle *h, int l) { return h->bf ? bf_dgrad(h, l) : launch_dgrad(h, h->stream, l, h->B); }
// (fp32: the LDS-DMA store kernel of the static bunch sizes is the one that counts its tiles, up to 4 layers per launch)
bool step_wgrads_count(const bp_handle *h) { return !h->bf && (h->B == 128 || h->B == 256 || h->B == 512) && h->L - 1 <= 4; }
unsigned step_wgrad_tiles(const bp_handle *h, int l) { return (unsigned)(((h->ld[l - 1] + 63) / 64) * ((h->ld[l] + 63) / 64)); }
hipError_t step_wgrads_store(bp_handle *h, const int *ls, int n, const float *x0, unsigned *const *done)
{
    if (h->bf) return bf_wgrads(h, ls, n, false);
    Prepared ws[BP_MAXLAYER];
    for (int i = 0; i < n; ++i) {
        ws[i] = prep_wgrad(h, ls[i], h->B, ls[i] == 1 ? x0 : h->y[ls[i] - 1], false);
        if (done) ws[i].e.done = done[ls[i]];
    }
    return run_wgrads(h->stream, ws, n);
}

// One bunch starting at chunk frame `first`: forward + backward.  fused: momentum update inside
// the wgrad epilogues (train_bunch_single); else gradients to the flat buffer.  Everything is
// enqueued on one stream in the reference's order (BP_GPU.cu:518-671); every dgrad of the step
// sees pre-update weights because wgrad+update(l) always follows dgrad(l).
hipError_t bunch(bp_handle *h, int first, bool fused)
{
    const int L = h->L, B = h->B;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    const float *x0, *tg;
    CKE(step_inputs(h, first, &x0, &tg));
    int ls[BP_MAXLAYER];
    for (int l = 1; l < L; ++l) ls[l - 1] = l;          // wgrad problems: layer 1 (the largest) first
    if (h->bf) {
        for (int l = 1; l < L; ++l) CKE(step_forward(h, l, x0, tg));
        for (int l = L - 1; l >= 2; --l) CKE(bf_dgrad(h, l));   // every dgrad sees pre-update (shadow) weights
        return bf_wgrads(h, ls, L - 1, fused);
    }
    for (int l = 1; l < L; ++l) {
        CKE(launch_fwd(h, h->stream, l, B, l == 1 ? x0 : h->y[l - 1], tg, nullptr, true, 1.0f));
        CKE(prof_mark(h, l == 1 ? BP_PROF_FWD_L1 : (l == L - 1 ? BP_PROF_FWD_OUT : BP_PROF_FWD_HIDDEN)));
    }
    // Every dgrad of the step reads pre-update weights (BP_GPU.cu:636 runs before :643-652 of the same
    // layer and the lower layers' updates come later), so the wgrad+update problems can all wait until
    // the last dgrad and share grouped launches (bp_gemm_multi).
    for (int l = L - 1; l >= 2; --l) { CKE(launch_dgrad(h, h->stream, l, B)); CKE(prof_mark(h, l == L - 1 ? BP_PROF_DGRAD_OUT : BP_PROF_DGRAD_HIDDEN)); }
    Prepared ws[BP_MAXLAYER];
    for (int l = 1; l < L; ++l) ws[l - 1] = prep_wgrad(h, l, B, l == 1 ? x0 : h->y[l - 1], fused);
    CKE(run_wgrads(h->stream, ws, L - 1));
    CKE(prof_mark(h, BP_PROF_WGRAD));
#undef CKE
    return hipSuccess;
}

// ------------------------------------------------------------------ chunk interface
extern "C" int bp_upload_chunk(bp_handle *h, int n_frames, const float *in, const float *targ)
{
    if (!h || !in) return fail(BP_ERR_ARG, "bp_upload_chunk: null argument");
    if (n_frames < 0 || n_frames > h->cap) return fail(BP_ERR_ARG, "bp_upload_chunk: n_frames exceeds chunk capacity");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int L = h->L;
    { const int r = ensure_stacked(h); if (r != BP_OK) return r; }
    h->windows = false;
    if (n_frames > 0) {
        // into the buffer pair that is not current, on the copy stream: the bunches of the previous chunk (still
        // running on the main stream out of the current pair) overlap this upload
        if (!h->in_alt) {
            const size_t capp = (size_t)h->cap + 64;
            int r;
            if ((r = dev_alloc(h, &h->in_alt, capp * h->ld[0])) != BP_OK || (r = dev_alloc(h, &h->targ_alt, capp * h->ld[L - 1])) != BP_OK)
                return r;
            HIPCHK(hipStreamSynchronize(h->stream));            // (dev_alloc zero-fills on the main stream)
        }
        if (h->retired_valid) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_retired, 0));
        HIPCHK(hipMemcpy2DAsync(h->in_alt, (size_t)h->ld[0] * 4, in, (size_t)h->s[0] * 4, (size_t)h->s[0] * 4, n_frames,
                                hipMemcpyHostToDevice, h->copy_stream));
        if (targ)
            HIPCHK(hipMemcpy2DAsync(h->targ_alt, (size_t)h->ld[L - 1] * 4, targ, (size_t)h->s[L - 1] * 4,
                                    (size_t)h->s[L - 1] * 4, n_frames, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(hipEventRecord(h->ev_copy, h->copy_stream));
        // the caller may overwrite in/targ as soon as we return (BPtrain.cc:50-53)
        HIPCHK(hipStreamSynchronize(h->copy_stream));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy, 0));
        HIPCHK(hipEventRecord(h->ev_retired, h->stream));       // everything queued so far read the old pair
        h->retired_valid = true;
        std::swap(h->in, h->in_alt);
        std::swap(h->targ, h->targ_alt);
    }
    h->chunk_frames = n_frames;
    h->wgen++; h->pre.valid = false; h->next_first = -1;    // (a tile pre-staged from the old chunk is void)
    return BP_OK;
}

// ---- on-device frame stacking (host counterpart: Interface.cc:757-797)
static int raw_reserve(bp_handle *h, int set, int which, size_t bytes)
{
    bp_handle::Raw &r = h->wset[set].r[which];
    if (bytes <= r.bytes) return BP_OK;
    if (r.p) {
        HIPCHK(hipStreamSynchronize(h->copy_stream)); HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipFree(r.p); r.p = nullptr; r.bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&r.p, want);
    if (e != hipSuccess) return fail(BP_ERR_NOMEM, std::string("hipMalloc (window staging): ") + hipGetErrorString(e));
    r.bytes = want;
    return BP_OK;
}

// The stacked chunk buffers, on the first stacked upload (window chunks never need them).
static int ensure_stacked(bp_handle *h)
{
    const size_t capp = (size_t)h->cap + 64;
    bool fresh = false;
    int r;
    if (!h->in) { if ((r = dev_alloc(h, &h->in, capp * h->ld[0])) != BP_OK) return r; fresh = true; }
    if (!h->targ) { if ((r = dev_alloc(h, &h->targ, capp * h->ld[h->L - 1])) != BP_OK) return r; fresh = true; }
    if (fresh) HIPCHK(hipStreamSynchronize(h->stream));         // (dev_alloc zero-fills on the main stream)
    return h->th_vis ? ensure_stage_tiles(h) : BP_OK;
}

// The two staged bunch tiles [Bp][ld_0] (+ target tiles for window chunks), on first need.
static int ensure_stage_tiles(bp_handle *h)
{
    if (h->x0s) return BP_OK;
    int r;
    for (int k = 0; k < 2; ++k)
        if ((r = dev_alloc(h, &h->x0s2[k], (size_t)h->Bp * h->ld[0])) != BP_OK || (r = dev_alloc(h, &h->tgs2[k], (size_t)h->Bp * h->ld[h->L - 1])) != BP_OK)
            return r;
    h->stage_cur = 0; h->x0s = h->x0s2[0]; h->tgs = h->tgs2[0];
    HIPCHK(hipStreamSynchronize(h->stream));                    // (dev_alloc zero-fills on the main stream)
    return BP_OK;
}

// Stack rows [first, first+rows) of the resident window chunk into the bunch tile (x0s, and tgs when the chunk carries
// targets); train: with the visible-layer dropout of this step.
static StageArgs stage_args(bp_handle *h, int first, int rows, bool train, int tile, uint32_t step)
{
    const int L = h->L, ld0 = h->ld[0], ldL = h->ld[L - 1];
    StageArgs a; memset(&a, 0, sizeof(a));
    a.x = h->x0s2[tile]; a.ld = ld0; a.width = h->s[0];
    a.rows = rows; a.thresh = train ? h->th_vis : 0u; a.frame_off = h->cfg.rank_frame_offset;
    a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32); a.step = step;
    a.nbx = (rows + 3) / 4;
    if (!h->windows) {
        // stacked chunk: rows [first, first+rows) as they lie (no tables: win_start == null selects the row-copy form, one thread =
        // 4 rows x 4 columns), targets are read in place
        a.fea = h->in + (size_t)first * ld0; a.fea_dim = ld0; a.win = h->s[0];
        a.yb_in = (ld0 / 4 + 255) / 256;
        return a;
    }
    a.fea = h->wv.fea; a.fea_dim = h->wv.D; a.win = h->wv.win; a.nat = h->wv.nat;
    a.win_start = h->wv.ws + first; a.nat_row = h->wv.nr ? h->wv.nr + first : (const int *)nullptr;
    a.t = h->wv.tg ? h->tgs2[tile] : (float *)nullptr; a.ldt = ldL; a.twidth = h->s[L - 1];
    a.targ_frames = h->wv.tg; a.targ_frame = h->wv.tf ? h->wv.tf + first : (const int *)nullptr;
    a.yb_in = (ld0 + 255) / 256;
    return a;
}
static int stage_blocks(const bp_handle *h, const StageArgs &a) { return a.nbx * (a.yb_in + (a.t ? (h->ld[h->L - 1] + 255) / 256 : 0)); }

static hipError_t stage_bunch(bp_handle *h, int first, int rows, bool train)
{
    // the bunch may already sit in the other tile: stacked by the previous bunch's output-layer reduce launch (same chunk, same
    // dropout stream position)
    if (train && h->pre.valid && h->pre.first == first && h->pre.step == h->step && h->pre.gen == h->wgen && rows == h->B) {
        h->stage_cur = h->pre.tile; h->pre.valid = false;
        h->x0s = h->x0s2[h->stage_cur]; h->tgs = h->tgs2[h->stage_cur];
        return hipSuccess;
    }
    h->pre.valid = false;
    h->x0s = h->x0s2[h->stage_cur]; h->tgs = h->tgs2[h->stage_cur];
    const StageArgs a = stage_args(h, first, rows, train, h->stage_cur, h->step);
    hipLaunchKernelGGL(bp_stage_bunch, dim3((unsigned)stage_blocks(h, a)), dim3(256), 0, h->stream, a);
    return hipGetLastError();
}

static int upload_windows(bp_handle *h, const bp_window_chunk *c, bool with_targ, const char *who)
{
    if (!h || !c) return fail(BP_ERR_ARG, std::string(who) + ": null argument");
    const int L = h->L, n = c->n_samples, D = c->fea_dim, ctx = c->context, sL = h->s[L - 1];
    if (n < 0 || n > h->cap) return fail(BP_ERR_ARG, std::string(who) + ": n_samples exceeds chunk capacity");
    if (D < 1 || ctx < 1 || c->n_frames < 0) return fail(BP_ERR_ARG, std::string(who) + ": bad fea_dim / context / n_frames");
    const bool nat = c->nat != nullptr;
    // (a chunk may legitimately hold 0 samples -- the planner's last chunk, Interface.cc:607-614 -- and then carries no tables)
    if (n > 0 && (long)ctx * D + (nat ? D : 0) != (long)h->s[0])
        return fail(BP_ERR_ARG, std::string(who) + ": layersizes[0] != context*fea_dim (+ fea_dim with a NAT block)");
    if (n > 0 && (!c->fea || !c->win_start || (with_targ && (!c->targ_frames || !c->targ_frame)) ||
                  (nat && (!c->nat_row || c->n_nat < 1))))
        return fail(BP_ERR_ARG, std::string(who) + ": null table");
    for (int i = 0; i < n; ++i) {
        if (c->win_start[i] < 0 || c->win_start[i] + ctx > c->n_frames)
            return fail(BP_ERR_ARG, std::string(who) + ": win_start out of range");
        if (with_targ && (c->targ_frame[i] < 0 || c->targ_frame[i] >= c->n_frames))
            return fail(BP_ERR_ARG, std::string(who) + ": targ_frame out of range");
        if (nat && (c->nat_row[i] < 0 || c->nat_row[i] >= c->n_nat))
            return fail(BP_ERR_ARG, std::string(who) + ": nat_row out of range");
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    { const int r = ensure_stage_tiles(h); if (r != BP_OK) return r; }
    if (n > 0) {
        const size_t fea_b = (size_t)c->n_frames * D * 4, tg_b = with_targ ? (size_t)c->n_frames * sL * 4 : 0;
        const size_t nat_b = nat ? (size_t)c->n_nat * D * 4 : 0, idx_b = (size_t)n * 4;
        // into the staging set that is not current, on the copy stream: the bunches of the previous chunk (still reading
        // the current set on the main stream) overlap this upload
        const int set = 1 - h->wcur;                            // (always alternate: ev_wretired covers exactly the other set)
        int r;
        if ((r = raw_reserve(h, set, 0, fea_b)) != BP_OK || (r = raw_reserve(h, set, 1, tg_b)) != BP_OK ||
            (r = raw_reserve(h, set, 2, nat_b)) != BP_OK || (r = raw_reserve(h, set, 3, 3 * idx_b)) != BP_OK)
            return r;
        bp_handle::Raw *rw = h->wset[set].r;
        float *d_fea = (float *)rw[0].p, *d_tg = (float *)rw[1].p, *d_nat = (float *)rw[2].p;
        int *d_ws = (int *)rw[3].p, *d_tf = d_ws + n, *d_nr = d_tf + n;
        hipStream_t cs = h->copy_stream;
        if (h->wretired_valid) HIPCHK(hipStreamWaitEvent(cs, h->ev_wretired, 0));
        HIPCHK(hipMemcpyAsync(d_fea, c->fea, fea_b, hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemcpyAsync(d_ws, c->win_start, idx_b, hipMemcpyHostToDevice, cs));
        if (nat) {
            HIPCHK(hipMemcpyAsync(d_nat, c->nat, nat_b, hipMemcpyHostToDevice, cs));
            HIPCHK(hipMemcpyAsync(d_nr, c->nat_row, idx_b, hipMemcpyHostToDevice, cs));
        }
        if (with_targ) {
            HIPCHK(hipMemcpyAsync(d_tg, c->targ_frames, tg_b, hipMemcpyHostToDevice, cs));
            HIPCHK(hipMemcpyAsync(d_tf, c->targ_frame, idx_b, hipMemcpyHostToDevice, cs));
        }
        HIPCHK(hipEventRecord(h->ev_copy, cs));
        HIPCHK(hipStreamSynchronize(cs));                       // the caller may overwrite its buffers as soon as we return
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copy, 0));
        HIPCHK(hipEventRecord(h->ev_wretired, h->stream));      // everything queued so far read the other set
        h->wretired_valid = true;
        h->wcur = set;
        h->wv.fea = d_fea; h->wv.tg = with_targ ? d_tg : nullptr; h->wv.nat = nat ? d_nat : nullptr;
        h->wv.ws = d_ws; h->wv.tf = with_targ ? d_tf : nullptr; h->wv.nr = nat ? d_nr : nullptr;
        h->wv.D = D; h->wv.win = ctx * D;
    }
    h->windows = true;
    h->wgen++; h->pre.valid = false; h->next_first = -1;
    h->chunk_frames = n;
    return BP_OK;
}

extern "C" int bp_upload_chunk_windows(bp_handle *h, const bp_window_chunk *c)
{
    return upload_windows(h, c, true, "bp_upload_chunk_windows");
}

extern "C" int bp_fill_chunk_synthetic(bp_handle *h, int n_frames, uint64_t seed)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    if (n_frames < 0 || n_frames > h->cap) return fail(BP_ERR_ARG, "bp_fill_chunk_synthetic: n_frames exceeds capacity");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int L = h->L;
    { const int r = ensure_stacked(h); if (r != BP_OK) return r; }
    h->windows = false;
    if (n_frames > 0) {
        size_t n4 = (size_t)n_frames * (h->ld[0] / 4);
        hipLaunchKernelGGL(bp_fill_normal, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, h->stream, h->in, h->ld[0],
                           h->s[0], n_frames, (uint32_t)seed, (uint32_t)(seed >> 32), 0u);
        HIPCHK(hipGetLastError());
        n4 = (size_t)n_frames * (h->ld[L - 1] / 4);
        hipLaunchKernelGGL(bp_fill_normal, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, h->stream, h->targ,
                           h->ld[L - 1], h->s[L - 1], n_frames, (uint32_t)seed, (uint32_t)(seed >> 32), 1u);
        HIPCHK(hipGetLastError());
    }
    h->chunk_frames = n_frames;
    h->wgen++; h->pre.valid = false; h->next_first = -1;
    return BP_OK;
}

extern "C" int bp_train_resident(bp_handle *h, int first_frame, int n_frames)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    if (first_frame < 0 || n_frames < 0 || first_frame + n_frames > h->chunk_frames)
        return fail(BP_ERR_ARG, "bp_train_resident: frame range outside the resident chunk");
    if (h->Bg != h->B && !h->dp)
        return fail(BP_ERR_STATE, "bp_train_resident: data-parallel handle (global_bunchsize != bunchsize): attach it to its group first "
                                  "(bp_dp_attach: the exchange and the sharded update run inside the library)");
    if (h->windows && n_frames >= h->B && !h->wv.tg)
        return fail(BP_ERR_STATE, "bp_train_resident: the resident window chunk was uploaded without targets (forward / CV upload)");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int nb = n_frames / h->B;          // partial last bunch ignored (BP_GPU.cu:315-318)
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipError_t er = hipSuccess;
    for (int i = 0; i < nb && er == hipSuccess; ++i) {
        h->next_first = (step_stages(h) && !h->bf && i + 1 < nb) ? first_frame + (i + 1) * h->B : -1;
        er = h->dp ? dp_bunch(h, first_frame + i * h->B) : bunch(h, first_frame + i * h->B, true);
        if (er == hipSuccess) h->step++;
    }
    h->next_first = -1;                      // (also on the error path: a stale index would stage rows of a later, smaller chunk)
    if (er != hipSuccess) {
        h->pre.valid = false;
        return fail(BP_ERR_DEVICE, std::string("bp_train_resident: ") + hipGetErrorString(er));
    }
    if (h->dp && nb > 0) HIPCHK(dp_flush(h));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->last_bunches = nb;
    return BP_OK;
}

extern "C" int bp_last_train_ms(bp_handle *h, float *ms, int *bunches)
{
    if (!h || !ms) return fail(BP_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    if (bunches) *bunches = h->last_bunches;
    return BP_OK;
}

extern "C" int bp_train_chunk(bp_handle *h, int n_frames, const float *in, const float *targ)
{
    if (!h || !in || !targ) return fail(BP_ERR_ARG, "bp_train_chunk: null argument");
    int r = bp_upload_chunk(h, n_frames, in, targ);
    if (r != BP_OK) return r;
    if (n_frames % h->B)
        printf("this bunch has only %d samples and is ignored.\n", n_frames % h->B);   // BP_GPU.cu:317
    return bp_train_resident(h, 0, n_frames);
}

extern "C" int bp_train_chunk_windows(bp_handle *h, const bp_window_chunk *c)
{
    int r = upload_windows(h, c, true, "bp_train_chunk_windows");
    if (r != BP_OK) return r;
    if (c->n_samples % h->B)
        printf("this bunch has only %d samples and is ignored.\n", c->n_samples % h->B);   // BP_GPU.cu:317
    return bp_train_resident(h, 0, c->n_samples);
}


// Parity-test entry (no reference counterpart; the reference's masks come from cuRAND seeded by time(NULL),
// BP_GPU.cu:77-78,534-551): the bunch loop of bp_train_resident with CALLER-SUPPLIED dropout masks instead of the
// Philox stream, so that fixtures with stored masks (tests/golden/*dropout*.npz, computed in fp64) can be trained
// on the device.  masks[l], l = 0..numlayers-2: host array [n_frames][layersizes[l]] of bytes, 1 = drop the output
// of layer l for that frame (l = 0: the input frame), or NULL for no dropout on that layer.  fp32 single-device only.
extern "C" int bp_train_resident_masked(bp_handle *h, int first_frame, int n_frames, const uint8_t *const *masks)
{
    if (!h || !masks) return fail(BP_ERR_ARG, "bp_train_resident_masked: null argument");
    if (h->bf || h->dp || h->Bg != h->B) return fail(BP_ERR_STATE, "bp_train_resident_masked: fp32 single-device handles only");
    if (h->windows) return fail(BP_ERR_STATE, "bp_train_resident_masked: stacked chunks only (bp_upload_chunk)");
    if (first_frame < 0 || n_frames < 0 || first_frame + n_frames > h->chunk_frames)
        return fail(BP_ERR_ARG, "bp_train_resident_masked: frame range outside the resident chunk");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int L = h->L, B = h->B, nb = n_frames / B;
    uint8_t *dm[BP_MAXLAYER] = {nullptr};
    float *xm = nullptr;
    int rc = BP_OK;
    hipError_t er = hipSuccess;
    for (int l = 0; l < L - 1 && er == hipSuccess; ++l) {
        if (!masks[l]) continue;
        const size_t bytes = (size_t)nb * B * h->s[l];
        er = hipMalloc((void **)&dm[l], bytes ? bytes : 1);
        if (er == hipSuccess) er = hipMemcpyAsync(dm[l], masks[l], bytes, hipMemcpyHostToDevice, h->stream);
    }
    if (er == hipSuccess && dm[0]) er = hipMalloc((void **)&xm, ((size_t)B + 64) * h->ld[0] * sizeof(float) + SLACK * sizeof(float));
    if (er == hipSuccess && xm) er = hipMemsetAsync(xm, 0, ((size_t)B + 64) * h->ld[0] * sizeof(float) + SLACK * sizeof(float), h->stream);
    for (int i = 0; i < nb && er == hipSuccess; ++i) {
        const int first = first_frame + i * B;
        if (dm[0]) {
            hipLaunchKernelGGL(bp_apply_mask, dim3((h->ld[0] + 255) / 256, B), dim3(256), 0, h->stream, h->in + (size_t)first * h->ld[0], xm,
                               h->ld[0], h->s[0], dm[0] + (size_t)i * B * h->s[0], B);
            er = hipGetLastError();
            h->inj_x0 = xm;
        }
        for (int l = 1; l < L - 1; ++l) h->inj_mask[l] = dm[l] ? dm[l] + (size_t)i * B * h->s[l] : nullptr;
        // with injected masks the Philox thresholds must stay out of the way: layers without a mask get no dropout
        const uint32_t th_hid = h->th_hid, th_vis = h->th_vis;
        h->th_hid = 0u; h->th_vis = 0u;
        if (er == hipSuccess) er = bunch(h, first, true);
        h->th_hid = th_hid; h->th_vis = th_vis;
        h->step++;
        if (er == hipSuccess) er = hipStreamSynchronize(h->stream);     // (xm is reused by the next bunch)
    }
    h->inj_x0 = nullptr;
    for (int l = 0; l < BP_MAXLAYER; ++l) h->inj_mask[l] = nullptr;
    if (er != hipSuccess) rc = fail(BP_ERR_DEVICE, std::string("bp_train_resident_masked: ") + hipGetErrorString(er));
    (void)hipStreamSynchronize(h->stream);
    for (auto p : dm) if (p) (void)hipFree(p);
    if (xm) (void)hipFree(xm);
    return rc;
}

// ------------------------------------------------------------------ gradients without the update (parity tests)
// forward + backward of ONE local bunch with the weight gradients stored into the flat buffer [W_1|b_1|W_2|b_2|...]
// instead of being applied: the kernels of the data-parallel step (wgrad "store" form), exposed so that a test can
// compare the gradient itself with the oracle's.  State (weights, momentum, step counter) is untouched.
extern "C" int bp_grads_resident(bp_handle *h, int first_frame)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    if (first_frame < 0 || first_frame + h->B > h->chunk_frames)
        return fail(BP_ERR_ARG, "bp_grads_resident: bunch outside the resident chunk");
    if (h->dp) return fail(BP_ERR_STATE, "bp_grads_resident: not on an attached handle (the exchange owns the gradient buffer)");
    //  a window chunk uploaded for forward / CV carries no targets: bunch() would stage only the input rows and
    // back-propagate against whatever an earlier bunch left in the staged target tile
    if (h->windows && !h->wv.tg)
        return fail(BP_ERR_STATE, "bp_grads_resident: the resident window chunk was uploaded without targets (forward / CV upload)");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (!h->grad) { int r = dev_alloc(h, &h->grad, h->grad_floats); if (r != BP_OK) return r; }
    h->next_first = -1;
    HIPCHK(bunch(h, first_frame, false));
    return BP_OK;
}
extern "C" int bp_grad_floats(bp_handle *h, size_t *n_floats)
{
    if (!h || !n_floats) return fail(BP_ERR_ARG, "null argument");
    *n_floats = h->grad_floats;
    return BP_OK;
}
extern "C" int bp_grad_layout(bp_handle *h, int layer, size_t *offset, size_t *count)
{
    if (!h || layer < 1 || layer >= h->L || !offset || !count) return fail(BP_ERR_ARG, "bp_grad_layout: bad argument");
    *offset = h->g_off[layer]; *count = h->g_cnt[layer];
    return BP_OK;
}
extern "C" int bp_read_grads(bp_handle *h, float *host_dst, size_t n_floats)
{
    if (!h || !host_dst) return fail(BP_ERR_ARG, "null argument");
    if (!h->grad) return fail(BP_ERR_STATE, "bp_read_grads: no gradients (call bp_grads_resident first)");
    if (n_floats != h->grad_floats) return fail(BP_ERR_ARG, "bp_read_grads: size must equal bp_grad_floats");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(host_dst, h->grad, n_floats * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return BP_OK;
}

// Hidden-layer outputs y_l (post-activation, post-dropout: layer_y, BP_GPU.h:27) of the bunch processed last, as
// [bunchsize][layersizes[layer]] without padding: lets a parity test see WHICH units are on (ReLU decisions that fall
// within rounding of zero are summation-order dependent; tests/test_gpu_parity.py counts them).  fp32 handles.
extern "C" int bp_read_layer_output(bp_handle *h, int layer, float *host_dst, size_t n_floats)
{
    if (!h || !host_dst) return fail(BP_ERR_ARG, "null argument");
    if (h->bf) return fail(BP_ERR_STATE, "bp_read_layer_output: fp32 handles only");
    if (layer < 1 || layer >= h->L - 1) return fail(BP_ERR_ARG, "bp_read_layer_output: hidden layers 1 .. numlayers-2 only");
    if (n_floats != (size_t)h->B * h->s[layer]) return fail(BP_ERR_ARG, "bp_read_layer_output: size must be bunchsize*layersizes[layer]");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpy2DAsync(host_dst, (size_t)h->s[layer] * 4, h->y[layer], (size_t)h->ld[layer] * 4, (size_t)h->s[layer] * 4, h->B,
                            hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return BP_OK;
}
// Pin caller-owned host memory (hipHostRegister): uploads from it are then true DMA transfers on the copy engines instead
// of staged copies through the runtime's bounce buffers (the reference stages its uploads through pinned memory too,
// devnew_vf / cublasSetVectorAsync, BP_GPU.cu:926-992).  Optional: every upload entry point accepts pageable memory.
extern "C" int bp_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return fail(BP_ERR_ARG, "bp_host_register: null argument");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return BP_OK;
}
extern "C" int bp_host_unregister(void *p)
{
    if (!p) return fail(BP_ERR_ARG, "bp_host_unregister: null argument");
    HIPCHK(hipHostUnregister(p));
    return BP_OK;
}

extern "C" int bp_device_pci_bus_id(int device, char *buf, int len)
{
    if (!buf || len < 16) return fail(BP_ERR_ARG, "bp_device_pci_bus_id: buffer of at least 16 bytes needed");
    HIPCHK(hipDeviceGetPCIBusId(buf, len, device));
    return BP_OK;
}
// ------------------------------------------------------------------ inference / CV
// Outputs of a whole chunk stay on the device until ONE device-to-host copy at the end (the reference copies and
// synchronises per bunch and cudaMallocs per call, BP_GPU.cu:699,762-763).
static int out_chunk_reserve(bp_handle *h, int n_frames)
{
    if ((size_t)n_frames <= h->out_chunk_frames) return BP_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->out_chunk) { (void)hipFree(h->out_chunk); h->out_chunk = nullptr; }
    if (h->host_out) { (void)hipHostFree(h->host_out); h->host_out = nullptr; }
    h->out_chunk_frames = 0;
    const size_t want = (size_t)n_frames + (size_t)n_frames / 4 + 64;
    const size_t bytes = want * h->ld[h->L - 1] * sizeof(float);
    if (hipMalloc((void **)&h->out_chunk, bytes + SLACK * sizeof(float)) != hipSuccess) return fail(BP_ERR_NOMEM, "hipMalloc (chunk outputs)");
    if (hipHostMalloc((void **)&h->host_out, bytes) != hipSuccess) return fail(BP_ERR_NOMEM, "hipHostMalloc (chunk outputs)");
    h->out_chunk_frames = want;
    return BP_OK;
}

// forward of frames [first, first+fb) of the resident chunk with CV semantics; output rows go to out_chunk[first ..]
static int forward_bunch(bp_handle *h, int first, int fb)
{
    const int L = h->L;
    const float vis_keep = 1.0f - h->cfg.visible_omit, hid_keep = 1.0f - h->cfg.hid_omit;   // BP_GPU.cu:703-704
    float *out = h->out_chunk + (size_t)first * h->ld[L - 1];
    const float *x0 = h->windows ? h->x0s : h->in + (size_t)first * h->ld[0];
    if (h->windows) HIPCHK(stage_bunch(h, first, fb, false));
    if (h->bf) HIPCHK(bf_input(h, x0, fb));
    for (int l = 1; l < L; ++l) {
        float alpha = 1.0f;
        if (h->cfg.dropoutflag == 1) alpha = (l == 1) ? vis_keep : hid_keep;
        if (h->bf) { HIPCHK(bf_fwd(h, l, fb, nullptr, out, false, alpha)); continue; }
        const float *yp = (l == 1) ? x0 : h->y[l - 1];
        HIPCHK(launch_fwd(h, h->stream, l, fb, yp, nullptr, out, false, alpha));
    }
    return BP_OK;
}

// every bunch of the resident chunk (partial last bunch included, BP_GPU.cu:450-453), then one copy into host_out
static int forward_chunk(bp_handle *h, int n)
{
    int r = out_chunk_reserve(h, n);
    if (r != BP_OK) return r;
    for (int i = 0; i < n; i += h->B) {
        const int fb = h->B > n - i ? n - i : h->B;
        if ((r = forward_bunch(h, i, fb)) != BP_OK) return r;
    }
    if (n > 0) HIPCHK(hipMemcpyAsync(h->host_out, h->out_chunk, (size_t)n * h->ld[h->L - 1] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return BP_OK;
}

extern "C" int bp_forward(bp_handle *h, int n_frames, const float *in, float *out)
{
    if (!h || !in || !out) return fail(BP_ERR_ARG, "bp_forward: null argument");
    int r = bp_upload_chunk(h, n_frames, in, nullptr);
    if (r != BP_OK) return r;
    if ((r = forward_chunk(h, n_frames)) != BP_OK) return r;
    const int sL = h->s[h->L - 1], ldL = h->ld[h->L - 1];
    for (int j = 0; j < n_frames; ++j) memcpy(out + (size_t)j * sL, h->host_out + (size_t)j * ldL, sizeof(float) * sL);
    return BP_OK;
}

extern "C" int bp_forward_windows(bp_handle *h, const bp_window_chunk *c, float *out)
{
    if (!out) return fail(BP_ERR_ARG, "bp_forward_windows: null argument");
    int r = upload_windows(h, c, false, "bp_forward_windows");
    if (r != BP_OK) return r;
    if ((r = forward_chunk(h, c->n_samples)) != BP_OK) return r;
    const int sL = h->s[h->L - 1], ldL = h->ld[h->L - 1];
    for (int j = 0; j < c->n_samples; ++j) memcpy(out + (size_t)j * sL, h->host_out + (size_t)j * ldL, sizeof(float) * sL);
    return BP_OK;
}

extern "C" int bp_cv_chunk(bp_handle *h, int n_frames, const float *in, const float *targ, float *sq_err_sum)
{
    if (!h || !in || !targ || !sq_err_sum) return fail(BP_ERR_ARG, "bp_cv_chunk: null argument");
    int r = bp_upload_chunk(h, n_frames, in, nullptr);
    if (r != BP_OK) return r;
    if ((r = forward_chunk(h, n_frames)) != BP_OK) return r;
    const int sL = h->s[h->L - 1], ldL = h->ld[h->L - 1];
    float squared_err = 0.0f;
    for (int j = 0; j < n_frames; ++j)                   // fp32, frame-major / bin-minor (BP_GPU.cu:458-467)
        for (int d = 0; d < sL; ++d) {
            const float e = h->host_out[(size_t)j * ldL + d] - targ[(size_t)j * sL + d];
            squared_err = squared_err + e * e;
        }
    *sq_err_sum = squared_err;
    return BP_OK;
}

extern "C" int bp_cv_chunk_windows(bp_handle *h, const bp_window_chunk *c, float *sq_err_sum)
{
    if (!sq_err_sum) return fail(BP_ERR_ARG, "bp_cv_chunk_windows: null argument");
    int r = upload_windows(h, c, false, "bp_cv_chunk_windows");
    if (r != BP_OK) return r;
    const int L = h->L, sL = h->s[L - 1], ldL = h->ld[L - 1], n = c->n_samples;
    if (n > 0 && (!c->targ_frames || !c->targ_frame)) return fail(BP_ERR_ARG, "bp_cv_chunk_windows: null targets");
    for (int i = 0; i < n; ++i)
        if (c->targ_frame[i] < 0 || c->targ_frame[i] >= c->n_frames) return fail(BP_ERR_ARG, "bp_cv_chunk_windows: targ_frame out of range");
    if ((r = forward_chunk(h, n)) != BP_OK) return r;
    float squared_err = 0.0f;
    for (int j = 0; j < n; ++j) {                        // fp32, frame-major / bin-minor (BP_GPU.cu:458-467)
        const float *t = c->targ_frames + (size_t)c->targ_frame[j] * sL;
        for (int d = 0; d < sL; ++d) {
            const float e = h->host_out[(size_t)j * ldL + d] - t[d];
            squared_err = squared_err + e * e;
        }
    }
    *sq_err_sum = squared_err;
    return BP_OK;
}

static int get_params(bp_handle *h, float *const *w, float *const *b, bool deltas)
{
    if (!h || !w || !b) return fail(BP_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (deltas && dp_gathers_deltas(h)) { int r = dp_gather_deltas(h); if (r != BP_OK) return r; }
    for (int l = 1; l < h->L; ++l) {
        if (!w[l] || !b[l]) return fail(BP_ERR_ARG, "weights[l]/bias[l] null");
        HIPCHK(hipMemcpy2DAsync(w[l], (size_t)h->s[l] * 4, deltas ? h->dW[l] : h->W[l], (size_t)h->ld[l] * 4,
                                (size_t)h->s[l] * 4, h->s[l - 1], hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(b[l], deltas ? h->db[l] : h->b[l], (size_t)h->s[l] * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));    // reference relies on pageable-copy semantics (BP_GPU.cu:920-921)
    return dp_check(h);
}
extern "C" int bp_get_weights(bp_handle *h, float *const *w, float *const *b) { return get_params(h, w, b, false); }
extern "C" int bp_get_deltas(bp_handle *h, float *const *w, float *const *b) { return get_params(h, w, b, true); }

