// bp_engine.hip -- C-ABI implementation (include/bp_c_api.h): device state of one BP_GPU
// replacement object and the per-bunch launch sequence.  gfx950 only.
//
// Device layout (all fp32): every layer width s_l is padded to ld_l = roundup(s_l, 64); pad
// columns/rows are zero and stay zero under the step (DESIGN.md "padding invariants"), so the
// GEMM tiles never need column predicates and every row is 256-byte aligned.
//   W_l   [ld_{l-1}][ld_l]   (reference layout weights[l][p*cur+c], BP_GPU.cu:139)
//   y_l   [B][ld_l]          post-activation, post-dropout output of layer l (layer_y)
//   dx_l  [B][ld_l]          dE/dx of layer l (layer_dedx); layer_x/dydx/dedy are never stored
//   in    [cap][ld_0], targ [cap][ld_{L-1}]   resident chunk (dev.in/dev.targ, BP_GPU.cu:127-130)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bp_c_api.h"
#include "bp_kernels.h"
#include "bp_bf16.h"
#include "bp_dp.h"
#include "bp_rdv.h"
#include "bp_wgrad_dma.h"
#include "bp_wgrad_dma_bf16.h"

#include <dlfcn.h>

struct ncclUniqueIdBytes { char internal[128]; };   // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t _e = (x);                                                                          \
        if (_e != hipSuccess)                                                                         \
            return fail(BP_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(_e));               \
    } while (0)

static inline int pad64(int x) { return (x + 63) & ~63; }

struct bp_handle {
    bp_config cfg;
    int L;                       // number of layer sizes
    int s[BP_MAXLAYER], ld[BP_MAXLAYER];
    int B, Bg;                   // local / global bunch
    int cap, chunk_frames;
    hipStream_t own_stream, stream;
    bool grouped;                // all wide wgrad+update problems of a step in one grouped launch
    // parameters and momentum state live in two flat arenas with the layout of the flat gradient buffer
    // ([W_1|b_1|W_2|b_2|...], padded; g_off/g_cnt) so that data-parallel ranks can export them as ONE hipIpc
    // allocation each and the sharded update is a flat elementwise pass (bp_dp.h); W/b/dW/db point into them
    float *params, *deltas;
    float *W[BP_MAXLAYER], *b[BP_MAXLAYER], *dW[BP_MAXLAYER], *db[BP_MAXLAYER];
    struct bp_dp *dp;            // attached data-parallel group (bp_dp_attach) or null
    struct StepProf *prof;       // bp_profile_step in progress: an event after every launch of the step
    const uint8_t *inj_mask[BP_MAXLAYER];   // bp_train_resident_masked in progress: device masks of this bunch per layer output
    const float *inj_x0;                    // ... and the masked copy of its input rows
    float *y[BP_MAXLAYER], *dx[BP_MAXLAYER];
    float *in, *in_drop, *targ, *out_dev;
    float *slabs; size_t slab_stride; int out_splits;   // split-K workspace of the output layer
    float *grad; size_t grad_floats; size_t g_off[BP_MAXLAYER], g_cnt[BP_MAXLAYER];
    float *host_out;             // pinned staging for CV outputs (grow-only, whole chunk)
    float *out_chunk;            // device: network outputs of a whole chunk [frames][ld_L] (CV / forward), grow-only
    size_t out_chunk_frames;
    uint32_t step;               // bunches trained so far (dropout stream position)
    long mask_lo, mask_hi; uint32_t mask_step0;
    uint32_t th_vis, th_hid;
    hipEvent_t ev0, ev1; float last_ms; int last_bunches;
    std::vector<void *> allocs;
    // Upload path: host->device copies run on copy_stream so that chunk i+1 is uploaded while chunk i trains.
    // STACKED chunks (bp_upload_chunk: the caller hands [frames][layersizes[0]] rows, the reference's interface) alternate
    // between two device buffer pairs (in/targ and in_alt/targ_alt; allocated on first use).
    // WINDOW chunks (bp_upload_chunk_windows: raw frames + index tables) stay as they are uploaded -- two
    // grow-only staging sets alternate the same way -- and every bunch stacks ITS rows into the tile x0s/tgs right
    // before its forward (bp_stage_bunch): no stacked chunk, no masked copy of it.
    struct Raw { void *p; size_t bytes; };
    struct WinSet { Raw r[4]; } wset[2];      // raw frames, raw target frames, NAT rows, tables (win_start | targ_frame | nat_row)
    int wcur;                                 // staging set of the resident window chunk
    bool windows;                             // the resident chunk is a window chunk
    struct { const float *fea, *tg, *nat; const int *ws, *tf, *nr; int D, win; } wv;   // views of set wcur
    float *x0s, *tgs;                         // [Bp][ld_0], [Bp][ld_L]: the staged bunch (= tile stage_cur of the pair below)
    float *x0s2[2], *tgs2[2]; int stage_cur;  // two staged tiles: while bunch i trains out of one, the output layer's reduce launch
                                              // of bunch i stacks bunch i+1 into the other (bp_out_reduce_stage)
    int next_first;                           // chunk frame of the bunch that follows the one being enqueued (-1: none / not a window chunk)
    struct { bool valid; int first, tile; uint32_t step; unsigned gen; } pre;   // what the other tile holds
    unsigned wgen;                            // bumped by every window upload (a pre-staged tile of the old chunk is void)
    hipStream_t copy_stream;
    hipEvent_t ev_copy;            // copy_stream: this chunk's H2D copies are done
    hipEvent_t ev_retired;         // main stream: the stacked buffer pair that is NOT current is no longer read
    hipEvent_t ev_wretired;        // main stream: the window staging set that is NOT current is no longer read
    bool retired_valid, wretired_valid;
    float *in_alt, *targ_alt;
    // compute_dtype == 1 (bp_bf16.h): bf16 copies, each in both orientations
    bool bf;
    int Bp;                                                  // bunch rows rounded up to 64
    bf16_t *Wb[BP_MAXLAYER];                                 // ONE bf16 shadow of the weights, [prev][cur] (the forward reads it through the LDS transpose read)
    bf16_t *yb[BP_MAXLAYER], *ybT[BP_MAXLAYER];              // [Bp][ld_l], [ld_l][Bp]   (l = 0: the input bunch)
    bf16_t *dxb[BP_MAXLAYER], *dxbT[BP_MAXLAYER];
};

// bp_profile_step: one HIP event after every launch of the step on the launch stream; the duration attributed to a
// launch is the time between the previous event and its own (= kernel + the dependent-launch boundary in front of it).
struct StepProf {
    std::vector<hipEvent_t> ev; std::vector<int> kind; size_t used;
};
static hipError_t prof_mark(bp_handle *h, int kind)
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

extern "C" const char *bp_last_error(void) { return g_err.c_str(); }
extern "C" int bp_abi_version(void) { return 4; }   // 4: host-driven DP split removed; bp_rdv_*, bp_dp_attach_ex (RCCL transport), bp_dp_peer_info
extern "C" const char *bp_build_target(void) { return "gfx950"; }
extern "C" int bp_device_count(int *n)
{
    if (!n) return fail(BP_ERR_ARG, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return BP_OK;
}

// Every device buffer gets SLACK floats of zeroed tail so that whole-tile reads of the GEMM
// loaders (no predicates, see GemmArgs) stay inside the allocation.
static const size_t SLACK = 4096;
static int dev_alloc(bp_handle *h, float **p, size_t n_floats)
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

static int dp_check(bp_handle *h);
static int check_hyper(float, float, float, int, float, float, const char *);
static hipError_t dp_bunch(bp_handle *h, int first);
static int ensure_stacked(bp_handle *h);
static hipError_t stage_bunch(bp_handle *h, int first, int rows, bool train);
static StageArgs stage_args(bp_handle *h, int first, int rows, bool train, int tile, uint32_t step);
static int stage_blocks(const bp_handle *h, const StageArgs &a);
static hipError_t dp_flush(bp_handle *h);
static int dp_gather_deltas(bp_handle *h);
static int bf_alloc(bp_handle *h, bf16_t **p, size_t n_halfs);
static hipError_t bf_shadow(bp_handle *h, int l);

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
    h->step = 0; h->mask_lo = h->mask_hi = -1; h->mask_step0 = 0;
    h->th_vis = cfg->dropoutflag == 1 ? drop_threshold(cfg->visible_omit) : 0u;
    h->th_hid = cfg->dropoutflag == 1 ? drop_threshold(cfg->hid_omit) : 0u;
    for (int l = 0; l < h->L; ++l) { h->s[l] = cfg->layersizes[l]; h->ld[l] = pad64(h->s[l]); }
    h->own_stream = nullptr; h->host_out = nullptr; h->ev0 = h->ev1 = nullptr;
    h->grouped = getenv("BP_NO_GROUPED") == nullptr;
    h->in = h->in_drop = h->targ = h->out_dev = h->grad = nullptr; h->slabs = nullptr; h->out_splits = 1;
    h->last_ms = 0.f; h->last_bunches = 0; h->dp = nullptr; h->params = h->deltas = nullptr;
    h->next_first = -1; h->pre.valid = false; h->wgen = 0; h->stage_cur = 0;

#define CK(x) do { int _r = (x); if (_r != BP_OK) { std::string m = g_err; bp_destroy(h); g_err = m; return _r; } } while (0)
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
    // (the stacked chunk buffers in / in_drop / targ are allocated by the first stacked upload, ensure_stacked():
    // a caller that only ever hands window chunks never pays for them)
    CK(dev_alloc(h, &h->out_dev, Bp * h->ld[L - 1]));
    // narrow output layer (e.g. 2048 -> 257): too few 32x32 tiles to fill 256 CUs, so its k range is
    // split over 4 workgroup rows that write partial-sum slabs; bp_out_reduce finishes the layer
    if (h->ld[L - 1] <= 512 && h->ld[L - 2] >= 1024 && h->ld[L - 2] % 256 == 0) {
        h->out_splits = 4;
        // (measured round 3: 8 / 16 k-slices with 64x64 workgroup tiles -- half the operand bytes per FLOP -- are no faster:
        // C2 step 0.2216 ms with 4 slices, 0.2204 with 8, 0.2237 with 16; the layer is launch/latency-bound)
        h->slab_stride = Bp * h->ld[L - 1];
        CK(dev_alloc(h, &h->slabs, h->slab_stride * h->out_splits));
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
        if (th_vis && h->in && !h->in_drop) {                   // (no stacked buffers yet: ensure_stacked allocates it with them)
            HIPCHK(hipStreamSynchronize(h->stream));            // (an allocation in the middle of queued bunches: drain first)
            r = dev_alloc(h, &h->in_drop, ((size_t)h->cap + 64) * h->ld[0]);
            if (r != BP_OK) return r;
        }
        h->cfg.dropoutflag = dropoutflag; h->cfg.visible_omit = visible_omit; h->cfg.hid_omit = hid_omit;
        h->th_vis = th_vis;
        h->pre.valid = false;                                   // (a pre-staged bunch was masked with the old rate)
        h->th_hid = dropoutflag == 1 ? drop_threshold(hid_omit) : 0u;
        h->mask_lo = h->mask_hi = -1;
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
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int EPI, int PF = 1, int TAG = 0>
static hipError_t launch(hipStream_t st, GemmArgs g, const EpiArgs &e, int M, int N, int max_grid = 0)
{
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    int grid = g.tiles_m * g.tiles_n;
    if (max_grid > 0 && grid > max_grid) grid = max_grid;     // persistent: workgroups loop over tiles
    hipLaunchKernelGGL((bp_gemm<BM, BN, BK, WM, WN, A_KC, B_KC, EPI, PF, 0, TAG>), dim3(grid), dim3(256), 0, st, g, e);
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
static hipError_t launch_fwd(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, const float *targ,
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
        if (l == 1) return launch<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN, 1, 1>(st, g, e, M, cur);   // (own name in profiles)
        return launch<32, 64, 64, 1, 2, true, false, EPI_FWD_HIDDEN>(st, g, e, M, cur);
    }
    e.scale = 2.0f / (float)h->Bg;                       // kernSubClean: 2.0f/rows (global rows under DP)
    if (h->out_splits > 1) {
        g.K = prev / h->out_splits; g.k_split = g.K; g.slab_stride = h->slab_stride;
        e.C = h->slabs; e.ldc = cur;
        g.tiles_m = (M + 31) / 32; g.tiles_n = (cur + 31) / 32;
        hipLaunchKernelGGL((bp_gemm<32, 32, 64, 1, 1, true, false, EPI_PARTIAL, 1>),
                           dim3(g.tiles_m * g.tiles_n, h->out_splits), dim3(256), 0, st, g, e);
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) return er;
        const int n4 = M * (cur / 4), n_reduce = (n4 + 255) / 256;
        OutReduceArgs ra; memset(&ra, 0, sizeof(ra));
        ra.slabs = h->slabs; ra.slab_stride = h->slab_stride; ra.nsplit = h->out_splits; ra.M = M; ra.ld = cur; ra.n_true = h->s[l];
        ra.bias = h->b[l]; ra.alpha = alpha; ra.targ = targ; ra.scale = e.scale; ra.out = out; ra.dedx = train ? h->dx[l] : (float *)nullptr;
        if (train && h->windows && h->next_first >= 0 && st == h->stream) {
            // a window chunk with another bunch behind this one: stack (and mask, with the NEXT step's Philox position) that bunch
            // into the other tile from the spare workgroups of this launch
            const int tile = 1 - h->stage_cur;
            const StageArgs sa = stage_args(h, h->next_first, h->B, true, tile, h->step + 1);
            hipLaunchKernelGGL(bp_out_reduce_stage, dim3((unsigned)(n_reduce + stage_blocks(h, sa))), dim3(256), 0, st, ra, n_reduce, sa);
            h->pre.valid = true; h->pre.first = h->next_first; h->pre.tile = tile; h->pre.step = h->step + 1; h->pre.gen = h->wgen;
            return hipGetLastError();
        }
        hipLaunchKernelGGL(bp_out_reduce, dim3((unsigned)n_reduce), dim3(256), 0, st, ra);
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
// register-staged unrolled form it replaced was GemmKernel<64, 64, 32, 2, 2, false, false, EPI_WGRAD_UPDATE, 1, 8> (84.7 vs 80.4 us)
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

// The wgrad problems ps[0..n) (all fused or all store): grouped launches of up to 4 problems, or one each.
static hipError_t run_wgrads(hipStream_t st, Prepared *ps, int n, bool grouped)
{
    for (int i = 0; i < n;) {
        const int m = grouped ? (n - i < 4 ? n - i : 4) : 1;
        static const bool no_static = getenv("BP_WGRAD_DYNAMIC") != nullptr;    // development A/B switch
        // bunches of 128 / 256 / 512 frames (the shipped .pl uses 128, BASELINE.json 256 and 512): LDS-DMA kernel, unrolled
        int kk = !no_static ? ps[i].g.K : 0;
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

static hipError_t launch_dgrad(bp_handle *h, hipStream_t st, int l, int M)
{
    Prepared p = prep_dgrad(h, l, M);
    if (p.cfg == CFG_DGRAD_WIDE128) return run_multi<KDgradWide128, 32, 64>(st, &p, 1);
    return p.cfg == CFG_DGRAD_WIDE ? run_multi<KDgradWide, 32, 64>(st, &p, 1) : run_multi<KDgradNarrow, 32, 32>(st, &p, 1);
}
static hipError_t launch_wgrad(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, bool fused)
{
    Prepared p = prep_wgrad(h, l, M, y_prev, fused);
    return run_wgrads(st, &p, 1, false);
}

// visible-layer dropout active: bunches read the masked copy of the chunk
static inline bool use_mask(const bp_handle *h) { return !h->windows && h->in_drop && h->th_vis; }

static hipError_t mask_range(bp_handle *h, int first, int n)
{
    if (!use_mask(h) || n <= 0) return hipSuccess;
    dim3 grid((n + 3) / 4, (h->ld[0] + 255) / 256);
    hipLaunchKernelGGL(bp_mask_input, grid, dim3(256), 0, h->stream, h->in, h->in_drop, h->ld[0], h->s[0], first, n,
                       h->B, h->cfg.rank_frame_offset, h->th_vis, (uint32_t)h->cfg.seed,
                       (uint32_t)(h->cfg.seed >> 32), h->step);
    h->mask_lo = first; h->mask_hi = (long)first + n; h->mask_step0 = h->step;
    return hipGetLastError();
}

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
// fp32 master weights of layer l -> bf16 shadow in both orientations (creation, data-parallel update)
static hipError_t bf_shadow(bp_handle *h, int l)
{
    const int prev = h->ld[l - 1], cur = h->ld[l];
    return bf_convert(h, h->W[l], cur, prev, cur, h->Wb[l], cur, nullptr, 0, prev, cur);
}
// BM = 32 tiles (128-thread workgroups) when 64-row tiles would leave CUs without work
template <int EPI, bool BKN = false>
static hipError_t bf_launch(bp_handle *h, BfGemmArgs g, const BfEpiArgs &e, int M, int N)
{
    g.tiles_n = N / 64;
    static const bool no128 = getenv("BP_BF16_NO128") != nullptr;                 // development A/B switch
    if (!no128 && M % 128 == 0 && (M / 128) * g.tiles_n >= 256) {                 // 128-row tiles still fill the chip
        g.tiles_m = M / 128;
        static const bool no_dma = getenv("BP_BF16_GEMM_NO_DMA") != nullptr;      // development A/B switch
        if constexpr (EPI == BEPI_FWD_HIDDEN || EPI == BEPI_DGRAD) {              // LDS-DMA staged loop (bp_bf16.h)
            if (!no_dma && (g.tiles_n & 7) == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && e.ldc % 8 == 0 && e.ldct % 8 == 0 && e.n_limit == N) {
                // k-tile offset between the m-tiles that share a weight panel (bp_bf16.h).  Measured in the step, configs[4], us per
                // launch: forward (weights cold behind the update launch) 38.1 in phase, 36.3 four tiles apart, 33.7 a quarter of K
                // apart; dgrad (weights read by the forward 0.3 ms earlier) 31.4 in phase, 30.4 two tiles apart, 32.3 a quarter apart.
                // A quarter of K apart the four sharers no longer meet in L2 and the forward fetches its panel FOUR times (164 MB
                // instead of 68 MB per launch at the L2's memory side, profiles/r04_bf16_gemm_probe.txt): 2.6 us per launch are not
                // worth 2.4x the fabric traffic, so both stay within reach of each other's lines.  BP_BF16_ROT_FWD overrides (A/B).
                static const int rot_fwd = getenv("BP_BF16_ROT_FWD") ? atoi(getenv("BP_BF16_ROT_FWD")) : 4;
                g.k_rot = EPI == BEPI_FWD_HIDDEN ? rot_fwd : 2;
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
// G_l = y_{l-1}^T . dEdX_l  (+ fused update and shadow refresh, or store into the flat buffer), bias gradient
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
static bool bf_dma_ok(const bp_handle *h)
{
    static const bool off = getenv("BP_BF16_NO_DMA") != nullptr;                  // development A/B switch
    return !off && (h->Bp == 128 || h->Bp == 256 || h->Bp == 512 || h->Bp == 1024);
}
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
#define BF_DMA_LAUNCH(K)                                                                                             \
        do { static const bool four = getenv("BP_BF16_WGRAD_FOUR_WAVES") != nullptr;   /* development A/B switch */     \
             /* fused update: six waves (two of them own W / delta), 64-frame k-tiles in a ring of 3 when the bunch has at least 4 */ \
             if (fused && !four) hipLaunchKernelGGL((bp_wgrad_dma_bf16_six<K, (K >= 256 ? 64 : 32), (K >= 256 ? 3 : 4)>), dim3(t), dim3(384), 0, h->stream, a); \
             else if (fused) hipLaunchKernelGGL((bp_wgrad_dma_bf16<K, false>), dim3(t), dim3(256), 0, h->stream, a);    \
             else hipLaunchKernelGGL((bp_wgrad_dma_bf16<K, true>), dim3(t), dim3(256), 0, h->stream, a); } while (0)
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
static hipError_t bf_bunch(bp_handle *h, const float *x0, const float *tg, bool fused)
{
    const int L = h->L;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    CKE(bf_input(h, x0, h->B));
    for (int l = 1; l < L; ++l) CKE(bf_fwd(h, l, h->B, tg, nullptr, true, 1.0f));
    for (int l = L - 1; l >= 2; --l) CKE(bf_dgrad(h, l));       // every dgrad sees pre-update (shadow) weights
    if (bf_dma_ok(h)) {
        int ls[BP_MAXLAYER];
        for (int l = 1; l < L; ++l) ls[l - 1] = l;
        CKE(bf_wgrads_dma(h, ls, L - 1, fused));
    } else {
        for (int l = 1; l < L; ++l) CKE(bf_wgrad(h, l, fused));
    }
#undef CKE
    return hipSuccess;
}

// One bunch starting at chunk frame `first`: forward + backward.  fused: momentum update inside
// the wgrad epilogues (train_bunch_single); else gradients to the flat buffer.  Everything is
// enqueued on one stream in the reference's order (BP_GPU.cu:518-671); every dgrad of the step
// sees pre-update weights because wgrad+update(l) always follows dgrad(l).
static hipError_t bunch(bp_handle *h, int first, bool fused)
{
    const int L = h->L, B = h->B;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    const float *x0, *tg;
    if (h->windows) {                                   // window chunk: stack (and mask) this bunch's rows now
        CKE(stage_bunch(h, first, B, true));
        x0 = h->x0s; tg = h->tgs;
    } else {
        x0 = h->in + (size_t)first * h->ld[0];
        tg = h->targ + (size_t)first * h->ld[L - 1];
        if (h->inj_x0) x0 = h->inj_x0;                  // bp_train_resident_masked: input rows with the injected visible mask
        else if (use_mask(h)) {
            const bool ok = h->mask_lo >= 0 && first >= h->mask_lo && first + B <= h->mask_hi &&
                            (uint32_t)((first - h->mask_lo) / B) + h->mask_step0 == h->step &&
                            (first - h->mask_lo) % B == 0;
            if (!ok) CKE(mask_range(h, first, B));
            x0 = h->in_drop + (size_t)first * h->ld[0];
        }
    }
    if (h->bf) return bf_bunch(h, x0, tg, fused);
    for (int l = 1; l < L; ++l) {
        CKE(launch_fwd(h, h->stream, l, B, l == 1 ? x0 : h->y[l - 1], tg, nullptr, true, 1.0f));
        CKE(prof_mark(h, l == 1 ? BP_PROF_FWD_L1 : (l == L - 1 ? BP_PROF_FWD_OUT : BP_PROF_FWD_HIDDEN)));
    }
    // Every dgrad of the step reads pre-update weights (BP_GPU.cu:636 runs before :643-652 of the same
    // layer and the lower layers' updates come later), so the wgrad+update problems can all wait until
    // the last dgrad and share grouped launches (bp_gemm_multi), layer 1 (the largest) first.
    Prepared ws[BP_MAXLAYER]; int nw = 0;
    for (int l = L - 1; l >= 1; --l) {
        if (l != 1) { CKE(launch_dgrad(h, h->stream, l, B)); CKE(prof_mark(h, l == L - 1 ? BP_PROF_DGRAD_OUT : BP_PROF_DGRAD_HIDDEN)); }
        if (h->grouped) ws[nw++] = prep_wgrad(h, l, B, l == 1 ? x0 : h->y[l - 1], fused);
        else { CKE(launch_wgrad(h, h->stream, l, B, l == 1 ? x0 : h->y[l - 1], fused)); CKE(prof_mark(h, BP_PROF_WGRAD)); }
    }
    for (int i = 0; i < nw / 2; ++i) { Prepared t = ws[i]; ws[i] = ws[nw - 1 - i]; ws[nw - 1 - i] = t; }
    if (nw) { CKE(run_wgrads(h->stream, ws, nw, true)); CKE(prof_mark(h, BP_PROF_WGRAD)); }
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
    h->mask_lo = h->mask_hi = -1;
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
    if (h->cfg.dropoutflag == 1 && h->th_vis && !h->in_drop) { if ((r = dev_alloc(h, &h->in_drop, capp * h->ld[0])) != BP_OK) return r; fresh = true; }
    if (fresh) HIPCHK(hipStreamSynchronize(h->stream));         // (dev_alloc zero-fills on the main stream)
    return BP_OK;
}

// Stack rows [first, first+rows) of the resident window chunk into the bunch tile (x0s, and tgs when the chunk carries
// targets); train: with the visible-layer dropout of this step.
static StageArgs stage_args(bp_handle *h, int first, int rows, bool train, int tile, uint32_t step)
{
    const int L = h->L, ld0 = h->ld[0], ldL = h->ld[L - 1];
    StageArgs a; memset(&a, 0, sizeof(a));
    a.x = h->x0s2[tile]; a.ld = ld0; a.width = h->s[0];
    a.fea = h->wv.fea; a.fea_dim = h->wv.D; a.win = h->wv.win; a.nat = h->wv.nat;
    a.win_start = h->wv.ws + first; a.nat_row = h->wv.nr ? h->wv.nr + first : (const int *)nullptr;
    a.rows = rows; a.thresh = train ? h->th_vis : 0u; a.frame_off = h->cfg.rank_frame_offset;
    a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32); a.step = step;
    a.t = h->wv.tg ? h->tgs2[tile] : (float *)nullptr; a.ldt = ldL; a.twidth = h->s[L - 1];
    a.targ_frames = h->wv.tg; a.targ_frame = h->wv.tf ? h->wv.tf + first : (const int *)nullptr;
    a.yb_in = (ld0 + 255) / 256; a.nbx = (rows + 3) / 4;
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
    if (!h->x0s) {
        int r;
        for (int k = 0; k < 2; ++k)
            if ((r = dev_alloc(h, &h->x0s2[k], (size_t)h->Bp * h->ld[0])) != BP_OK || (r = dev_alloc(h, &h->tgs2[k], (size_t)h->Bp * h->ld[L - 1])) != BP_OK)
                return r;
        h->stage_cur = 0; h->x0s = h->x0s2[0]; h->tgs = h->tgs2[0];
        HIPCHK(hipStreamSynchronize(h->stream));                // (dev_alloc zero-fills on the main stream)
    }
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
    h->wgen++; h->pre.valid = false;
    h->chunk_frames = n;
    h->mask_lo = h->mask_hi = -1;
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
    h->mask_lo = h->mask_hi = -1;
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
    if (nb > 0 && use_mask(h)) HIPCHK(mask_range(h, first_frame, nb * h->B));
    for (int i = 0; i < nb; ++i) {
        h->next_first = (h->windows && !h->bf && i + 1 < nb) ? first_frame + (i + 1) * h->B : -1;
        if (h->dp) HIPCHK(dp_bunch(h, first_frame + i * h->B));
        else HIPCHK(bunch(h, first_frame + i * h->B, true));
        h->step++;
    }
    h->next_first = -1;
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
static hipError_t bunch(bp_handle *h, int first, bool fused);
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

// ------------------------------------------------------------------ host rendezvous (bp_rdv.h), C ABI
extern "C" int bp_rdv_open(const char *key, int world, int rank, double timeout_s, bp_rdv **out)
{
    const int rc = rdv_open(key, world, rank, timeout_s, out);
    return rc == 0 ? BP_OK : fail(rc, g_rdv_err);
}
extern "C" int bp_rdv_barrier(bp_rdv *r)
{
    if (!r) return fail(BP_ERR_ARG, "null rendezvous");
    const int rc = rdv_barrier(r);
    return rc == 0 ? BP_OK : fail(rc, g_rdv_err);
}
extern "C" int bp_rdv_allgather(bp_rdv *r, const void *mine, size_t bytes, void *all)
{
    if (!r || !mine || !all) return fail(BP_ERR_ARG, "null argument");
    const int rc = rdv_allgather(r, mine, bytes, all);
    return rc == 0 ? BP_OK : fail(rc, g_rdv_err);
}
extern "C" int bp_rdv_close(bp_rdv *r) { rdv_close(r, false); return BP_OK; }

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

// ------------------------------------------------------------------ in-library data-parallel exchange (bp_dp.h)
// What every rank publishes through the rendezvous block (RdvShm::blob).
struct DpBlob {
    int device; char pci[20];
    hipIpcMemHandle_t params, grad, deltas, flags, probe_p, probe_g;
};
static_assert(sizeof(DpBlob) <= BP_RDV_BLOB_BYTES, "rendezvous blob too small");

// RCCL transport (north_star names it): resolved at run time from librccl.so so that the library itself
// carries no link dependency on it.  Signatures from rccl.h (ROCm 7.2).
struct RcclApi {
    void *lib;
    int (*GetUniqueId)(void *id);
    int (*CommInitRank)(void **comm, int nranks, ncclUniqueIdBytes id, int rank);
    int (*ReduceScatter)(const void *send, void *recv, size_t recvcount, int dtype, int op, void *comm, hipStream_t st);
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t st);
    int (*CommDestroy)(void *comm);
    const char *(*GetErrorString)(int);
};

struct bp_dp {
    int world, rank;
    bp_rdv *rdv;
    int backend;                  // 0: native peer kernels over hipIpc mappings | 1: RCCL reduce-scatter / all-gather
    int acquire_mode;             // 0: kernel boundary behind the wait kernel | 1: + explicit system-scope acquire on every XCD
    bool distinct_devices;        // at least two ranks sit on different physical devices
    int peer_device[BP_DP_MAXRANKS]; char peer_pci[BP_DP_MAXRANKS][20];
    float *p_params[BP_DP_MAXRANKS], *p_grad[BP_DP_MAXRANKS], *p_deltas[BP_DP_MAXRANKS];
    float *p_probe_p[BP_DP_MAXRANKS], *p_probe_g[BP_DP_MAXRANKS];
    unsigned *p_flags[BP_DP_MAXRANKS];
    float *grad_fine, *grad_prev; // fine-grained gradient buffer used while attached / the handle's own one (restored at detach)
    float *probe_p, *probe_g;     // self-test probes: ordinary (like the parameter arena) / fine-grained (like the gradient buffer)
    unsigned *flags;              // own flag words (fine-grained device memory, exported)
    unsigned *arrive;             // [BP_MAXLAYER] last-arriver counters of bp_dp_reduce_update
    unsigned *done;               // [BP_MAXLAYER] tiles of layer l's gradient segment stored so far (counted by the wgrad-store kernel itself)
    unsigned done_target[BP_MAXLAYER];   // host: value done[l] reaches when the current minibatch's tiles are in
    bool counters_ok;             // the in-kernel hand-off passed the attach-time self-test (else: event + kernel boundary per group of layers)
    unsigned *err;                // pinned host word the wait kernels raise on timeout
    hipStream_t comm;             // exchange stream: signal -> wait -> reduce/update/all-gather per layer
    hipEvent_t ev_g[BP_MAXLAYER]; // main stream: gradient segment l is complete
    hipEvent_t ev_w[BP_MAXLAYER]; // comm stream (RCCL backend): the weights of layer l have been gathered
    hipEvent_t ev_comm;           // comm stream: everything queued so far is done (flush)
    unsigned epoch;               // minibatches exchanged so far (flag value of the current one)
    size_t lo[BP_MAXLAYER], hi[BP_MAXLAYER];   // this rank's slice of layer l's flat segment
    unsigned long long budget_ticks;
    bool peers_open;
    RcclApi rccl; void *rccl_comm; float *red;   // RCCL backend: communicator, reduce-scatter landing buffer (largest slice)
};

static double dp_timeout_s()
{
    const char *e = getenv("BP_DP_TIMEOUT_S");
    const double v = e ? atof(e) : 60.0;
    return v > 0.5 ? v : 0.5;
}

static void dp_release(bp_handle *h, bool failed)
{
    bp_dp *d = h->dp;
    if (!d) return;
    if (d->comm) (void)hipStreamSynchronize(d->comm);
    if (d->rccl_comm && d->rccl.CommDestroy) (void)d->rccl.CommDestroy(d->rccl_comm);
    if (d->rccl.lib) dlclose(d->rccl.lib);
    if (d->peers_open)
        for (int p = 0; p < d->world; ++p) {
            if (p == d->rank) continue;
            for (void *q : {(void *)d->p_params[p], (void *)d->p_grad[p], (void *)d->p_deltas[p], (void *)d->p_flags[p],
                            (void *)d->p_probe_p[p], (void *)d->p_probe_g[p]})
                if (q) (void)hipIpcCloseMemHandle(q);
        }
    for (auto &e : d->ev_g) if (e) (void)hipEventDestroy(e);
    for (auto &e : d->ev_w) if (e) (void)hipEventDestroy(e);
    if (d->ev_comm) (void)hipEventDestroy(d->ev_comm);
    if (d->comm) (void)hipStreamDestroy(d->comm);
    if (d->grad_fine) { if (h->grad == d->grad_fine) h->grad = d->grad_prev; (void)hipFree(d->grad_fine); }
    for (void *q : {(void *)d->flags, (void *)d->arrive, (void *)d->done, (void *)d->probe_p, (void *)d->probe_g, (void *)d->red})
        if (q) (void)hipFree(q);
    if (d->err) (void)hipHostFree(d->err);
    rdv_close(d->rdv, failed);
    delete d;
    h->dp = nullptr;
}

extern "C" int bp_dp_detach(bp_handle *h)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    if (!h->dp) return BP_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    bp_dp *d = h->dp;
    // nobody may unmap a buffer a peer kernel could still touch: everyone arrives here quiescent first
    int r = BP_OK;
    if (d->peers_open && rdv_barrier(d->rdv) != 0) r = fail(BP_ERR_STATE, g_rdv_err);
    dp_release(h, r != BP_OK);
    return r;
}

static DpPeers dp_peers(const bp_dp *d)
{
    DpPeers p; memset(&p, 0, sizeof(p));
    for (int i = 0; i < d->world; ++i) p.flags[i] = d->p_flags[i];
    return p;
}

// Attach-time check of the memory-model contract on the group's real devices (bp_dp.h, "attach-time self-test").
// Returns the number of mismatching words seen by THIS rank over all rounds (W direction in *bad_w, G in *bad_g).
static int dp_selftest(bp_handle *h, int rounds, unsigned ep_base, unsigned *bad_w, unsigned *bad_g, unsigned *bad_c)
{
    bp_dp *d = h->dp;
    unsigned *cnt = nullptr;
    float *sink = nullptr;
    HIPCHK(hipHostMalloc((void **)&cnt, 3 * sizeof(unsigned), hipHostMallocMapped));
    cnt[0] = cnt[1] = cnt[2] = 0u;
    HIPCHK(hipMalloc((void **)&sink, 64));
    const DpPeers peers = dp_peers(d);
    DpReduceArgs a; memset(&a, 0, sizeof(a));
    for (int p = 0; p < d->world; ++p) { a.params[p] = d->p_probe_p[p]; a.grads[p] = d->p_probe_g[p]; }
    a.world = d->world; a.rank = d->rank; a.peers = peers;
    int rc = BP_OK;
    for (int r = 1; r <= rounds && rc == BP_OK; ++r) {
        const unsigned ep = ep_base + (unsigned)r;             // flag values of the probe words only ever grow (fresh flag array per attach)
        // ---- (W): warm this device's caches with the OLD contents, let the peers overwrite, wait, re-read plainly
        hipLaunchKernelGGL(bp_dp_probe_touch, dim3(64), dim3(256), 0, h->stream, d->probe_p, sink);
        HIPCHK(hipStreamSynchronize(h->stream));
        if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }
        a.flag_index = bp_dp_flag_index(BP_DP_FLAG_PROBE, 0, d->rank); a.epoch = ep;
        hipLaunchKernelGGL(bp_dp_probe_push, dim3(1), dim3(256), 0, d->comm, a, (unsigned)r);
        hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, h->stream, d->flags, bp_dp_flag_index(BP_DP_FLAG_PROBE, 0, 0), d->world, ep,
                           d->budget_ticks, d->err, 3u);
        if (d->acquire_mode) hipLaunchKernelGGL(bp_dp_l2_invalidate, dim3(64), dim3(64), 0, h->stream);
        hipLaunchKernelGGL(bp_dp_probe_check, dim3(64), dim3(256), 0, h->stream, d->probe_p, d->world, (unsigned)r, cnt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipStreamSynchronize(d->comm));
        // ---- (G): fill the fine-grained probe with plain stores, signal behind the kernel boundary, peers read it
        hipLaunchKernelGGL(bp_dp_probe_fill, dim3(64), dim3(256), 0, h->stream, d->probe_g, (unsigned)r, (unsigned)d->rank);
        HIPCHK(hipEventRecord(d->ev_comm, h->stream));
        HIPCHK(hipStreamWaitEvent(d->comm, d->ev_comm, 0));
        hipLaunchKernelGGL(bp_dp_signal, dim3(1), dim3(64), 0, d->comm, peers, d->world, bp_dp_flag_index(BP_DP_FLAG_PROBE, 1, d->rank), ep);
        hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, d->comm, d->flags, bp_dp_flag_index(BP_DP_FLAG_PROBE, 1, 0), d->world, ep,
                           d->budget_ticks, d->err, 3u);
        hipLaunchKernelGGL(bp_dp_probe_check_remote, dim3(8), dim3(256), 0, d->comm, a, (unsigned)r, cnt + 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(d->comm));
        if (*(volatile unsigned *)d->err) { rc = fail(BP_ERR_STATE, "data-parallel self-test: a peer's flag never arrived"); break; }
        if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }    // nobody refills a probe a peer still reads
        // ---- (C): the same direction with the step's IN-KERNEL hand-off -- the filling kernel counts its own workgroups, the
        // exchange stream's bp_dp_sync (already queued, running beside it) sees the count, tells the peers, the peers read.
        // No event and no kernel boundary between the stores and the readers' flag.
        if (d->counters_ok) {
            const unsigned ep2 = ep + 0x4000u;                     // (flag words only grow; probe word 2 is this direction's)
            hipLaunchKernelGGL(bp_dp_sync, dim3(1), dim3(64), 0, d->comm, d->done + BP_MAXLAYER, (ep_base + (unsigned)r) * 64u, peers, d->flags, d->world,
                               bp_dp_flag_index(BP_DP_FLAG_PROBE, 2, d->rank), bp_dp_flag_index(BP_DP_FLAG_PROBE, 2, 0), ep2, d->budget_ticks, d->err, 3u);
            hipLaunchKernelGGL(bp_dp_probe_check_remote, dim3(8), dim3(256), 0, d->comm, a, (unsigned)r + 100u, cnt + 2);
            hipLaunchKernelGGL(bp_dp_probe_fill_count, dim3(64), dim3(256), 0, h->stream, d->probe_g, (unsigned)r + 100u, (unsigned)d->rank, d->done + BP_MAXLAYER);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(hipStreamSynchronize(d->comm));
            if (*(volatile unsigned *)d->err) { rc = fail(BP_ERR_STATE, "data-parallel self-test: the in-kernel hand-off never completed"); break; }
            if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }
        }
    }
    *bad_w = cnt[0]; *bad_g = cnt[1]; *bad_c = cnt[2];
    (void)hipHostFree(cnt); (void)hipFree(sink);
    return rc;
}

static int dp_load_rccl(bp_dp *d)
{
    RcclApi &r = d->rccl;
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return fail(BP_ERR_STATE, std::string("bp_dp_attach: RCCL transport requested but librccl.so cannot be loaded: ") + dlerror());
    *(void **)&r.GetUniqueId = dlsym(r.lib, "ncclGetUniqueId");
    *(void **)&r.CommInitRank = dlsym(r.lib, "ncclCommInitRank");
    *(void **)&r.ReduceScatter = dlsym(r.lib, "ncclReduceScatter");
    *(void **)&r.AllGather = dlsym(r.lib, "ncclAllGather");
    *(void **)&r.CommDestroy = dlsym(r.lib, "ncclCommDestroy");
    *(void **)&r.GetErrorString = dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.ReduceScatter || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
        return fail(BP_ERR_STATE, "bp_dp_attach: librccl.so lacks an expected symbol");
    return BP_OK;
}

extern "C" int bp_dp_attach_ex(bp_handle *h, int world, int rank, const char *key, int transport)
{
    if (!h || !key || !*key) return fail(BP_ERR_ARG, "bp_dp_attach: null argument");
    if (world < 1 || world > BP_DP_MAXRANKS || rank < 0 || rank >= world)
        return fail(BP_ERR_ARG, "bp_dp_attach: world must be 1..8 and 0 <= rank < world");
    if (transport != BP_DP_TRANSPORT_NATIVE && transport != BP_DP_TRANSPORT_RCCL) return fail(BP_ERR_ARG, "bp_dp_attach: unknown transport");
    if (transport == BP_DP_TRANSPORT_RCCL && (world & (world - 1)) != 0)
        return fail(BP_ERR_ARG, "bp_dp_attach: the RCCL transport needs a world of 1, 2, 4 or 8 (equal slices)");
    if (h->dp) return fail(BP_ERR_STATE, "bp_dp_attach: handle is already attached");
    if (h->Bg != h->B * world || h->cfg.rank_frame_offset != rank * h->B)
        return fail(BP_ERR_ARG, "bp_dp_attach: create the handle with global_bunchsize = world*bunchsize and rank_frame_offset = rank*bunchsize");
    if (h->L - 1 >= 16) return fail(BP_ERR_ARG, "bp_dp_attach: too many layers");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    bp_dp *d = new bp_dp();
    memset((void *)d, 0, sizeof(*d));
    h->dp = d;
    d->world = world; d->rank = rank; d->epoch = 0; d->peers_open = false; d->backend = transport;
    d->budget_ticks = (unsigned long long)(dp_timeout_s() * 1.0e8);      // wall_clock64: 100 MHz
#define DK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::string m = std::string("bp_dp_attach: ") + #x + ": " + hipGetErrorString(_e); \
        dp_release(h, true); return fail(BP_ERR_DEVICE, m); } } while (0)
#define DR(x) do { int _r = (x); if (_r != BP_OK) { std::string m = g_err; dp_release(h, true); g_err = m; return _r; } } while (0)
    // the gradient buffer peers read: fine-grained (uncached in every mapping, written through by the wgrad kernels)
    // (if the runtime refuses a fine-grained allocation of this size, an ordinary one still works with the system-scope
    // loads of bp_dp_reduce_update on ONE device; across devices the self-test below decides)
    if (transport == BP_DP_TRANSPORT_RCCL ||      // (RCCL's kernels read it locally: ordinary cached memory)
        hipExtMallocWithFlags((void **)&d->grad_fine, (h->grad_floats + SLACK) * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        d->grad_fine = nullptr;
        DK(hipMalloc((void **)&d->grad_fine, (h->grad_floats + SLACK) * sizeof(float)));
    }
    DK(hipMemset(d->grad_fine, 0, (h->grad_floats + SLACK) * sizeof(float)));
    d->grad_prev = h->grad; h->grad = d->grad_fine;
    DK(hipExtMallocWithFlags((void **)&d->flags, BP_DP_FLAG_WORDS * sizeof(unsigned), hipDeviceMallocFinegrained));
    DK(hipMemset(d->flags, 0, BP_DP_FLAG_WORDS * sizeof(unsigned)));
    DK(hipMalloc((void **)&d->arrive, BP_MAXLAYER * sizeof(unsigned)));
    DK(hipMemset(d->arrive, 0, BP_MAXLAYER * sizeof(unsigned)));
    DK(hipMalloc((void **)&d->done, (BP_MAXLAYER + 1) * sizeof(unsigned)));       // (+1: the self-test's counter)
    DK(hipMemset(d->done, 0, (BP_MAXLAYER + 1) * sizeof(unsigned)));
    d->counters_ok = transport != BP_DP_TRANSPORT_RCCL && getenv("BP_DP_NO_COUNTERS") == nullptr;
    DK(hipMalloc((void **)&d->probe_p, BP_DP_PROBE_FLOATS * sizeof(float)));
    DK(hipMemset(d->probe_p, 0, BP_DP_PROBE_FLOATS * sizeof(float)));
    if (hipExtMallocWithFlags((void **)&d->probe_g, BP_DP_PROBE_FLOATS * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        d->probe_g = nullptr;
        DK(hipMalloc((void **)&d->probe_g, BP_DP_PROBE_FLOATS * sizeof(float)));
    }
    DK(hipMemset(d->probe_g, 0, BP_DP_PROBE_FLOATS * sizeof(float)));
    DK(hipHostMalloc((void **)&d->err, sizeof(unsigned), hipHostMallocMapped));
    *d->err = 0u;
    {   // the exchange yields to the GEMMs of the main stream when both have workgroups to place
        int lo_prio = 0, hi_prio = 0;
        DK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
        DK(hipStreamCreateWithPriority(&d->comm, hipStreamNonBlocking, lo_prio));
    }
    for (int l = 1; l < h->L; ++l) {
        DK(hipEventCreateWithFlags(&d->ev_g[l], hipEventDisableTiming));
        DK(hipEventCreateWithFlags(&d->ev_w[l], hipEventDisableTiming));
    }
    DK(hipEventCreateWithFlags(&d->ev_comm, hipEventDisableTiming));
    DK(hipStreamSynchronize(h->stream));
    size_t max_slice = 4;
    for (int l = 1; l < h->L; ++l) {                           // equal float4-aligned slices of [W_l|b_l]
        const size_t cnt4 = h->g_cnt[l] / 4, per4 = (cnt4 + world - 1) / world;
        const size_t a = per4 * rank < cnt4 ? per4 * rank : cnt4, b = per4 * (rank + 1) < cnt4 ? per4 * (rank + 1) : cnt4;
        d->lo[l] = h->g_off[l] + 4 * a; d->hi[l] = h->g_off[l] + 4 * b;
        if (4 * per4 > max_slice) max_slice = 4 * per4;
        if (transport == BP_DP_TRANSPORT_RCCL && per4 * world != cnt4) { dp_release(h, true); return fail(BP_ERR_ARG, "bp_dp_attach: RCCL transport: layer segment not divisible by the world"); }
    }
    // ---- rendezvous: publish device + hipIpc handles, map every peer's
    {
        bp_rdv *rv = nullptr;
        if (rdv_open(key, world, rank, dp_timeout_s(), &rv) != 0) { dp_release(h, true); return fail(BP_ERR_STATE, "bp_dp_attach: " + g_rdv_err); }
        d->rdv = rv;
    }
    DpBlob mine; memset(&mine, 0, sizeof(mine));
    mine.device = h->cfg.device;
    DK(hipDeviceGetPCIBusId(mine.pci, (int)sizeof(mine.pci), h->cfg.device));
    DK(hipIpcGetMemHandle(&mine.params, h->params));
    DK(hipIpcGetMemHandle(&mine.grad, h->grad));
    DK(hipIpcGetMemHandle(&mine.deltas, h->deltas));
    DK(hipIpcGetMemHandle(&mine.flags, d->flags));
    DK(hipIpcGetMemHandle(&mine.probe_p, d->probe_p));
    DK(hipIpcGetMemHandle(&mine.probe_g, d->probe_g));
    memcpy(d->rdv->shm->blob[rank], &mine, sizeof(mine));
    if (transport == BP_DP_TRANSPORT_RCCL) {
        DR(dp_load_rccl(d));
        if (rank == 0) {
            static_assert(sizeof(ncclUniqueIdBytes) <= sizeof(d->rdv->shm->shared), "unique id does not fit");
            ncclUniqueIdBytes id;
            const int e = d->rccl.GetUniqueId(&id);
            if (e != 0) { dp_release(h, true); return fail(BP_ERR_DEVICE, std::string("ncclGetUniqueId: ") + d->rccl.GetErrorString(e)); }
            memcpy(d->rdv->shm->shared, &id, sizeof(id));
        }
    }
    if (rdv_barrier(d->rdv) != 0) { dp_release(h, true); return fail(BP_ERR_STATE, g_rdv_err); }
    d->peers_open = true;
    for (int p = 0; p < world; ++p) {
        DpBlob pb; memcpy(&pb, d->rdv->shm->blob[p], sizeof(pb));
        d->peer_device[p] = pb.device; memcpy(d->peer_pci[p], pb.pci, sizeof(pb.pci)); d->peer_pci[p][sizeof(pb.pci) - 1] = 0;
        if (strcmp(pb.pci, mine.pci) != 0) d->distinct_devices = true;
        if (p == rank) {
            d->p_params[p] = h->params; d->p_grad[p] = h->grad; d->p_deltas[p] = h->deltas; d->p_flags[p] = d->flags;
            d->p_probe_p[p] = d->probe_p; d->p_probe_g[p] = d->probe_g;
            continue;
        }
        DK(hipIpcOpenMemHandle((void **)&d->p_params[p], pb.params, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_grad[p], pb.grad, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_deltas[p], pb.deltas, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_flags[p], pb.flags, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_probe_p[p], pb.probe_p, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_probe_g[p], pb.probe_g, hipIpcMemLazyEnablePeerAccess));
    }
    if (rdv_barrier(d->rdv) != 0) { dp_release(h, true); return fail(BP_ERR_STATE, g_rdv_err); }   // every rank has mapped every peer
    if (transport == BP_DP_TRANSPORT_RCCL) {
        ncclUniqueIdBytes id; memcpy(&id, d->rdv->shm->shared, sizeof(id));
        const int e = d->rccl.CommInitRank(&d->rccl_comm, world, id, rank);
        if (e != 0) { std::string m = std::string("ncclCommInitRank: ") + d->rccl.GetErrorString(e); dp_release(h, true); return fail(BP_ERR_DEVICE, m); }
        DK(hipMalloc((void **)&d->red, (max_slice + SLACK) * sizeof(float)));
    } else if (world > 1) {
        // ---- the memory-model contract of the native exchange, checked on these devices before anything relies on it
        for (int mode = 0; mode < 2; ++mode) {
            d->acquire_mode = mode;
            unsigned bw = 0, bg = 0, bc = 0;
            DR(dp_selftest(h, 4, 4u * (unsigned)mode, &bw, &bg, &bc));
            unsigned mine2[3] = {bw, bg, bc}, all[3 * BP_DP_MAXRANKS];
            if (rdv_allgather(d->rdv, mine2, sizeof(mine2), all) != 0) { dp_release(h, true); return fail(BP_ERR_STATE, g_rdv_err); }
            unsigned tw = 0, tg = 0, tc = 0;
            for (int p = 0; p < world; ++p) { tw += all[3 * p]; tg += all[3 * p + 1]; tc += all[3 * p + 2]; }
            if (tc) d->counters_ok = false;                        // (every rank sees the same verdict) the event + kernel-boundary hand-off stays
            if (tg) { dp_release(h, true); return fail(BP_ERR_STATE, "bp_dp_attach: self-test failed: peers read stale gradient words from a fine-grained buffer (" + std::to_string(tg) + " words)"); }
            if (!tw) break;
            if (mode == 1) { dp_release(h, true); return fail(BP_ERR_STATE, "bp_dp_attach: self-test failed: stale weights after a peer's write-through stores even with an explicit acquire (" + std::to_string(tw) + " words)"); }
        }
    }
#undef DK
#undef DR
    return BP_OK;
}
extern "C" int bp_dp_attach(bp_handle *h, int world, int rank, const char *key) { return bp_dp_attach_ex(h, world, rank, key, BP_DP_TRANSPORT_NATIVE); }

static int dp_check(bp_handle *h)
{
    if (h->dp && *(volatile unsigned *)h->dp->err) {
        const unsigned e = *(volatile unsigned *)h->dp->err;
        return fail(BP_ERR_STATE, "data-parallel exchange timed out on the device: waiting for rank " + std::to_string((e % 1000u) - 1u) +
                                      (e / 1000u == 1 ? " (gradient ready)" : " (weights gathered)"));
    }
    return BP_OK;
}

// main stream: the weights of layer l gathered from every rank for minibatch `epoch` (none before the first)
static hipError_t dp_wait_weights(bp_handle *h, int l, unsigned epoch)
{
    bp_dp *d = h->dp;
    if (epoch == 0) return hipSuccess;
    if (d->backend == BP_DP_TRANSPORT_RCCL) return hipStreamWaitEvent(h->stream, d->ev_w[l], 0);
    hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, h->stream, d->flags, bp_dp_flag_index(BP_DP_FLAG_W, l, 0), d->world, epoch,
                       d->budget_ticks, d->err, 2u);
    if (d->acquire_mode) hipLaunchKernelGGL(bp_dp_l2_invalidate, dim3(64), dim3(64), 0, h->stream);
    return hipGetLastError();
}

static void dp_update_args(bp_handle *h, int l, DpReduceArgs &a)
{
    bp_dp *d = h->dp;
    memset(&a, 0, sizeof(a));
    a.delta = h->deltas; a.lo = d->lo[l]; a.hi = d->hi[l];
    a.w_end = h->g_off[l] + (size_t)h->ld[l - 1] * h->ld[l];
    a.world = d->world; a.rank = d->rank;
    const float m = h->cfg.momentum, lr = h->cfg.lrate;
    a.mom = m; a.c1 = h->cfg.momentum_rule == 1 ? lr : (1 - m) * lr; a.wc = h->cfg.weightcost; a.ndiv = (float)h->Bg;
    a.arrive = d->arrive + l; a.peers = dp_peers(d); a.flag_index = bp_dp_flag_index(BP_DP_FLAG_W, l, d->rank); a.epoch = d->epoch;
}
static int dp_update_grid(const DpReduceArgs &a)
{
    const size_t n4 = (a.hi - a.lo) / 4;
    static const int max_grid = getenv("BP_DP_GRID") ? atoi(getenv("BP_DP_GRID")) : 128;   // development A/B switch
    // few, deep workgroups: at 2048 workgroups the exchange kernel crowds the GEMMs it runs beside out of the CUs'
    // memory pipes (C2 step through the exchange path on one GPU: 0.51 ms at 2048, 0.34 at 512, 0.28 at 128, 0.31 at 64)
    int grid = (int)((n4 + 256 * BP_DP_UNROLL - 1) / (256 * BP_DP_UNROLL));
    if (grid > max_grid) grid = max_grid;
    return grid < 1 ? 1 : grid;                                // (an empty slice still raises its flag)
}

// comm stream: reduce this rank's slice of layer l over all ranks, update it, write the new weights to every rank (native transport)
static hipError_t dp_reduce_layer(bp_handle *h, int l)
{
    bp_dp *d = h->dp;
    DpReduceArgs a;
    dp_update_args(h, l, a);
    for (int p = 0; p < d->world; ++p) { a.grads[p] = d->p_grad[p]; a.params[p] = d->p_params[p]; }
    const int grid = dp_update_grid(a);
    switch (d->world) {
    case 1: hipLaunchKernelGGL(bp_dp_reduce_update<1>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 2: hipLaunchKernelGGL(bp_dp_reduce_update<2>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 4: hipLaunchKernelGGL(bp_dp_reduce_update<4>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 8: hipLaunchKernelGGL(bp_dp_reduce_update<8>, dim3(grid), dim3(256), 0, d->comm, a); break;
    default: hipLaunchKernelGGL(bp_dp_reduce_update<0>, dim3(grid), dim3(256), 0, d->comm, a); break;
    }
    return hipGetLastError();
}

// comm stream, after the gradient segments of layers ls[0..n) are complete on the main stream (ONE event: every event
// record costs the main stream a ~6 us bubble, profiles/r03_dp_world1_timeline.txt): tell every rank, wait for every
// rank's segments, then per layer reduce this rank's slice, update it and write the new weights to every rank
static hipError_t dp_exchange_layers(bp_handle *h, const int *ls, int n)
{
    bp_dp *d = h->dp;
    hipError_t er;
    if ((er = hipEventRecord(d->ev_g[ls[0]], h->stream)) != hipSuccess) return er;
    if ((er = hipStreamWaitEvent(d->comm, d->ev_g[ls[0]], 0)) != hipSuccess) return er;
    if (d->backend != BP_DP_TRANSPORT_RCCL) {
        const DpPeers peers = dp_peers(d);
        DpIdx sig, wt; sig.n = wt.n = n;
        for (int i = 0; i < n; ++i) { sig.index[i] = bp_dp_flag_index(BP_DP_FLAG_GRAD, ls[i], d->rank); wt.index[i] = bp_dp_flag_index(BP_DP_FLAG_GRAD, ls[i], 0); }
        hipLaunchKernelGGL(bp_dp_signal_n, dim3(1), dim3(64), 0, d->comm, peers, d->world, sig, d->epoch);
        hipLaunchKernelGGL(bp_dp_wait_n, dim3(1), dim3(64), 0, d->comm, d->flags, wt, d->world, d->epoch, d->budget_ticks, d->err, 1u);
    }
    for (int i = 0; i < n; ++i) {
        const int l = ls[i];
        DpReduceArgs a;
        dp_update_args(h, l, a);
        if (d->backend == BP_DP_TRANSPORT_RCCL) {
            // reduce-scatter of the segment into `red` (this rank's slice), sharded update on it, all-gather of the new W
            // slice in place in the parameter arena; RCCL orders the ranks, the event orders the next forward of this layer
            const size_t cnt = d->hi[l] - d->lo[l];
            int e = d->rccl.ReduceScatter(h->grad + h->g_off[l], d->red, cnt, 7 /* ncclFloat32 */, 0 /* ncclSum */, d->rccl_comm, d->comm);
            if (e != 0) return hipErrorUnknown;
            a.grads[0] = d->red - a.lo;                            // the kernel indexes grads[p] + lo
            a.params[0] = h->params;
            a.world = 1; a.rank = 0;
            a.peers.flags[0] = d->flags;                           // (flag raised on this rank only; nobody waits for it)
            hipLaunchKernelGGL(bp_dp_reduce_update<1>, dim3(dp_update_grid(a)), dim3(256), 0, d->comm, a);
            if ((er = hipGetLastError()) != hipSuccess) return er;
            e = d->rccl.AllGather(h->params + d->lo[l], h->params + h->g_off[l], cnt, 7, d->rccl_comm, d->comm);
            if (e != 0) return hipErrorUnknown;
            if ((er = hipEventRecord(d->ev_w[l], d->comm)) != hipSuccess) return er;
            continue;
        }
        if ((er = dp_reduce_layer(h, l)) != hipSuccess) return er;
    }
    return hipSuccess;
}

// One data-parallel minibatch (this rank's shard starts at chunk frame `first`).  Per layer: wait for the gathered
// weights of the previous minibatch right before the layer's forward; after all dgrads the weight gradients go out
// largest segment first, each followed at once by its exchange on the comm stream -- so the exchange of layer l
// overlaps the remaining weight gradients and the NEXT minibatch's forward of the layers before l.
static hipError_t dp_bunch(bp_handle *h, int first)
{
    bp_dp *d = h->dp;
    const int L = h->L, B = h->B;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    const float *x0, *tg;
    if (h->windows) {                                   // window chunk: stack (and mask) this bunch's rows now
        CKE(stage_bunch(h, first, B, true));
        x0 = h->x0s; tg = h->tgs;
    } else {
        x0 = h->in + (size_t)first * h->ld[0];
        tg = h->targ + (size_t)first * h->ld[L - 1];
        if (use_mask(h)) {
            const bool ok = h->mask_lo >= 0 && first >= h->mask_lo && first + B <= h->mask_hi &&
                            (uint32_t)((first - h->mask_lo) / B) + h->mask_step0 == h->step && (first - h->mask_lo) % B == 0;
            if (!ok) CKE(mask_range(h, first, B));
            x0 = h->in_drop + (size_t)first * h->ld[0];
        }
    }
    const unsigned prev_epoch = d->epoch;
    d->epoch++;
    if (h->bf) {
        for (int l = 1; l < L; ++l) {
            CKE(dp_wait_weights(h, l, prev_epoch));
            if (prev_epoch) CKE(bf_shadow(h, l));               // bf16 copies of the gathered fp32 weights
            if (l == 1) CKE(bf_input(h, x0, B));
            CKE(bf_fwd(h, l, B, tg, nullptr, true, 1.0f));
        }
        for (int l = L - 1; l >= 2; --l) CKE(bf_dgrad(h, l));
        // layer 1 (the largest segment, needed first by the next forward) goes out alone; the rest as one group: its
        // exchange queues behind layer 1's on the comm stream anyway, and one launch + one event replace L-2 of each
        int rest[BP_MAXLAYER], nrest = 0;
        for (int l = 2; l < L; ++l) rest[nrest++] = l;
        const int one = 1;
        if (bf_dma_ok(h)) {
            CKE(bf_wgrads_dma(h, &one, 1, false)); CKE(dp_exchange_layers(h, &one, 1));
            if (nrest) { CKE(bf_wgrads_dma(h, rest, nrest, false)); CKE(dp_exchange_layers(h, rest, nrest)); }
        } else {
            for (int l = 1; l < L; ++l) { CKE(bf_wgrad(h, l, false)); CKE(dp_exchange_layers(h, &l, 1)); }
        }
    } else {
        for (int l = 1; l < L; ++l) {
            CKE(dp_wait_weights(h, l, prev_epoch));
            CKE(launch_fwd(h, h->stream, l, B, l == 1 ? x0 : h->y[l - 1], tg, nullptr, true, 1.0f));
        }
        for (int l = L - 1; l >= 2; --l) CKE(launch_dgrad(h, h->stream, l, B));
        Prepared ws[BP_MAXLAYER]; int rest[BP_MAXLAYER], nrest = 0;
        const int one = 1;
        const bool static_k = B == 128 || B == 256 || B == 512;         // (the LDS-DMA store kernel, the one that counts its tiles)
        if (d->counters_ok && d->backend != BP_DP_TRANSPORT_RCCL && static_k && L - 1 <= 4) {
            // ONE grouped weight-gradient launch, layer 1's tiles first; every tile counts itself into done[l] (bp_wgrad_dma.h), and
            // the exchange stream -- queued right here, running beside the launch -- picks each layer up as soon as its count is
            // complete: no event on this stream (each cost it a ~7 us bubble), no split of the launch, and layer 1's exchange
            // overlaps the other layers' tiles instead of waiting behind a launch boundary.
            int nw = 0;
            for (int l = 1; l < L; ++l) {
                ws[nw] = prep_wgrad(h, l, B, l == 1 ? x0 : h->y[l - 1], false);
                ws[nw].e.done = d->done + l;
                d->done_target[l] += (unsigned)(((h->ld[l - 1] + 63) / 64) * ((h->ld[l] + 63) / 64));
                ++nw;
            }
            const DpPeers peers = dp_peers(d);
            for (int l = 1; l < L; ++l) {
                hipLaunchKernelGGL(bp_dp_sync, dim3(1), dim3(64), 0, d->comm, d->done + l, d->done_target[l], peers, d->flags, d->world,
                                   bp_dp_flag_index(BP_DP_FLAG_GRAD, l, d->rank), bp_dp_flag_index(BP_DP_FLAG_GRAD, l, 0), d->epoch, d->budget_ticks, d->err, 1u);
                CKE(hipGetLastError());
                CKE(dp_reduce_layer(h, l));
            }
            CKE(run_wgrads(h->stream, ws, nw, true));
        } else {
            // layer 1 (the largest segment, needed first by the next forward) goes out alone; the rest as ONE grouped launch:
            // its exchange queues behind layer 1's on the comm stream anyway, and one launch + one event replace L-2 of each
            CKE(launch_wgrad(h, h->stream, 1, B, x0, false));
            CKE(dp_exchange_layers(h, &one, 1));
            for (int l = 2; l < L; ++l) { ws[nrest] = prep_wgrad(h, l, B, h->y[l - 1], false); rest[nrest++] = l; }
            if (nrest) {
                CKE(run_wgrads(h->stream, ws, nrest, true));
                CKE(dp_exchange_layers(h, rest, nrest));
            }
        }
    }
#undef CKE
    return hipSuccess;
}

// After the last minibatch of a call: the main stream waits until every layer's weights have been gathered (and
// therefore every peer has finished reading this rank's gradients), so that stream order again covers everything.
static hipError_t dp_flush(bp_handle *h)
{
    bp_dp *d = h->dp;
    hipError_t er;
    for (int l = 1; l < h->L; ++l) {
        if ((er = dp_wait_weights(h, l, d->epoch)) != hipSuccess) return er;
        if (h->bf && d->epoch && (er = bf_shadow(h, l)) != hipSuccess) return er;
    }
    if ((er = hipEventRecord(d->ev_comm, d->comm)) != hipSuccess) return er;
    return hipStreamWaitEvent(h->stream, d->ev_comm, 0);
}

// bp_get_deltas on an attached handle: the momentum state is sharded; pull the peers' slices into the local arena.
static int dp_gather_deltas(bp_handle *h)
{
    bp_dp *d = h->dp;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (rdv_barrier(d->rdv) != 0) return fail(BP_ERR_STATE, g_rdv_err);   // every rank quiescent: slices are final
    for (int l = 1; l < h->L; ++l) {
        const size_t cnt4 = h->g_cnt[l] / 4, per4 = (cnt4 + d->world - 1) / d->world;
        for (int p = 0; p < d->world; ++p) {
            if (p == d->rank) continue;
            const size_t a = per4 * p < cnt4 ? per4 * p : cnt4, b = per4 * (p + 1) < cnt4 ? per4 * (p + 1) : cnt4;
            if (b <= a) continue;
            const size_t off = h->g_off[l] + 4 * a;
            int grid = (int)((b - a + 255) / 256); if (grid > 1024) grid = 1024;
            hipLaunchKernelGGL(bp_dp_copy, dim3(grid), dim3(256), 0, h->stream, h->deltas + off, d->p_deltas[p] + off, (unsigned long long)(b - a));
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (rdv_barrier(d->rdv) != 0) return fail(BP_ERR_STATE, g_rdv_err);   // nobody resumes training while a peer still reads
    return BP_OK;
}

extern "C" int bp_dp_info(bp_handle *h, int *world, int *rank, unsigned *minibatches)
{
    if (!h) return fail(BP_ERR_ARG, "null handle");
    if (world) *world = h->dp ? h->dp->world : 0;
    if (rank) *rank = h->dp ? h->dp->rank : 0;
    if (minibatches) *minibatches = h->dp ? h->dp->epoch : 0;
    return BP_OK;
}

extern "C" int bp_dp_peer_info(bp_handle *h, int peer, int *device, char *pci_bus_id, int len, int *transport, int *acquire_mode)
{
    if (!h || !h->dp) return fail(BP_ERR_STATE, "bp_dp_peer_info: handle is not attached");
    if (peer < 0 || peer >= h->dp->world) return fail(BP_ERR_ARG, "bp_dp_peer_info: peer out of range");
    if (device) *device = h->dp->peer_device[peer];
    if (pci_bus_id && len > 0) { strncpy(pci_bus_id, h->dp->peer_pci[peer], (size_t)len - 1); pci_bus_id[len - 1] = 0; }
    if (transport) *transport = h->dp->backend;
    if (acquire_mode) *acquire_mode = h->dp->acquire_mode;
    return BP_OK;
}
extern "C" int bp_dp_barrier(bp_handle *h)
{
    if (!h || !h->dp) return fail(BP_ERR_STATE, "bp_dp_barrier: handle is not attached");
    return rdv_barrier(h->dp->rdv) == 0 ? BP_OK : fail(BP_ERR_STATE, g_rdv_err);
}
extern "C" int bp_dp_allgather(bp_handle *h, const void *mine, size_t bytes, void *all)
{
    if (!h || !h->dp || !mine || !all) return fail(BP_ERR_STATE, "bp_dp_allgather: handle is not attached / null argument");
    return rdv_allgather(h->dp->rdv, mine, bytes, all) == 0 ? BP_OK : fail(BP_ERR_STATE, g_rdv_err);
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
    if (deltas && h->dp && h->dp->world > 1) { int r = dp_gather_deltas(h); if (r != BP_OK) return r; }
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


// ------------------------------------------------------------------ in-step kernel timing + measured peaks
// bp_train_resident over [first_frame, first_frame + n_bunches*bunchsize) with an event after every launch: the
// per-class average duration of the step's kernels AS THEY RUN IN THE STEP (same order, same cache state as the
// timed loop), for the roofline object.  Classes: BP_PROF_* in bp_c_api.h.  fp32 single-device handles only.
extern "C" int bp_profile_step(bp_handle *h, int first_frame, int n_bunches, float *avg_ms, int *launches_per_step)
{
    if (!h || !avg_ms) return fail(BP_ERR_ARG, "bp_profile_step: null argument");
    if (h->bf || h->dp || h->Bg != h->B) return fail(BP_ERR_STATE, "bp_profile_step: fp32 single-device handles only");
    if (n_bunches < 1 || first_frame < 0 || (long)first_frame + (long)n_bunches * h->B > h->chunk_frames)
        return fail(BP_ERR_ARG, "bp_profile_step: frame range outside the resident chunk");
    HIPCHK(hipSetDevice(h->cfg.device));
    StepProf prof; prof.used = 0;
    int rc = BP_OK;
    if (use_mask(h)) HIPCHK(mask_range(h, first_frame, n_bunches * h->B));
    h->prof = &prof;
    hipError_t er = prof_mark(h, -1);                        // origin
    for (int i = 0; er == hipSuccess && i < n_bunches; ++i) {
        er = bunch(h, first_frame + i * h->B, true);
        h->step++;
    }
    h->prof = nullptr;
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    double sum[BP_PROF_KINDS] = {0}; long cnt[BP_PROF_KINDS] = {0};
    for (size_t k = 1; er == hipSuccess && k < prof.used; ++k) {
        float ms = 0.f;
        er = hipEventElapsedTime(&ms, prof.ev[k - 1], prof.ev[k]);
        if (prof.kind[k] >= 0 && prof.kind[k] < BP_PROF_KINDS) { sum[prof.kind[k]] += ms; cnt[prof.kind[k]]++; }
    }
    for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
    if (er != hipSuccess) rc = fail(BP_ERR_DEVICE, std::string("bp_profile_step: ") + hipGetErrorString(er));
    for (int k = 0; k < BP_PROF_KINDS; ++k) {
        avg_ms[k] = cnt[k] ? (float)(sum[k] / (double)cnt[k]) : 0.f;
        if (launches_per_step) launches_per_step[k] = (int)(cnt[k] / n_bunches);
    }
    return rc;
}

// Measured peaks of THIS device, taken in the same process as the benchmark: a bare v_mfma_f32_32x32x2_f32 loop
// (4 independent accumulator chains per wave, 4 waves per SIMD-quad workgroup, no memory traffic) and a float4
// device-to-device copy of 2 x 1 GiB (read + write bytes counted).
__global__ __launch_bounds__(256) void bp_peak_mfma_f32(float *sink, int iters, float seed)
{
    f32x16 a0, a1, a2, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = seed; a1[r] = seed; a2[r] = seed; a3[r] = seed; }
    const float x = seed + (float)threadIdx.x * 1e-9f, y = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 1.2345e-30f) sink[threadIdx.x] = s;
}
// One float4 per thread, no loop, the whole 1 GiB in one grid (tools/copy_probe.hip: 6.27 TB/s plain, 6.59 TB/s with
// nontemporal accesses on these boxes; the grid-stride form with 4 loads in flight that stood here before reached 4.5-4.8).
template <bool NT>
__global__ __launch_bounds__(256) void bp_peak_copy(float4 *dst, const float4 *src, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4v *>(src) + i), reinterpret_cast<f4v *>(dst) + i);
    else dst[i] = src[i];
}
extern "C" int bp_measure_peaks(bp_handle *h, float *mfma_f32_tflops, float *hbm_copy_gbs)
{
    if (!h || !mfma_f32_tflops || !hbm_copy_gbs) return fail(BP_ERR_ARG, "bp_measure_peaks: null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    float *sink = nullptr;
    HIPCHK(hipMalloc((void **)&sink, 4096));
    const int iters = 4096, wgs = 256 * 8;
    float ms = 0.f, best = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        HIPCHK(hipEventRecord(a, h->stream));
        hipLaunchKernelGGL(bp_peak_mfma_f32, dim3(wgs), dim3(256), 0, h->stream, sink, iters, 1.0f);
        HIPCHK(hipEventRecord(b, h->stream));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        const float tf = (float)((double)wgs * 4 * iters * 4 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12);
        if (rep > 0 && tf > best) best = tf;
    }
    *mfma_f32_tflops = best;
    (void)hipFree(sink);
    const size_t bytes = (size_t)1 << 30;
    float4 *src = nullptr, *dst = nullptr;
    HIPCHK(hipMalloc((void **)&src, bytes)); HIPCHK(hipMalloc((void **)&dst, bytes));
    HIPCHK(hipMemsetAsync(src, 1, bytes, h->stream));
    best = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        HIPCHK(hipEventRecord(a, h->stream));
        if (rep < 3) hipLaunchKernelGGL(bp_peak_copy<false>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, h->stream, dst, src, bytes / 16);
        else hipLaunchKernelGGL(bp_peak_copy<true>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, h->stream, dst, src, bytes / 16);   // plain and nontemporal, best reported
        HIPCHK(hipEventRecord(b, h->stream));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        const float gbs = (float)(2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        if (rep > 0 && gbs > best) best = gbs;
    }
    *hbm_copy_gbs = best;
    (void)hipFree(src); (void)hipFree(dst);
    HIPCHK(hipEventDestroy(a)); HIPCHK(hipEventDestroy(b));
    return BP_OK;
}

// ------------------------------------------------------------------ isolated kernel timing
extern "C" int bp_time_kernel(bp_handle *h, int which, int iters, float *avg_ms)
{
    if (!h || !avg_ms || iters < 1) return fail(BP_ERR_ARG, "bp_time_kernel: bad argument");
    if (h->L < 4 && (which == 0 || which == 1 || which == 2))
        return fail(BP_ERR_ARG, "bp_time_kernel: needs a hidden->hidden layer (numlayers >= 4)");
    if (h->bf) return fail(BP_ERR_STATE, "bp_time_kernel: fp32 kernels only");
    if (h->chunk_frames < h->B || h->windows) return fail(BP_ERR_STATE, "bp_time_kernel: no resident stacked chunk");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int L = h->L, B = h->B;
    hipEvent_t a, b;
    float *scratch_w = nullptr, *scratch_d = nullptr, *scratch_b = nullptr;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    for (int it = -2; it < iters; ++it) {
        if (it == 0) HIPCHK(hipEventRecord(a, h->stream));
        hipError_t er = hipSuccess;
        switch (which) {
        case 0: er = launch_fwd(h, h->stream, 2, B, h->y[1], nullptr, nullptr, true, 1.0f); break;
        case 1: er = launch_dgrad(h, h->stream, 3 < L ? 3 : 2, B); break;
        case 2: case 5: {
            // wgrad + fused update on scratch copies of W / delta (same traffic, state untouched)
            const int l = which == 2 ? 2 : 1;
            const size_t nw = (size_t)h->ld[l - 1] * h->ld[l];
            if (!scratch_w) {
                HIPCHK(hipMalloc((void **)&scratch_w, nw * 4)); HIPCHK(hipMalloc((void **)&scratch_d, nw * 4));
                HIPCHK(hipMalloc((void **)&scratch_b, (size_t)h->ld[l] * 8));
                HIPCHK(hipMemcpyAsync(scratch_w, h->W[l], nw * 4, hipMemcpyDeviceToDevice, h->stream));
                HIPCHK(hipMemsetAsync(scratch_d, 0, nw * 4, h->stream));
                HIPCHK(hipMemsetAsync(scratch_b, 0, (size_t)h->ld[l] * 8, h->stream));
            }
            float *W0 = h->W[l], *D0 = h->dW[l], *b0 = h->b[l], *db0 = h->db[l];
            h->W[l] = scratch_w; h->dW[l] = scratch_d; h->b[l] = scratch_b; h->db[l] = scratch_b + h->ld[l];
            er = launch_wgrad(h, h->stream, l, B, l == 1 ? h->in : h->y[l - 1], true);
            h->W[l] = W0; h->dW[l] = D0; h->b[l] = b0; h->db[l] = db0;
            break;
        }
        case 3: er = launch_fwd(h, h->stream, 1, B, h->in, nullptr, nullptr, true, 1.0f); break;
        case 4: er = launch_fwd(h, h->stream, L - 1, B, h->y[L - 2], h->targ, nullptr, true, 1.0f); break;
        default: HIPCHK(hipEventDestroy(a)); HIPCHK(hipEventDestroy(b));
                 return fail(BP_ERR_ARG, "bp_time_kernel: unknown kernel id");
        }
        HIPCHK(er);
    }
    HIPCHK(hipEventRecord(b, h->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *avg_ms = ms / iters;
    if (scratch_w) { (void)hipFree(scratch_w); (void)hipFree(scratch_d); (void)hipFree(scratch_b); }
    HIPCHK(hipEventDestroy(a)); HIPCHK(hipEventDestroy(b));
    return BP_OK;
}
