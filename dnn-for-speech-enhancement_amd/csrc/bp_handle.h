// bp_handle.h -- what the translation units of libbp_hip.so share: the device state of one BP_GPU replacement object
// (bp_handle), error plumbing, and the "step operations" that bp_step.hip exports to the data-parallel driver
// (bp_dp.hip) and the measurement entry points (bp_profile.hip).  Internal: nothing in here is part of the C ABI
// (include/bp_c_api.h).
//
//   bp_step.hip     handle construction, chunk interface, every kernel launch of the training / CV / forward step
//   bp_dp.hip       in-library data-parallel exchange (rendezvous, hipIpc peers, RCCL transport, the sharded step driver)
//   bp_profile.hip  in-step event profile, measured peaks, isolated kernel timing
//
// Device layout (all fp32 unless a bf16 copy is named): every layer width s_l is padded to ld_l = roundup(s_l, 64); pad
// columns/rows are zero and stay zero under the step (DESIGN.md "padding invariants"), so the GEMM tiles never need
// column predicates and every row is 256-byte aligned.
//   W_l   [ld_{l-1}][ld_l]   (reference layout weights[l][p*cur+c], BP_GPU.cu:139)
//   y_l   [B][ld_l]          post-activation, post-dropout output of layer l (layer_y)
//   dx_l  [B][ld_l]          dE/dx of layer l (layer_dedx); layer_x/dydx/dedy are never stored
//   in    [cap][ld_0], targ [cap][ld_{L-1}]   resident chunk (dev.in/dev.targ, BP_GPU.cu:127-130)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/bp_c_api.h"

typedef uint16_t bf16_t;

extern thread_local std::string g_bp_err;
static inline int fail(int code, const std::string &msg) { g_bp_err = msg; return code; }
#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t _e = (x);                                                                          \
        if (_e != hipSuccess)                                                                         \
            return fail(BP_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(_e));               \
    } while (0)

static inline int pad64(int x) { return (x + 63) & ~63; }

// Development switches (A/B aids of the measurements quoted in DESIGN.md): environment variables that only a library
// built with -DBP_DEV (`make dev` -> libbp_hip_dev.so, loaded through BP_HIP_LIB) reads.  The shipped library has ONE
// code path per shape and reads no environment except BP_DP_TIMEOUT_S.
#ifdef BP_DEV
static inline bool dev_flag(const char *name) { return getenv(name) != nullptr; }
static inline int dev_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#endif

struct StepProf;
struct bp_dp;

struct bp_handle {
    bp_config cfg;
    int L;                       // number of layer sizes
    int s[BP_MAXLAYER], ld[BP_MAXLAYER];
    int B, Bg;                   // local / global bunch
    int cap, chunk_frames;
    hipStream_t own_stream, stream;
    // parameters and momentum state live in two flat arenas with the layout of the flat gradient buffer
    // ([W_1|b_1|W_2|b_2|...], padded; g_off/g_cnt) so that data-parallel ranks can export them as ONE hipIpc
    // allocation each and the sharded update is a flat elementwise pass (bp_dp.h); W/b/dW/db point into them
    float *params, *deltas;
    float *W[BP_MAXLAYER], *b[BP_MAXLAYER], *dW[BP_MAXLAYER], *db[BP_MAXLAYER];
    bp_dp *dp;                   // attached data-parallel group (bp_dp_attach) or null
    StepProf *prof;              // bp_profile_step in progress: an event after every launch of the step
    const uint8_t *inj_mask[BP_MAXLAYER];   // bp_train_resident_masked in progress: device masks of this bunch per layer output
    const float *inj_x0;                    // ... and the masked copy of its input rows
    float *y[BP_MAXLAYER], *dx[BP_MAXLAYER];
    float *in, *targ, *out_dev;
    float *slabs; size_t slab_stride; int out_splits;   // split-K workspace of the output layer
    unsigned *out_ticket;                               // ... and its ticket words (one per 32 x 32 output tile)
    float *grad; size_t grad_floats; size_t g_off[BP_MAXLAYER], g_cnt[BP_MAXLAYER];
    float *host_out;             // pinned staging for CV outputs (grow-only, whole chunk)
    float *out_chunk;            // device: network outputs of a whole chunk [frames][ld_L] (CV / forward), grow-only
    size_t out_chunk_frames;
    uint32_t step;               // bunches trained so far (dropout stream position)
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
                                              // of bunch i stacks bunch i+1 into the other (bp_out_split_stage)
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
    float *bf_ks_slab; unsigned *bf_ks_cnt;                  // split-k output forward (bp_bf16.h, KS): partial tiles and ticket words, or null
};

// Every device buffer gets SLACK floats of zeroed tail so that whole-tile reads of the GEMM loaders (no predicates,
// see GemmArgs) stay inside the allocation.
static const size_t SLACK = 4096;
int dev_alloc(bp_handle *h, float **p, size_t n_floats);

// bp_profile_step: one HIP event after every launch of the step on the launch stream; the duration attributed to a
// launch is the time between the previous event and its own (= kernel + the dependent-launch boundary in front of it).
struct StepProf {
    std::vector<hipEvent_t> ev; std::vector<int> kind; size_t used;
};

// ------------------------------------------------------------------ step operations (bp_step.hip)
// One bunch starting at chunk frame `first`: forward + backward; fused: momentum update inside the wgrad epilogues
// (train_bunch_single, BP_GPU.cu:484-673), else gradients to the flat buffer.
hipError_t bunch(bp_handle *h, int first, bool fused);
// The pieces of a bunch, for the driver that cuts the step at the gradient exchange (bp_dp.hip):
hipError_t step_inputs(bp_handle *h, int first, const float **x0, const float **tg);   // stage / mask the bunch's rows; where they lie
hipError_t step_forward(bp_handle *h, int l, const float *x0, const float *tg);        // training forward of weight layer l (bf16: converts the input rows at l == 1)
hipError_t step_dgrad(bp_handle *h, int l);                                             // dEdX_{l-1} from dEdX_l and the pre-update W_l
// weight + bias gradients of layers ls[0..n) into the flat gradient buffer, ONE grouped launch where the kernel set allows it.
// done != null: done[l] is a device counter every tile of layer l's segment bumps behind its stores (in-kernel hand-off);
// only legal when step_wgrads_count(h) says the launch really counts.
hipError_t step_wgrads_store(bp_handle *h, const int *ls, int n, const float *x0, unsigned *const *done);
bool step_wgrads_count(const bp_handle *h);
unsigned step_wgrad_tiles(const bp_handle *h, int l);                                   // tiles of layer l in that launch
hipError_t step_shadow(bp_handle *h, int l);                                            // fp32 master W_l -> bf16 shadow (bf16 mode)
bool step_stages(const bp_handle *h);                                                   // the bunch's input rows go through the staged tile (window chunk | visible dropout)
// single launches (bp_profile.hip: isolated kernel timing)
hipError_t launch_fwd(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, const float *targ, float *out, bool train, float alpha);
hipError_t launch_dgrad(bp_handle *h, hipStream_t st, int l, int M);
hipError_t launch_wgrad(bp_handle *h, hipStream_t st, int l, int M, const float *y_prev, bool fused);
hipError_t prof_mark(bp_handle *h, int kind);

// ------------------------------------------------------------------ data-parallel driver (bp_dp.hip)
int dp_check(bp_handle *h);                       // BP_OK, or the device-side timeout an exchange kernel raised
hipError_t dp_bunch(bp_handle *h, int first);
hipError_t dp_flush(bp_handle *h);
int dp_gather_deltas(bp_handle *h);
bool dp_gathers_deltas(const bp_handle *h);       // attached with more than one rank: the momentum state is sharded
