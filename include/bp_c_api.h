/*
 * bp_c_api.h -- C ABI of the MI355X-native trainer that replaces the reference's `BP_GPU`
 * object (the frame-wise DNN forward / backward / momentum-update hot path).
 *
 * Every entry point below is what a binding for this path binds.  Each cites the reference
 * interface it replaces (paths relative to the reference tree).  Plain pointers and sizes
 * only; no HIP, torch or C++ types.  All functions return 0 on success and a negative
 * bp_status on failure (the reference instead prints and calls exit(0): BP_GPU.cu:20-24,
 * 929-933 -- the C++ shim include/BP_GPU.h reproduces that convention on top of this ABI).
 * bp_last_error() returns the message of the last failure on the calling thread.
 *
 * Host data layout (identical to the reference):
 *   weights[l][p*cur + c]  (l = 1..numlayers-1, index 0 unused, [prev][cur] row-major)
 *   bias[l][c], in[f*s0 + k], targ[f*sL + d], all fp32 little-endian.
 * Ownership: the library copies in/out; it never keeps or frees caller pointers.  On return
 * from bp_train_chunk / bp_cv_chunk the caller may overwrite `in`/`targ` at once (the
 * reference gives the same guarantee through pageable-memory cublasSetVectorAsync,
 * BP_GPU.cu:274-276); bp_get_weights returns with the data in place (the reference relies
 * on pageable-memory semantics at BP_GPU.cu:920-921).
 */
#ifndef BP_C_API_H
#define BP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_MAXLAYER 10          /* BP_GPU.h:13  (#define MAXLAYER 10)           */
#define BP_MAXCACHEFRAME 200000 /* BP_GPU.h:14  (#define MAXCACHEFRAME 200000)  */

typedef enum {
    BP_OK = 0,
    BP_ERR_ARG = -1,     /* bad argument / configuration                         */
    BP_ERR_DEVICE = -2,  /* HIP runtime error (message has the HIP error string) */
    BP_ERR_STATE = -3,   /* call not valid in the handle's current state         */
    BP_ERR_NOMEM = -4
} bp_status;

/* Constructor arguments of BP_GPU (BP_GPU.h:43-44, BP_GPU.cu:10-12) plus the switches the
 * reference exposes only by editing source.  A zero-initialised tail reproduces the live
 * reference behaviour. */
typedef struct bp_config {
    int   gpu_used;                 /* a_GPU_selected. Reference: device count; here the ranks of
                                       a data-parallel job are separate processes, so this is
                                       informational and must be >= 1                          */
    int   numlayers;                /* number of layer SIZES (2..9, WorkPara limit Interface.h:44) */
    int   layersizes[BP_MAXLAYER];
    int   bunchsize;                /* frames per minibatch handled by THIS device             */
    float lrate, momentum, weightcost;
    int   dropoutflag;              /* 1 = dropout on (BP_GPU.cu:534-551)                      */
    float visible_omit, hid_omit;
    /* ---- extensions (0 = live reference) ---- */
    int   activation;               /* 0 ReLU (DevFunc.cu:67-97 live) | 1 Sigmoid (.bak)        */
    int   momentum_rule;            /* 0 (1-m) rule DevFunc.cu:313-318 | 1 classic :306-311     */
    uint64_t seed;                  /* dropout Philox key (reference: time(NULL), BP_GPU.cu:77) */
    int   device;                   /* HIP device ordinal for this handle                       */
    int   global_bunchsize;         /* data parallel: frames per minibatch over ALL ranks
                                       (0 = bunchsize).  Scales dEdX_L by 2/global and the
                                       update by 1/global                        */
    int   rank_frame_offset;        /* data parallel: index of this rank's first frame inside
                                       the global bunch (keys the dropout stream)               */
    int   max_chunk_frames;         /* capacity of the resident chunk cache; 0 = BP_MAXCACHEFRAME */
    int   compute_dtype;            /* 0 = fp32 everywhere (the reference).  1 = bf16 GEMM operands
                                       (activations, errors, a shadow copy of the weights) with fp32
                                       accumulation, fp32 master weights / momentum / update
                                       (BASELINE.json configs[4]); parity tolerance 2e-2 instead of
                                       1e-4.  Every call works in this mode except
                                       bp_time_kernel (fp32 kernels only)                        */
} bp_config;

typedef struct bp_handle bp_handle;

const char *bp_last_error(void);
/* Library/ABI version and the gfx target the kernels were compiled for ("gfx950"). */
int         bp_abi_version(void);
const char *bp_build_target(void);
/* Number of MI355X devices visible to the process (hipGetDeviceCount), for hosts that do not link the HIP runtime. */
int         bp_device_count(int *n);

/* BP_GPU::BP_GPU (BP_GPU.cu:10-197): select device, allocate device state, upload weights
 * and biases (index 1..numlayers-1).  Momentum state starts at zero (devnew_vf zero-fill,
 * BP_GPU.cu:137-138,938-940). */
int bp_create(const bp_config *cfg, const float *const *weights, const float *const *bias,
              bp_handle **out);

/* BP_GPU::~BP_GPU (BP_GPU.cu:199-239). */
int bp_destroy(bp_handle *h);

/* The reference reads its public members lrate / momentum / weightcost / dropoutflag / visible_omit /
 * hid_omit afresh on every bunch (train_bunch_single: `cur_lrate = lrate`, BP_GPU.cu:488-500), so a caller
 * may change them between chunks.  This pushes new values into the handle; the C++ shim calls it at the
 * start of train() and CrossValid() with the current member values. */
int bp_set_hyper(bp_handle *h, float lrate, float momentum, float weightcost, int dropoutflag,
                 float visible_omit, float hid_omit);

/* BP_GPU::train (BP_GPU.cu:241-331): upload a chunk of n_frames stacked input frames and
 * targets, then run one SGD-momentum step (train_bunch_single, BP_GPU.cu:484-673) per
 * consecutive full bunch; the partial last bunch is ignored (:315-318).  Synchronous with
 * respect to `in`/`targ`. */
int bp_train_chunk(bp_handle *h, int n_frames, const float *in, const float *targ);

/* BP_GPU::CrossValid (BP_GPU.cu:408-479) + cv_bunch_single (:676-773): inference forward
 * (weights scaled by keep when dropoutflag==1), partial bunch processed, returns the SUM of
 * squared errors accumulated in fp32 in frame-major / bin-minor order (:458-467). */
int bp_cv_chunk(bp_handle *h, int n_frames, const float *in, const float *targ,
                float *sq_err_sum);

/* cv_bunch_single (BP_GPU.cu:676-773) exposed for parity tests and batch enhancement:
 * out[n_frames][sL] = network(in) with CV semantics. */
int bp_forward(bp_handle *h, int n_frames, const float *in, float *out);

/* BP_GPU::returnWeights (BP_GPU.cu:910-923). */
int bp_get_weights(bp_handle *h, float *const *weights, float *const *bias);
/* Momentum state (delta_weights/delta_bias, BP_WorkSpace BP_GPU.h:35-36); the reference never
 * downloads it -- provided for parity tests and checkpointing. */
int bp_get_deltas(bp_handle *h, float *const *delta_weights, float *const *delta_bias);

/* ------------------------------------------------------------------------------------
 * Resident-chunk interface: the same step as bp_train_chunk, split so that a caller who
 * already has the chunk in device memory (benchmarks, on-device data producers) can drive
 * bunches without the host upload.  bp_upload_chunk = the H2D part of BP_GPU::train
 * (BP_GPU.cu:269-277); bp_train_resident = the bunch loop (:294-326) over frames
 * [first_frame, first_frame + n_frames) of the resident chunk.  Asynchronous: returns after
 * enqueueing; bp_sync waits. */
int bp_upload_chunk(bp_handle *h, int n_frames, const float *in, const float *targ);
int bp_fill_chunk_synthetic(bp_handle *h, int n_frames, uint64_t seed); /* N(0,1) in/targ on device */
int bp_train_resident(bp_handle *h, int first_frame, int n_frames);
int bp_sync(bp_handle *h);
/* Parity-test entry: bp_train_resident with CALLER-SUPPLIED dropout masks instead of the Philox stream (the reference's
 * masks come from cuRAND seeded by time(NULL), BP_GPU.cu:77-78,534-551, so "same result given the same mask" is the
 * parity statement).  masks[l], l = 0 .. numlayers-2: host [n_frames][layersizes[l]] bytes, 1 = drop the output of
 * layer l for that frame (l = 0: the input frame); NULL = no dropout on that layer.  fp32 single-device handles. */
int bp_train_resident_masked(bp_handle *h, int first_frame, int n_frames, const uint8_t *const *masks);

/* ------------------------------------------------------------------------------------
 * On-device frame stacking.  The reference's reader materialises every
 * sample on the host as `context` consecutive normalised frames [+ the noise-aware block]
 * (Interface.cc:757-790) and uploads context x the raw volume.  Here the caller hands over the
 * RAW normalised frames of the chunk once plus three index tables; they STAY raw in device
 * memory and every bunch (training, CV, forward) stacks -- and, in training, dropout-masks -- its
 * own rows right before its layer-1 kernels: the stacked chunk is never materialised.  Row i of
 * the chunk, as the network sees it, is
 *     in[i]   = fea[win_start[i] .. win_start[i]+context)   (context*fea_dim contiguous floats)
 *               ++ nat[nat_row[i]]                          (fea_dim floats, only when nat != NULL)
 *     targ[i] = targ_frames[targ_frame[i]]                  (layersizes[L-1] floats)
 * i.e. bit-identical to what bp_upload_chunk would have received.  layersizes[0] must equal
 * context*fea_dim (+ fea_dim with a NAT block).  The caller may overwrite everything as soon
 * as the call returns. */
typedef struct bp_window_chunk {
    int n_samples;             /* rows (samples) of the chunk (<= chunk capacity) */
    int n_frames;              /* raw frames in fea / targ_frames */
    int fea_dim, context;
    int n_nat;                 /* rows of nat (0 when nat == NULL) */
    const float *fea;          /* [n_frames][fea_dim], already mean/variance normalised */
    const float *targ_frames;  /* [n_frames][layersizes[L-1]] */
    const float *nat;          /* [n_nat][fea_dim] or NULL */
    const int *win_start;      /* [n_samples] first raw frame of the window */
    const int *targ_frame;     /* [n_samples] raw frame whose target row is used */
    const int *nat_row;        /* [n_samples] or NULL */
} bp_window_chunk;
int bp_upload_chunk_windows(bp_handle *h, const bp_window_chunk *c);             /* then bp_train_resident etc. */
int bp_train_chunk_windows(bp_handle *h, const bp_window_chunk *c);              /* = BP_GPU::train on the stacked chunk */
int bp_cv_chunk_windows(bp_handle *h, const bp_window_chunk *c, float *sq_err_sum); /* = BP_GPU::CrossValid */
int bp_forward_windows(bp_handle *h, const bp_window_chunk *c, float *out);         /* = bp_forward: out[n_samples][sL] (enhancement) */

/* ------------------------------------------------------------------------------------
 * Gradients without the update (parity tests; no reference counterpart -- the reference never
 * exposes layer_ydedx).  bp_grads_resident runs forward + backward of ONE local bunch starting at
 * chunk frame first_frame with the kernels of the data-parallel step and leaves the weight and bias
 * gradients (G_l = y_{l-1}^T . dEdX_l, BP_GPU.cu:642,647: sums over the bunch, not yet divided by n)
 * in a flat fp32 buffer [W_1 | b_1 | W_2 | b_2 ...] of padded rows (bp_grad_layout: offset/count of
 * layer l's segment; W_l is [pad64(prev)][pad64(cur)]); weights, momentum and the dropout stream
 * position stay untouched.  Not valid on an attached handle. */
int bp_grads_resident(bp_handle *h, int first_frame);
int bp_grad_floats(bp_handle *h, size_t *n_floats);
int bp_grad_layout(bp_handle *h, int layer, size_t *offset, size_t *count);
int bp_read_grads(bp_handle *h, float *host_dst, size_t n_floats);
/* Output y_l of hidden layer `layer` (1 .. numlayers-2; layer_y, BP_GPU.h:27: post-activation, post-dropout) for the
 * bunch processed last, [bunchsize][layersizes[layer]] floats.  Parity tests use it to count ReLU decisions that
 * fall within fp32 rounding of zero (they depend on the GEMM's summation order).  fp32 handles only. */
int bp_read_layer_output(bp_handle *h, int layer, float *host_dst, size_t n_floats);

/* ------------------------------------------------------------------------------------
 * In-library data-parallel exchange.  One process per GPU; each rank creates its handle
 * with bunchsize = frames of a minibatch it owns, global_bunchsize = world * bunchsize and
 * rank_frame_offset = rank * bunchsize, then joins the group.  `key` names the job (every rank passes the
 * same string, unique per job on the machine; a POSIX shared-memory block "/bpdp-<key>" carries the
 * rendezvous).  After bp_dp_attach, bp_train_resident / bp_train_chunk / bp_train_chunk_windows run
 * the data-parallel step on this rank's shard of every minibatch: per layer reduce-scatter of the
 * gradients (peer reads over xGMI through hipIpc mappings), momentum update of this rank's 1/world slice
 * of W and delta, all-gather of the new W (peer writes), ordered by device-side flags -- the semantics of
 * the reference's commented-out train_bunch_multi (BP_GPU.cu:775-908: gradient SUM, one update with
 * n = global bunch, identical weights everywhere) without routing through GPU 0.  No collective library
 * and no host synchronisation on the data path.  Every rank must make the same sequence of training
 * calls with the same number of minibatches.  Ranks may share a device (functional testing).
 * bp_get_weights works on every rank (weights are replicated); bp_get_deltas gathers the sharded momentum
 * state and must be called by all ranks together.  A rank that stops responding makes the others fail
 * with BP_ERR_STATE after BP_DP_TIMEOUT_S seconds (default 60) instead of hanging.  bp_set_hyper on an
 * attached handle must be given the same values by every rank.
 *
 * Before the first step relies on it, bp_dp_attach CHECKS the memory-model contract of the exchange on the
 * group's actual devices (peer write-through stores seen by the owner's cached loads; the owner's stores to
 * fine-grained memory seen by peers' system-scope loads), falls back to an explicit acquire behind every wait
 * if the first fails without one (bp_dp_peer_info reports the mode), and fails with BP_ERR_STATE otherwise.
 *
 * bp_dp_attach_ex selects the transport of the same sharded step: BP_DP_TRANSPORT_NATIVE (default: the peer
 * kernels above) or BP_DP_TRANSPORT_RCCL (ncclReduceScatter of every layer's gradient segment, the sharded
 * update, ncclAllGather of the new weights; librccl.so is loaded at run time; world 1, 2, 4 or 8; one rank
 * per device) or BP_DP_TRANSPORT_NATIVE_PUSH (round 6: the native exchange with the reduce-scatter turned round -- every rank
 * WRITES slice r of its gradient segment into rank r's receive buffer with write-through stores, the owner sums its world
 * local slots in the same fixed order; posted writes instead of peer reads on the fabric; same results bit for bit) or
 * BP_DP_TRANSPORT_NATIVE_PUSH_BF16 (the push form with every rank's contribution rounded to bf16 on the way out: half the bytes on the
 * fabric, fp32 summation at the owner in the same order; NOT the default for fp32 nets -- it changes results at the 1e-3 level,
 * tests/test_dp_native.py states and checks its tolerance). */
enum { BP_DP_TRANSPORT_NATIVE = 0, BP_DP_TRANSPORT_RCCL = 1, BP_DP_TRANSPORT_NATIVE_PUSH = 2, BP_DP_TRANSPORT_NATIVE_PUSH_BF16 = 3 };
int bp_dp_attach(bp_handle *h, int world, int rank, const char *key);
int bp_dp_attach_ex(bp_handle *h, int world, int rank, const char *key, int transport);
int bp_dp_detach(bp_handle *h);     /* collective; also done by bp_destroy */
int bp_dp_info(bp_handle *h, int *world, int *rank, unsigned *minibatches);
/* What rank `peer` of the group attached to: HIP device ordinal in ITS process and PCI bus id ("0000:c1:00.0";
 * buffer of >= 16 bytes), plus this group's transport and acquire mode (0 kernel boundary, 1 explicit acquire). */
int bp_dp_peer_info(bp_handle *h, int peer, int *device, char *pci_bus_id, int len, int *transport, int *acquire_mode);
/* How this group hands a layer's gradient segment to the exchange: 1 = inside the running weight-gradient launch (its tiles
 * count themselves behind system-scope write-through stores; native transport, bunches of 128 / 256 / 512 frames, fp32, at most
 * 4 weight layers = one grouped launch), 0 = event + kernel boundary per group of layers (RCCL, bf16, other bunch sizes, deeper
 * nets, or a group -- of one rank too -- whose attach-time self-test
 * found that form unusable on its devices, e.g. streams that do not run concurrently).  The same on every rank. */
int bp_dp_handoff(bp_handle *h, int *in_kernel);
/* Host-side barrier / all-gather of one small record (<= 64 bytes) per rank over the group's rendezvous block, for
 * launchers that have no other channel between the ranks (bench.py: barrier around the timed region, max over ranks). */
int bp_dp_barrier(bp_handle *h);
int bp_dp_allgather(bp_handle *h, const void *mine, size_t bytes, void *all);

/* The rendezvous underneath bp_dp_attach, usable on its own and WITHOUT a GPU (host code only): a POSIX
 * shared-memory block "/bpdp-<key>" created by rank 0, joined by ranks 1..world-1 within timeout_s, unlinked as
 * soon as everyone has joined (nothing is left behind by a later crash; a stale block of a crashed job with the
 * same key is ignored and replaced).  Replaces what the reference would need an MPI/NCCL bootstrap for; the
 * reference itself is single-process (BP_GPU.cu:29-36). */
typedef struct bp_rdv bp_rdv;
int bp_rdv_open(const char *key, int world, int rank, double timeout_s, bp_rdv **out);
int bp_rdv_barrier(bp_rdv *r);
int bp_rdv_allgather(bp_rdv *r, const void *mine, size_t bytes, void *all);   /* bytes <= 64 */
int bp_rdv_close(bp_rdv *r);
/* Pin / unpin caller-owned host memory (hipHostRegister) so that uploads from it (bp_train_chunk[_windows], bp_upload_*)
 * are DMA transfers instead of staged pageable copies; optional -- every entry point accepts pageable memory, which is what
 * the reference's callers have (`new[]`, Interface.cc:401-403; the reference stages through pinned memory itself,
 * BP_GPU.cu:926-992).  For hosts that do not link the HIP runtime. */
int bp_host_register(void *p, size_t bytes);
int bp_host_unregister(void *p);
/* PCI bus id of a visible device (hipDeviceGetPCIBusId), for hosts that do not link the HIP runtime. */
int bp_device_pci_bus_id(int device, char *buf, int len);

/* Timing of the dominant kernels for roofline reporting: average duration (ms) of the last
 * bp_train_resident call's whole bunch loop measured with HIP events on the handle's stream. */
int bp_last_train_ms(bp_handle *h, float *ms, int *bunches);
/* Time `iters` launches of one kernel of the step in isolation (HIP events on the handle's
 * stream).  which: 0 fwd hidden GEMM(layer 2), 1 dgrad hidden, 2 wgrad+update hidden,
 * 3 fwd layer 1, 4 fwd output layer, 5 wgrad+update layer 1.  Returns average ms. */
int bp_time_kernel(bp_handle *h, int which, int iters, float *avg_ms);

/* The same bunch loop as bp_train_resident over n_bunches bunches with a HIP event recorded after every launch
 * on the launch stream: avg_ms[BP_PROF_*] = average duration of one launch of that class AS IT RUNS INSIDE THE
 * STEP (previous event -> own event, i.e. the kernel plus the dependent-launch boundary in front of it),
 * launches_per_step[] = how many launches of the class a step makes.  Trains like bp_train_resident does. */
enum { BP_PROF_FWD_L1 = 0, BP_PROF_FWD_HIDDEN = 1, BP_PROF_FWD_OUT = 2 /* split-K GEMM with its reduce, one launch */, BP_PROF_DGRAD_OUT = 3,
       BP_PROF_DGRAD_HIDDEN = 4, BP_PROF_WGRAD = 5 /* every layer's wgrad + update: one grouped launch */, BP_PROF_KINDS = 6 };
int bp_profile_step(bp_handle *h, int first_frame, int n_bunches, float *avg_ms, int *launches_per_step);
/* Peaks measured on this device in this process: bare v_mfma_f32_32x32x2_f32 loop (TFLOP/s) and a float4
 * device copy of 1 GiB (GB/s, read + write bytes), to stand beside the vendor figures in a roofline report. */
int bp_measure_peaks(bp_handle *h, float *mfma_f32_tflops, float *hbm_copy_gbs);

#ifdef __cplusplus
}
#endif
#endif /* BP_C_API_H */
