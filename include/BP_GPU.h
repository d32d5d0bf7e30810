/*
 * BP_GPU.h -- drop-in replacement for the reference's BP_GPU.h (class BP_GPU, BP_GPU.h:40-88)
 * on top of the MI355X C ABI (bp_c_api.h / libbp_hip.so).  Header-only; no CUDA, HIP, cuBLAS or
 * cuRAND types leak into the caller, so BPtrain.cc / Interface.cc-style code compiles against it
 * with a plain C++ compiler and links with -lbp_hip.
 *
 * Kept verbatim from the reference (BP_GPU.h:13-14, 43-70): the macros MAXLAYER and
 * MAXCACHEFRAME, the constructor signature, train / CrossValid / returnWeights, and the public
 * data members.  Error convention of the reference is reproduced: a message on stdout and
 * exit(0) (BP_GPU.cu:20-24, 929-933).
 *
 * Differences a caller can observe (all documented in DESIGN.md):
 *   - `a_GPU_selected` > 1 does not fan out inside one process (the reference's multi-GPU path
 *     is commented out and trains nothing, BP_GPU.cu:301-313); data parallelism is one process
 *     per GPU: the bp_config constructor below joins a group (bptrain gpu_used=N forks the ranks).
 *   - the public members lrate / momentum / weightcost / dropoutflag / visible_omit / hid_omit are pushed into the
 *     library at the start of every train / CrossValid call (the reference reads them per bunch, BP_GPU.cu:488-500).
 *   - train_bunch_single / train_bunch_multi / cv_bunch_single take DEVICE pointers in the
 *     reference and are only called from inside the class; they are not re-exported.
 *   - train_windows / CrossValid_windows are additions (on-device frame stacking).
 *   - optional behaviour switches the reference has only as source edits can be set through
 *     environment variables before construction: BP_ACTIVATION=sigmoid|relu,
 *     BP_MOMENTUM_RULE=classic|live, BP_SEED=<u64>, BP_DEVICE=<ordinal>, BP_COMPUTE_DTYPE=fp32|bf16.
 */
#ifndef BP_GPU_SHIM_H
#define BP_GPU_SHIM_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bp_c_api.h"

#define MAXLAYER 10           /* BP_GPU.h:13 */
#define MAXCACHEFRAME 200000  /* BP_GPU.h:14 */

class BP_GPU
{
public:
    /* BP_GPU.h:43-44 / BP_GPU.cu:10-12 */
    BP_GPU(int a_GPU_selected, int a_numlayers, int *a_layersizes, int a_bunchsize, float a_lrate, float a_momentum,
           float a_weightcost, float **weights, float **bias, int a_dropoutflag, float a_visible_omit, float a_hid_omit)
        : handle_(0)
    {
        if (a_GPU_selected < 1) {                       /* BP_GPU.cu:20-24 */
            printf("GPU Num %d Not In Range %d-\n", a_GPU_selected, 1);
            exit(0);
        }
        printf("Use GPU Device : %d\n", a_GPU_selected);
        bp_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.gpu_used = a_GPU_selected;
        cfg.numlayers = a_numlayers;
        for (int i = 0; i < a_numlayers && i < MAXLAYER; ++i) cfg.layersizes[i] = a_layersizes[i];
        cfg.bunchsize = a_bunchsize;
        cfg.lrate = a_lrate; cfg.momentum = a_momentum; cfg.weightcost = a_weightcost;
        cfg.dropoutflag = a_dropoutflag; cfg.visible_omit = a_visible_omit; cfg.hid_omit = a_hid_omit;
        /* switches the reference has only as source edits: an UNMODIFIED caller (the reference's BPtrain.cc) can
         * still reach them through the environment; callers that can be edited use the bp_config constructor below */
        const char *e;
        if ((e = getenv("BP_ACTIVATION")) != 0) cfg.activation = strcmp(e, "sigmoid") == 0 ? 1 : 0;
        if ((e = getenv("BP_MOMENTUM_RULE")) != 0) cfg.momentum_rule = strcmp(e, "classic") == 0 ? 1 : 0;
        if ((e = getenv("BP_SEED")) != 0) cfg.seed = strtoull(e, 0, 10);
        if ((e = getenv("BP_DEVICE")) != 0) cfg.device = atoi(e);
        if ((e = getenv("BP_COMPUTE_DTYPE")) != 0) cfg.compute_dtype = strcmp(e, "bf16") == 0 ? 1 : 0;
        init(cfg, weights, bias, 1, 0, 0);
    }
    /* Extension: explicit configuration (every bp_config field: activation, momentum rule, dropout seed, device,
     * compute dtype, data-parallel geometry) instead of process environment, so that several trainers can coexist in one
     * process.  dp_world > 1: this object is rank dp_rank of a data-parallel group of dp_world processes (one per
     * GPU) named dp_key (required, unique per job on the machine); cfg.bunchsize is then the frames of a minibatch THIS rank owns and the constructor fills in
     * global_bunchsize / rank_frame_offset and joins the group (bp_dp_attach, bp_c_api.h). */
    BP_GPU(const bp_config &a_cfg, float **weights, float **bias, int dp_world = 1, int dp_rank = 0, const char *dp_key = 0)
        : handle_(0)
    {
        bp_config cfg = a_cfg;
        if (dp_world > 1) { cfg.global_bunchsize = cfg.bunchsize * dp_world; cfg.rank_frame_offset = cfg.bunchsize * dp_rank; }
        init(cfg, weights, bias, dp_world, dp_rank, dp_key);
    }
    ~BP_GPU() { bp_destroy(handle_); }

    /* BP_GPU.h:50 / BP_GPU.cu:241-331 */
    void train(int n_frames, float *in, const float *targ) { push_hyper(); check(bp_train_chunk(handle_, n_frames, in, targ)); }
    /* BP_GPU.h:57 / BP_GPU.cu:408-479: returns the SUM of squared errors */
    float CrossValid(int n_frames, const float *in, const float *targ)
    {
        float e = 0.0f;
        push_hyper();
        check(bp_cv_chunk(handle_, n_frames, in, targ, &e));
        return e;
    }
    /* Extensions (no reference counterpart): the same two calls with the frame stacking done on the
     * device from raw frames + index tables (bp_window_chunk, bp_c_api.h). */
    void train_windows(const bp_window_chunk &c) { push_hyper(); check(bp_train_chunk_windows(handle_, &c)); }
    float CrossValid_windows(const bp_window_chunk &c)
    {
        float e = 0.0f;
        push_hyper();
        check(bp_cv_chunk_windows(handle_, &c, &e));
        return e;
    }
    /* leave the data-parallel group (collective over its ranks); the weights stay valid on every rank */
    void dp_detach() { check(bp_dp_detach(handle_)); }
    /* BP_GPU.h:60 / BP_GPU.cu:910-923 */
    void returnWeights(float **weights, float **bias) { check(bp_get_weights(handle_, weights, bias)); }

    int numlayers;                 /* BP_GPU.h:62-70 */
    int layersizes[MAXLAYER];
    int bunchsize;
    float lrate;
    float momentum;
    float weightcost;
    int dropoutflag;
    float visible_omit;
    float hid_omit;

private:
    BP_GPU(const BP_GPU &);
    BP_GPU &operator=(const BP_GPU &);
    void init(const bp_config &cfg, float **weights, float **bias, int dp_world, int dp_rank, const char *dp_key)
    {
        numlayers = cfg.numlayers; bunchsize = cfg.bunchsize; lrate = cfg.lrate; momentum = cfg.momentum;
        weightcost = cfg.weightcost; dropoutflag = cfg.dropoutflag; visible_omit = cfg.visible_omit; hid_omit = cfg.hid_omit;
        for (int i = 0; i < MAXLAYER; ++i) layersizes[i] = i < cfg.numlayers ? cfg.layersizes[i] : 0;
        check(bp_create(&cfg, weights, bias, &handle_));
        if (dp_world > 1) {
            /* no default key: two jobs on one machine sharing a key would meet in each other's rendezvous */
            if (!dp_key || !*dp_key) { printf("BP_GPU: a data-parallel group needs a job key (dp_key)\n"); exit(0); }
            check(bp_dp_attach(handle_, dp_world, dp_rank, dp_key));
        }
        printf("Created net with %d layers, bunchsize %d.\n", numlayers, bunchsize);   /* BP_GPU.cu:196 */
    }
    /* the reference reads these members afresh on every bunch (BP_GPU.cu:488-500): a caller may assign them between chunks */
    void push_hyper() { check(bp_set_hyper(handle_, lrate, momentum, weightcost, dropoutflag, visible_omit, hid_omit)); }
    void check(int rc)
    {
        if (rc != 0) {             /* reference convention: message + exit(0) */
            printf("%s\n", bp_last_error());
            exit(0);
        }
    }
    bp_handle *handle_;
};

#endif /* BP_GPU_SHIM_H */
