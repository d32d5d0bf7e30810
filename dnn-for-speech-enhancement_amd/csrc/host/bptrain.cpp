// bptrain.cpp -- command-line compatible replacement of the reference executable `BPtrain`
// ("next" row N2 of SURVEY.md 8f): same `name=value` arguments (Interface.cc:89-244), same log lines,
// same weight-file bytes, one epoch = train over the chunks of train_sent_range in shuffled order,
// save weights, cross-validate over cv_sent_range (BPtrain.cc:16-101).  The trainer behind it is the
// MI355X library through the drop-in class include/BP_GPU.h; the Perl epoch driver
// (finetune_DNN_speech_enhancement_dropout_NAT.pl) can call this binary unchanged.
//
// gpu_used=N (N > 1) is data parallel, which the reference only has as commented-out code (BP_GPU.cu:29-36, 775-908):
// the process forks N ranks, one per GPU (rank r on device r % visible devices); `bunchsize` is the GLOBAL minibatch as
// in the reference's per-GPU split (BP_GPU.cu:29-36), every rank trains on its bunchsize/N frames of each minibatch and
// the library exchanges gradients / weights itself (bp_dp_attach, include/bp_c_api.h).  Rank 0 writes the log, the
// weights file and runs the cross-validation.  ONE reader per node, as in the reference (one Interface feeding all
// devices, Interface.cc:689-861, BP_GPU.cu:269-277): the ranks share a two-slot chunk ring in shared memory
// (chunk_ring.h); rank 0 plans and shuffles, every rank converts 1/N of each chunk's frames, every Pfile byte is read
// once per node, and each rank uploads the raw frames plus only ITS rows' index tables.
//
// Extra optional keys (defaults = live reference behaviour): activation=relu|sigmoid,
// momentum_rule=live|classic, seed=<u64> (dropout stream), device=<ordinal>, compute=fp32|bf16;
// stack=device|host (default device: raw frames + index tables go to the GPU, which builds the context
// windows -- 11x less host work and upload, identical samples; host = the reference's Readchunk layout),
// prefetch=1|0 (default 1: the next chunk is read while the current one is uploaded / trained).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <functional>
#include <future>
#include <string>
#include <vector>

#include "../../../include/BP_GPU.h"
#include "chunk_ring.h"
#include "pfile_reader.h"
#include "wts_io.h"

#include <thread>

struct Params {
    std::string fea_file, norm_file, targ_file, outwts_file, log_file, initwts_file, train_range, cv_range;
    int fea_dim = 0, fea_context = 0, targ_offset = 0, dropoutflag = 0, traincache = 0, bunchsize = 0, gpu_used = 1;
    int seed = 0, numlayers = 0, layersizes[MAXLAYER] = {0};
    float momentum = 0, weightcost = 0, lrate = 0, visible_omit = 0, hid_omit = 0;
    float wmin = -0.1f, wmax = 0.1f, bmin = -0.1f, bmax = 0.1f;          // Interface.cc:79-82
    bool stack_on_device = true, prefetch = true;
    int activation = 0, momentum_rule = 0, device = -1, compute_dtype = 0;
    unsigned long long dropout_seed = 0;
};

using bp::shard_rows;

// A rank that leaves through exit() -- every error path of this file and of the reader does, the reference's convention --
// raises the ring's abort flag on its way out, so that its peers fail at once instead of waiting for it.
static bp::ChunkRing *g_ring = nullptr;
static bool g_ring_finished = false;
static void ring_abort_at_exit() { if (g_ring && !g_ring_finished) g_ring->abort(); }

typedef bp::PfileReader::WindowChunk WindowChunk;
static bp_window_chunk describe(const WindowChunk &w, int context)
{
    bp_window_chunk c;
    memset(&c, 0, sizeof(c));
    c.n_samples = w.n_samples; c.n_frames = w.n_frames; c.fea_dim = w.fea_dim; c.context = context;
    c.n_nat = w.n_nat();
    c.fea = w.fea.data(); c.targ_frames = w.targ.data(); c.nat = w.nat.empty() ? nullptr : w.nat.data();
    c.win_start = w.win_start.data(); c.targ_frame = w.targ_frame.data(); c.nat_row = w.nat_row.empty() ? nullptr : w.nat_row.data();
    return c;
}

// Chunks of one plan, in the given order, read one ahead of the consumer on a helper thread (reads stay strictly
// sequential, so the lrand48 stream is consumed in the reference's order).
class ChunkStream {
public:
    // pin: the slots are DMA sources (stack=device: bp_*_windows uploads straight out of them).  With stack=host they are not --
    // expand() copies into the caller's stacked buffers -- and pinning them would lock hundreds of MB and cost the
    // registration time for nothing.
    ChunkStream(bp::PfileReader &r, const bp::PfileReader::Plan &p, const std::vector<int> &order, bool shuffle, bool prefetch, bool pin)
        : r_(r), p_(p), order_(order), shuffle_(shuffle), prefetch_(prefetch)
    {
        // both slots get their final capacity now and are pinned (bp_host_register): the uploads of bp_*_windows are then DMA
        // transfers on the copy engines, not staged pageable copies that compete with the training kernels for the CUs
        size_t fcap = 0;
        for (size_t c = 0; c < p.chunk_frame_st.size(); ++c) { const size_t f = (size_t)r.chunk_shape(p, (int)c).n_frames; if (f > fcap) fcap = f; }
        for (auto &s : slot_) {
            s.fea.reserve(fcap * r.fea_dim()); s.targ.reserve(fcap * r.out_dim());
            if (!pin || !fcap) continue;
            for (std::vector<float> *v : {&s.fea, &s.targ}) {
                if (bp_host_register(v->data(), v->capacity() * sizeof(float)) == 0) pinned_.push_back(v->data());
                else fprintf(stderr, "bptrain: could not pin a %zu MB read-ahead slot (%s); its uploads will be staged copies\n",
                             v->capacity() * sizeof(float) >> 20, bp_last_error());
            }
        }
        if (prefetch_ && !order_.empty()) start(0);
    }
    ~ChunkStream()
    {
        if (fut_.valid()) fut_.wait();
        for (void *q : pinned_) bp_host_unregister(q);
    }
    const WindowChunk &get(int i)
    {
        // (a reader error found by the read-ahead thread is reported HERE, on the main thread: print + exit(0) from a helper
        // thread would run the exit handlers while this thread is still inside the library)
        std::string err;
        if (prefetch_) {
            err = fut_.get();
            if (err.empty() && i + 1 < (int)order_.size()) start(i + 1);
        } else {
            err = r_.try_read_chunk_windows(p_, order_[i], shuffle_, slot_[i & 1]);
        }
        if (!err.empty()) bp::die("%s", err.c_str());
        return slot_[i & 1];
    }
private:
    void start(int i)
    {
        fut_ = std::async(std::launch::async, [this, i] { return r_.try_read_chunk_windows(p_, order_[i], shuffle_, slot_[i & 1]); });
    }
    bp::PfileReader &r_; const bp::PfileReader::Plan &p_; std::vector<int> order_; bool shuffle_, prefetch_;
    WindowChunk slot_[2];
    std::vector<void *> pinned_;
    std::future<std::string> fut_;
};

static void parse_range(const std::string &r, int *st, int *en, FILE *log)
{
    const size_t p = r.find('-');
    if (p == std::string::npos) { fprintf(log, "sent range: %s format error.\n", r.c_str()); exit(0); }
    *st = atoi(r.substr(0, p).c_str());
    *en = atoi(r.substr(p + 1).c_str());
}

int main(int argc, char **argv)
{
    const double t_start = (double)time(NULL);
    Params P;
    for (int i = 1; i < argc; ++i) {
        char *eq = strchr(argv[i], '=');
        if (!eq) { printf("Arg: %s  Format Error\n", argv[i]); exit(0); }
        const std::string k(argv[i], eq - argv[i]), v(eq + 1);
        if (k == "fea_file") P.fea_file = v; else if (k == "norm_file") P.norm_file = v;
        else if (k == "targ_file") P.targ_file = v; else if (k == "outwts_file") P.outwts_file = v;
        else if (k == "log_file") P.log_file = v; else if (k == "initwts_file") P.initwts_file = v;
        else if (k == "train_sent_range") P.train_range = v; else if (k == "cv_sent_range") P.cv_range = v;
        else if (k == "fea_dim") P.fea_dim = atoi(v.c_str()); else if (k == "fea_context") P.fea_context = atoi(v.c_str());
        else if (k == "targ_offset") P.targ_offset = atoi(v.c_str()); else if (k == "dropoutflag") P.dropoutflag = atoi(v.c_str());
        else if (k == "traincache") P.traincache = atoi(v.c_str()); else if (k == "bunchsize") P.bunchsize = atoi(v.c_str());
        else if (k == "gpu_used") P.gpu_used = atoi(v.c_str()); else if (k == "init_randem_seed") P.seed = atoi(v.c_str());
        else if (k == "momentum") P.momentum = (float)atof(v.c_str()); else if (k == "weightcost") P.weightcost = (float)atof(v.c_str());
        else if (k == "lrate") P.lrate = (float)atof(v.c_str()); else if (k == "visible_omit") P.visible_omit = (float)atof(v.c_str());
        else if (k == "hid_omit") P.hid_omit = (float)atof(v.c_str());
        else if (k == "init_randem_weight_min") P.wmin = (float)atof(v.c_str()); else if (k == "init_randem_weight_max") P.wmax = (float)atof(v.c_str());
        else if (k == "init_randem_bias_min") P.bmin = (float)atof(v.c_str()); else if (k == "init_randem_bias_max") P.bmax = (float)atof(v.c_str());
        else if (k == "layersizes") {
            P.numlayers = 0;
            size_t pos = 0;
            while (P.numlayers < MAXLAYER) {
                const size_t c = v.find(',', pos);
                P.layersizes[P.numlayers++] = atoi(v.substr(pos, c == std::string::npos ? c : c - pos).c_str());
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        }
        // switches the reference only has as source edits (handed to the shim in its bp_config constructor)
        else if (k == "activation") P.activation = v == "sigmoid" ? 1 : 0;
        else if (k == "momentum_rule") P.momentum_rule = v == "classic" ? 1 : 0;
        else if (k == "seed") P.dropout_seed = strtoull(v.c_str(), 0, 10);
        else if (k == "device") P.device = atoi(v.c_str());
        else if (k == "compute") P.compute_dtype = v == "bf16" ? 1 : 0;         // fp32 (default) | bf16
        else if (k == "stack") P.stack_on_device = (v != "host");
        else if (k == "prefetch") P.prefetch = atoi(v.c_str()) != 0;
        // unknown names are silently ignored, as in the reference (e.g. the .pl passes numlayers=)
    }
    // ---- gpu_used > 1: data-parallel ranks, forked further down -- after the parent has opened the files, initialised the
    // weights, planned and shuffled the chunks and mapped the shared chunk ring (all inherited), but before anything
    // touches the GPU
    const int world = P.gpu_used > 1 ? P.gpu_used : 1;
    int rank = 0;
    std::vector<pid_t> kids;
    const std::string dp_key = "bptrain-" + std::to_string((long)getpid());
    if (world > 1 && (world > 8 || P.bunchsize % world != 0)) {
        printf("gpu_used=%d: needs 2..8 GPUs and a bunchsize that is a multiple of it\n", world);
        exit(0);
    }
    bool lead = true;
    FILE *log = fopen(P.log_file.c_str(), "wt");
    if (!log) { printf("can not open output log file: %s\n", P.log_file.c_str()); exit(0); }
    FILE *fp_out = fopen(P.outwts_file.c_str(), "wb");
    if (!fp_out) { fprintf(log, "can not open output weights file: %s\n", P.outwts_file.c_str()); exit(0); }
    const int L = P.numlayers;
    if (L < 2 || L > MAXLAYER - 1) { fprintf(log, "layersizes: need 2..%d layer sizes\n", MAXLAYER - 1); exit(0); }
    // parameter echo (Interface.cc:267-298)
    fprintf(log, "parameters input:\n");
    fprintf(log, "fea_file:             %s\n", P.fea_file.c_str());
    fprintf(log, "norm_file:            %s\n", P.norm_file.c_str());
    fprintf(log, "targ_file:            %s\n", P.targ_file.c_str());
    fprintf(log, "outwts_file:          %s\n", P.outwts_file.c_str());
    fprintf(log, "log_file:\t\t          %s\n", P.log_file.c_str());
    fprintf(log, "initwts_file:         %s\n", P.initwts_file.c_str());
    fprintf(log, "train_sent_range:     %s\n", P.train_range.c_str());
    fprintf(log, "cv_sent_range:        %s\n", P.cv_range.c_str());
    fprintf(log, "fea_dim:\t\t          %d\n", P.fea_dim);
    fprintf(log, "fea_context:\t\t      %d\n", P.fea_context);
    fprintf(log, "bunchsize:\t\t        %d\n", P.bunchsize);
    fprintf(log, "gpu_used:\t\t          %d\n", P.gpu_used);
    fprintf(log, "train_cache:\t\t      %d\n", P.traincache);
    fprintf(log, "init_randem_seed:\t\t  %d\n", P.seed);
    fprintf(log, "targ_offset:\t\t      %d\n", P.targ_offset);
    fprintf(log, "dropoutflag:\t\t      %d\n", P.dropoutflag);
    fprintf(log, "init_randem_weight_max:\t\t  %f\n", P.wmax);
    fprintf(log, "init_randem_weight_min:\t\t  %f\n", P.wmin);
    fprintf(log, "init_randem_bias_max:\t\t    %f\n", P.bmax);
    fprintf(log, "init_randem_bias_min:\t\t    %f\n", P.bmin);
    fprintf(log, "momentum:\t\t                %f\n", P.momentum);
    fprintf(log, "weightcost:\t\t              %f\n", P.weightcost);
    fprintf(log, "learnrate:\t\t              %f\n", P.lrate);
    fprintf(log, "visible_omit:\t\t      %f\n", P.visible_omit);
    fprintf(log, "hid_omit:\t\t      %f\n", P.hid_omit);
    fprintf(log, "layersizes:\t\t              ");
    for (int j = 0; j < L; ++j) fprintf(log, "%d,", P.layersizes[j]);
    fprintf(log, "\nPlease check...\n");

    bp::ReaderConfig rc;
    rc.fea_file = P.fea_file; rc.targ_file = P.targ_file; rc.norm_file = P.norm_file;
    rc.fea_dim = P.fea_dim; rc.fea_context = P.fea_context; rc.targ_offset = P.targ_offset;
    rc.out_dim = P.layersizes[L - 1]; rc.traincache = P.traincache; rc.input_dim = P.layersizes[0];
    if (P.traincache < 1 || P.traincache > MAXCACHEFRAME) { fprintf(log, "traincache must be in 1..%d\n", MAXCACHEFRAME); exit(0); }
    bp::PfileReader reader(rc);
    fprintf(log, "Loading Norm file...\n");
    reader.open();
    fprintf(log, "Norm file loaded.\n");

    std::vector<std::vector<float>> Wv(L), Bv(L);
    float *weights[MAXLAYER] = {0}, *bias[MAXLAYER] = {0};
    for (int i = 1; i < L; ++i) {
        Wv[i].assign((size_t)P.layersizes[i] * P.layersizes[i - 1], 0.f); Bv[i].assign(P.layersizes[i], 0.f);
        weights[i] = Wv[i].data(); bias[i] = Bv[i].data();
    }
    srand48(P.seed);                                    // once, for weights and every shuffle (Interface.cc:338)
    if (P.initwts_file.empty()) {
        fprintf(log, "Getting Randemed initial weights...\n");
        bp::random_weights(L, P.layersizes, weights, bias, P.wmin, P.wmax, P.bmin, P.bmax);
        fprintf(log, "Randemed initial weights getted.\n");
    } else {
        FILE *fi = fopen(P.initwts_file.c_str(), "rb");
        if (!fi) { fprintf(log, "can not open initial weights file: %s\n", P.initwts_file.c_str()); exit(0); }
        fprintf(log, "Loading Init weight file...\n");
        const std::string err = bp::read_weights(fi, L, P.layersizes, weights, bias);
        fclose(fi);
        if (!err.empty()) { fprintf(log, "%s\n", err.c_str()); exit(0); }
        fprintf(log, "Init weight file loaded.\n");
    }
    fflush(log);
    std::vector<float> indata, targ;           // stacked host copies, only for stack=host
    if (!P.stack_on_device) {
        indata.resize((size_t)P.layersizes[0] * P.traincache); targ.resize((size_t)P.layersizes[L - 1] * P.traincache);
    }

    // Interface::get_pfile_info's progress lines (Interface.cc:479,512,528,536,554); the checks themselves ran in reader.open()
    fprintf(log, "begin to read in_pfile\nbegin to read target_pfile\n");
    fprintf(log, "tmpsentnum=%d,tmpframenum=%d,total_frames=%d\n", (int)reader.total_sents(), (int)reader.total_frames(), (int)reader.total_frames());
    fprintf(log, "frames or sentence num in target pfile and data pfile is consistent.\n");
    fprintf(log, "Get pfile info over: Training data has %u frames, %u sentences.\n", reader.total_frames(), reader.total_sents());
    int st, en;
    parse_range(P.train_range, &st, &en, log);
    const bp::PfileReader::Plan tp = reader.plan(st, en);
    const int nchunks = (int)tp.chunk_frame_st.size();
    fprintf(log, "Get chunk info over: Training sentences have %d chunks, %d samples.\n", nchunks, (int)tp.total_samples);
    std::vector<int> chunk_index(nchunks);
    for (int i = 0; i < nchunks; ++i) chunk_index[i] = i;
    bp::PfileReader::rand_index(chunk_index.data(), nchunks);           // BPtrain.cc:47
    if (P.gpu_used < 1) { printf("GPU Num %d Not In Range %d-\n", P.gpu_used, 1); exit(0); }      // BP_GPU.cu:20-24

    // ---- fork the data-parallel ranks (everything above is inherited; nothing has touched the GPU yet)
    bp::ChunkRing *ring = nullptr;
    if (world > 1) {
        int fcap = 1;
        for (int c = 0; c < nchunks; ++c) { const int f = reader.chunk_shape(tp, c).n_frames; if (f > fcap) fcap = f; }
        ring = new bp::ChunkRing(world, fcap, P.traincache, en - st + 2, P.fea_dim, P.layersizes[L - 1], reader.nat());
        g_ring = ring; atexit(ring_abort_at_exit);
        fflush(stdout); fflush(log);
        for (int r = 1; r < world; ++r) {
            const pid_t c = fork();
            if (c < 0) { printf("fork failed\n"); exit(0); }
            if (c == 0) { rank = r; kids.clear(); lead = false; log = fopen("/dev/null", "wt"); fp_out = nullptr; break; }
            kids.push_back(c);
        }
    }

    // ---- BPtrain.cc:31-96
    bp_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.gpu_used = P.gpu_used > 0 ? P.gpu_used : 1; cfg.numlayers = L;
    for (int i = 0; i < L; ++i) cfg.layersizes[i] = P.layersizes[i];
    cfg.bunchsize = P.bunchsize / world;                    // BP_GPU.cu:29-36: the bunch is split over the GPUs
    cfg.lrate = P.lrate; cfg.momentum = P.momentum; cfg.weightcost = P.weightcost;
    cfg.dropoutflag = P.dropoutflag; cfg.visible_omit = P.visible_omit; cfg.hid_omit = P.hid_omit;
    cfg.activation = P.activation; cfg.momentum_rule = P.momentum_rule; cfg.seed = P.dropout_seed; cfg.compute_dtype = P.compute_dtype;
    cfg.max_chunk_frames = P.traincache;
    if (P.device >= 0) cfg.device = P.device + (world > 1 ? rank : 0);
    else if (world > 1) {
        int ndev = 1;
        if (bp_device_count(&ndev) != 0 || ndev < 1) { printf("%s\n", bp_last_error()); if (ring) ring->abort(); exit(0); }
        cfg.device = rank % ndev;
    }
    if (lead) printf("Use GPU Device : %d\n", P.gpu_used);
    BP_GPU *TrainObj = new BP_GPU(cfg, weights, bias, world, rank, dp_key.c_str());
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    if (world > 1) {
        // one reader per node: every rank's helper thread takes its part of producing each chunk into the shared ring
        // (rank 0: tables + shuffle + noise-aware rows; everyone: 1/world of the frame conversion), the main thread
        // consumes: this rank's rows of every global minibatch
        if (!P.stack_on_device && lead) fprintf(log, "(gpu_used > 1: stack=host is not used, the context windows are built on the device)\n");
        ring->register_rank(rank);                          // (peers notice this process dying without its atexit hook)
        { const unsigned hw = std::thread::hardware_concurrency(); reader.set_convert_threads(hw > (unsigned)world ? (int)(hw / (unsigned)world) : 1); }   // every rank converts at once: cores / ranks workers each
        const bool ring_pinned = bp_host_register(ring->base(), ring->bytes()) == 0;      // (per process: after the fork)
        if (!ring_pinned) fprintf(stderr, "bptrain: rank %d could not pin the shared chunk ring (%s); its uploads will be staged copies\n", rank, bp_last_error());
        std::thread helper([&] {
            for (int i = 0; i < nchunks; ++i)
                if (!ring->produce(reader, tp, i, chunk_index[i], true, rank)) return;
        });
        for (int i = 0; i < nchunks; ++i) {
            bp::ChunkRing::View v;
            if (!ring->acquire(i, v)) {
                const std::string why = ring->error();
                printf("bptrain: the data-parallel group was aborted%s%s\n", why.empty() ? "" : ": ", why.c_str());
                fflush(stdout); _exit(3);
            }
            fprintf(log, "Starting chunk %d of %d containing %d samples.\n", i + 1, nchunks, v.n_samples);
            fflush(log);
            const std::vector<int> rows = shard_rows(v.n_samples, P.bunchsize, world, rank);
            if (lead && v.n_samples % P.bunchsize)
                printf("this bunch has only %d samples and is ignored.\n", v.n_samples % P.bunchsize);   // BP_GPU.cu:317
            std::vector<int> ws(rows.size()), tf(rows.size()), nr(v.nat_row ? rows.size() : 0);
            for (size_t k = 0; k < rows.size(); ++k) {
                ws[k] = v.win_start[rows[k]]; tf[k] = v.targ_frame[rows[k]];
                if (v.nat_row) nr[k] = v.nat_row[rows[k]];
            }
            bp_window_chunk c;
            memset(&c, 0, sizeof(c));
            c.n_samples = (int)rows.size(); c.n_frames = v.n_frames; c.fea_dim = P.fea_dim; c.context = P.fea_context; c.n_nat = v.n_nat;
            c.fea = v.fea; c.targ_frames = v.targ; c.nat = v.nat;
            c.win_start = ws.data(); c.targ_frame = tf.data(); c.nat_row = v.nat_row ? nr.data() : nullptr;
            TrainObj->train_windows(c);                     // returns once frames and tables are on the device: the slot is free
            ring->done(i);
        }
        helper.join();
        g_ring_finished = true;
        if (ring_pinned) bp_host_unregister(ring->base());
    } else {
        ChunkStream chunks(reader, tp, chunk_index, true, P.prefetch, P.stack_on_device);
        const bool chunk_times = getenv("BPTRAIN_CHUNK_TIMES") != nullptr;
        for (int i = 0; i < nchunks; ++i) {
            const WindowChunk &w = chunks.get(i);           // (the read of chunk i+1 is now running behind us)
            if (chunk_times) {                              // development aid: when each chunk was handed over (the host blocks on the
                struct timespec tc; clock_gettime(CLOCK_MONOTONIC, &tc);   // device two chunks back, so the intervals are device time per chunk)
                fprintf(stderr, "chunk %d at %.4f s\n", i + 1, (tc.tv_sec - ts0.tv_sec) + 1e-9 * (tc.tv_nsec - ts0.tv_nsec));
            }
            fprintf(log, "Starting chunk %d of %d containing %d samples.\n", i + 1, nchunks, w.n_samples);
            fflush(log);
            if (P.stack_on_device) {
                TrainObj->train_windows(describe(w, P.fea_context));
            } else {
                reader.expand(w, indata.data(), targ.data());
                TrainObj->train(w.n_samples, indata.data(), targ.data());
            }                                               // both return once the chunk is on the device; the
        }                                                   // GPU works on it while the next one is prepared
    }
    if (lead) printf("begin to write weights\n");
    TrainObj->returnWeights(weights, bias);                 // (waits for the last chunk's bunches)
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (world > 1) {
        TrainObj->dp_detach();                              // collective; the weights are replicated on every rank
        if (!lead) { delete TrainObj; fflush(stdout); _exit(1); }
    }
    {
        const double dt = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
        fprintf(log, "Training pass: %u samples in %.3f s (%.0f frames/s, reader + upload + GPU).\n", tp.total_samples, dt,
                dt > 0 ? tp.total_samples / dt : 0.0);
    }
    fprintf(log, "Saving weights to file...\n");
    bp::write_weights(fp_out, L, P.layersizes, weights, bias);
    fclose(fp_out);
    fprintf(log, "Saving over.\n");
    printf("finish to write weights\n\n");

    printf("begin to CV\n");
    fprintf(log, "Starting CV.\n");
    parse_range(P.cv_range, &st, &en, log);
    const bp::PfileReader::Plan cp = reader.plan(st, en);
    fprintf(log, "Get cv chunk info over: CV sentences have %d chunks, %d samples.\n", (int)cp.chunk_frame_st.size(), (int)cp.total_samples);
    float squared_err = 0.0f;
    {
        std::vector<int> cv_order(cp.chunk_frame_st.size());
        for (size_t i = 0; i < cv_order.size(); ++i) cv_order[i] = (int)i;
        ChunkStream chunks(reader, cp, cv_order, false, P.prefetch, P.stack_on_device);
        for (int i = 0; i < (int)cv_order.size(); ++i) {
            const WindowChunk &w = chunks.get(i);
            printf("cur_chunk_samples=%d\n", w.n_samples);
            if (P.stack_on_device) {
                squared_err += TrainObj->CrossValid_windows(describe(w, P.fea_context));
            } else {
                reader.expand(w, indata.data(), targ.data());
                squared_err += TrainObj->CrossValid(w.n_samples, indata.data(), targ.data());
            }
        }
    }
    const float cvacc = squared_err / cp.total_samples;                  // BPtrain.cc:84
    fprintf(log, "CV over. squared error: %f\n", cvacc);
    fflush(log);
    fprintf(log, "Total cost time: %.1f s.\n", (double)time(NULL) - t_start);
    for (pid_t c : kids) { int st = 0; waitpid(c, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 1) fprintf(log, "rank process %d ended abnormally\n", (int)c); }
    printf("all finish!\n");
    delete TrainObj;
    fclose(log);
    return 1;                                                            // BPtrain.cc:100
}
