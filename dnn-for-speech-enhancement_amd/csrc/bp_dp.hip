// bp_dp.hip -- C-ABI implementation (include/bp_c_api.h), part 2 of 3: the in-library data-parallel exchange.  Host
// rendezvous (bp_rdv.h), hipIpc peer mappings, the attach-time self-test of the memory-model contract, the two transports
// (native peer kernels of bp_dp.h | RCCL reduce-scatter / all-gather resolved with dlopen) and the driver of one sharded
// minibatch, which cuts the step of bp_step.hip at the gradient exchange (semantics donor: the reference's commented-out
// train_bunch_multi, BP_GPU.cu:775-908).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "bp_handle.h"
#include "bp_dp.h"
#include "bp_rdv.h"

struct ncclUniqueIdBytes { char internal[128]; };   // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value

// ------------------------------------------------------------------ host rendezvous (bp_rdv.h), C ABI
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
// ------------------------------------------------------------------ in-library data-parallel exchange (bp_dp.h)
// What every rank publishes through the rendezvous block (RdvShm::blob).
struct DpBlob {
    int device; char pci[20];
    hipIpcMemHandle_t params, grad, deltas, flags, probe_p, probe_g, recv;
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
    int backend;                  // 0: native peer kernels over hipIpc mappings (reduce-scatter by peer reads) | 1: RCCL reduce-scatter / all-gather | 2: native, push form
    bool push;                    // backend 2 / 3: gradient slices are WRITTEN into the owners' receive buffers (bp_dp_push)
    bool gbf16;                   // backend 3: ... rounded to bf16 on the way (half the bytes on the fabric; fp32 sum at the owner)
    float *recv, *p_recv[BP_DP_MAXRANKS];   // push form: own receive buffer (fine-grained, exported) and every rank's mapping of its own
    size_t roff[BP_MAXLAYER], per4[BP_MAXLAYER];   // push form: the layer's region in a receive buffer (float index) and float4 per slice slot
    unsigned *arrive_push;        // [BP_MAXLAYER] last-arriver counters of bp_dp_push
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
                            (void *)d->p_probe_p[p], (void *)d->p_probe_g[p], (void *)d->p_recv[p]})
                if (q) (void)hipIpcCloseMemHandle(q);
        }
    for (auto &e : d->ev_g) if (e) (void)hipEventDestroy(e);
    for (auto &e : d->ev_w) if (e) (void)hipEventDestroy(e);
    if (d->ev_comm) (void)hipEventDestroy(d->ev_comm);
    if (d->comm) (void)hipStreamDestroy(d->comm);
    if (d->grad_fine) { if (h->grad == d->grad_fine) h->grad = d->grad_prev; (void)hipFree(d->grad_fine); }
    for (void *q : {(void *)d->flags, (void *)d->arrive, (void *)d->done, (void *)d->probe_p, (void *)d->probe_g, (void *)d->red, (void *)d->recv, (void *)d->arrive_push})
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
    if (hipMalloc((void **)&sink, 64) != hipSuccess) { (void)hipHostFree(cnt); return fail(BP_ERR_NOMEM, "hipMalloc (self-test sink)"); }
    // inside the loop a HIP error leaves through `break` so that the tail below frees cnt / sink
#define STK(x) { const hipError_t _e = (x); if (_e != hipSuccess) { rc = fail(BP_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(_e)); break; } }
    const DpPeers peers = dp_peers(d);
    DpReduceArgs a; memset(&a, 0, sizeof(a));
    for (int p = 0; p < d->world; ++p) { a.params[p] = d->p_probe_p[p]; a.grads[p] = d->p_probe_g[p]; }
    a.world = d->world; a.rank = d->rank; a.peers = peers;
    int rc = BP_OK;
    bool c_dead = false;                                       // round (C) timed out here: no further (C) launches, verdict "keep the events"
    for (int r = 1; r <= rounds && rc == BP_OK; ++r) {
        const unsigned ep = ep_base + (unsigned)r;             // flag values of the probe words only ever grow (fresh flag array per attach)
        // ---- (W): warm this device's caches with the OLD contents, let the peers overwrite, wait, re-read plainly
        hipLaunchKernelGGL(bp_dp_probe_touch, dim3(64), dim3(256), 0, h->stream, d->probe_p, sink);
        STK(hipStreamSynchronize(h->stream));
        if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }
        a.flag_index = bp_dp_flag_index(BP_DP_FLAG_PROBE, 0, d->rank); a.epoch = ep;
        hipLaunchKernelGGL(bp_dp_probe_push, dim3(1), dim3(256), 0, d->comm, a, (unsigned)r);
        hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, h->stream, d->flags, bp_dp_flag_index(BP_DP_FLAG_PROBE, 0, 0), d->world, ep,
                           d->budget_ticks, d->err, 3u);
        if (d->acquire_mode) hipLaunchKernelGGL(bp_dp_l2_invalidate, dim3(64), dim3(64), 0, h->stream);
        hipLaunchKernelGGL(bp_dp_probe_check, dim3(64), dim3(256), 0, h->stream, d->probe_p, d->world, (unsigned)r, cnt);
        STK(hipGetLastError());
        STK(hipStreamSynchronize(h->stream));
        STK(hipStreamSynchronize(d->comm));
        // ---- (G): fill the fine-grained probe with plain stores, signal behind the kernel boundary, peers read it
        hipLaunchKernelGGL(bp_dp_probe_fill, dim3(64), dim3(256), 0, h->stream, d->probe_g, (unsigned)r, (unsigned)d->rank);
        STK(hipEventRecord(d->ev_comm, h->stream));
        STK(hipStreamWaitEvent(d->comm, d->ev_comm, 0));
        hipLaunchKernelGGL(bp_dp_signal, dim3(1), dim3(64), 0, d->comm, peers, d->world, bp_dp_flag_index(BP_DP_FLAG_PROBE, 1, d->rank), ep);
        hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, d->comm, d->flags, bp_dp_flag_index(BP_DP_FLAG_PROBE, 1, 0), d->world, ep,
                           d->budget_ticks, d->err, 3u);
        hipLaunchKernelGGL(bp_dp_probe_check_remote, dim3(8), dim3(256), 0, d->comm, a, (unsigned)r, cnt + 1);
        STK(hipGetLastError());
        STK(hipStreamSynchronize(d->comm));
        if (*(volatile unsigned *)d->err) { rc = fail(BP_ERR_STATE, "data-parallel self-test: a peer's flag never arrived"); break; }
        if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }    // nobody refills a probe a peer still reads
        // ---- (C): the same direction with the step's IN-KERNEL hand-off -- the filling kernel counts its own workgroups, the
        // exchange stream's bp_dp_sync (already queued, running beside it) sees the count, tells the peers, the peers read.
        // No event and no kernel boundary between the stores and the readers' flag.
        // The spinning bp_dp_sync is queued BEFORE the kernel it waits for, so this form needs the two streams to run
        // concurrently.  Where they do not (a serialising profiler or debug setting, queue aliasing), the sync times out: that
        // is not an error of the group but a verdict on the hand-off -- short budget, the timeout is counted like a stale word,
        // every rank then keeps the event + kernel-boundary hand-off.  The barrier stays unconditional so that the
        // ranks remain in step whatever each of them saw.
        if (d->counters_ok) {
            if (!c_dead) {
                const unsigned ep2 = ep + 0x4000u;                     // (flag words only grow; probe word 2 is this direction's)
                const unsigned long long short_budget = d->budget_ticks < 200000000ull ? d->budget_ticks : 200000000ull;   // <= 2 s
                hipLaunchKernelGGL(bp_dp_sync, dim3(1), dim3(64), 0, d->comm, d->done + BP_MAXLAYER, (ep_base + (unsigned)r) * 64u, peers, d->flags, d->world,
                                   bp_dp_flag_index(BP_DP_FLAG_PROBE, 2, d->rank), bp_dp_flag_index(BP_DP_FLAG_PROBE, 2, 0), ep2, short_budget, d->err, 3u);
                hipLaunchKernelGGL(bp_dp_probe_check_remote, dim3(8), dim3(256), 0, d->comm, a, (unsigned)r + 100u, cnt + 2);
                hipLaunchKernelGGL(bp_dp_probe_fill_count, dim3(64), dim3(256), 0, h->stream, d->probe_g, (unsigned)r + 100u, (unsigned)d->rank, d->done + BP_MAXLAYER);
                STK(hipGetLastError());
                STK(hipStreamSynchronize(h->stream));
                STK(hipStreamSynchronize(d->comm));
                if (*(volatile unsigned *)d->err) { *(volatile unsigned *)d->err = 0u; c_dead = true; cnt[2] += 1u; }
            }
            if (rdv_barrier(d->rdv) != 0) { rc = fail(BP_ERR_STATE, g_rdv_err); break; }
        }
    }
#undef STK
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
    if (transport < BP_DP_TRANSPORT_NATIVE || transport > BP_DP_TRANSPORT_NATIVE_PUSH_BF16) return fail(BP_ERR_ARG, "bp_dp_attach: unknown transport");
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
    d->push = transport == BP_DP_TRANSPORT_NATIVE_PUSH || transport == BP_DP_TRANSPORT_NATIVE_PUSH_BF16;
    d->gbf16 = transport == BP_DP_TRANSPORT_NATIVE_PUSH_BF16;
    d->budget_ticks = (unsigned long long)(dp_timeout_s() * 1.0e8);      // wall_clock64: 100 MHz
#define DK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::string m = std::string("bp_dp_attach: ") + #x + ": " + hipGetErrorString(_e); \
        dp_release(h, true); return fail(BP_ERR_DEVICE, m); } } while (0)
#define DR(x) do { int _r = (x); if (_r != BP_OK) { std::string m = g_bp_err; dp_release(h, true); g_bp_err = m; return _r; } } while (0)
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
    d->counters_ok = transport != BP_DP_TRANSPORT_RCCL;
#ifdef BP_DEV
    if (dev_flag("BP_DP_NO_COUNTERS")) d->counters_ok = false;    // A/B: the event + kernel-boundary hand-off
#endif
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
    size_t max_slice = 4, recv_floats = 0;
    for (int l = 1; l < h->L; ++l) {                           // equal float4-aligned slices of [W_l|b_l]
        const size_t cnt4 = h->g_cnt[l] / 4, per4 = (cnt4 + world - 1) / world;
        d->per4[l] = per4; d->roff[l] = recv_floats; recv_floats += (size_t)world * 4 * per4;
        const size_t a = per4 * rank < cnt4 ? per4 * rank : cnt4, b = per4 * (rank + 1) < cnt4 ? per4 * (rank + 1) : cnt4;
        d->lo[l] = h->g_off[l] + 4 * a; d->hi[l] = h->g_off[l] + 4 * b;
        if (4 * per4 > max_slice) max_slice = 4 * per4;
        if (transport == BP_DP_TRANSPORT_RCCL && per4 * world != cnt4) { dp_release(h, true); return fail(BP_ERR_ARG, "bp_dp_attach: RCCL transport: layer segment not divisible by the world"); }
    }
    if (d->push) {
        // the receive buffer peers WRITE this rank's slice contributions into: fine-grained like the gradient buffer (never cached
        // dirty; read with system-scope loads), one region per layer of `world` slots
        if (hipExtMallocWithFlags((void **)&d->recv, (recv_floats + SLACK) * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            d->recv = nullptr;
            DK(hipMalloc((void **)&d->recv, (recv_floats + SLACK) * sizeof(float)));
        }
        DK(hipMemset(d->recv, 0, (recv_floats + SLACK) * sizeof(float)));
        DK(hipMalloc((void **)&d->arrive_push, BP_MAXLAYER * sizeof(unsigned)));
        DK(hipMemset(d->arrive_push, 0, BP_MAXLAYER * sizeof(unsigned)));
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
    if (d->push) DK(hipIpcGetMemHandle(&mine.recv, d->recv));
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
            d->p_probe_p[p] = d->probe_p; d->p_probe_g[p] = d->probe_g; d->p_recv[p] = d->recv;
            continue;
        }
        DK(hipIpcOpenMemHandle((void **)&d->p_params[p], pb.params, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_grad[p], pb.grad, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_deltas[p], pb.deltas, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_flags[p], pb.flags, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_probe_p[p], pb.probe_p, hipIpcMemLazyEnablePeerAccess));
        DK(hipIpcOpenMemHandle((void **)&d->p_probe_g[p], pb.probe_g, hipIpcMemLazyEnablePeerAccess));
        if (d->push) DK(hipIpcOpenMemHandle((void **)&d->p_recv[p], pb.recv, hipIpcMemLazyEnablePeerAccess));
    }
    if (rdv_barrier(d->rdv) != 0) { dp_release(h, true); return fail(BP_ERR_STATE, g_rdv_err); }   // every rank has mapped every peer
    if (transport == BP_DP_TRANSPORT_RCCL) {
        ncclUniqueIdBytes id; memcpy(&id, d->rdv->shm->shared, sizeof(id));
        const int e = d->rccl.CommInitRank(&d->rccl_comm, world, id, rank);
        if (e != 0) { std::string m = std::string("ncclCommInitRank: ") + d->rccl.GetErrorString(e); dp_release(h, true); return fail(BP_ERR_DEVICE, m); }
        DK(hipMalloc((void **)&d->red, (max_slice + SLACK) * sizeof(float)));
    } else {
        // ---- the memory-model contract of the native exchange, checked on these devices before anything relies on it.  A group of ONE
        // rank runs it too: rounds (W) and (G) pass trivially there, round (C) finds out whether the two streams really run side by
        // side -- where they do not (serialised dispatch under a counter-collecting profiler) the spinning bp_dp_sync queued in front
        // of the kernel it waits for would run out its whole budget on every layer of every step.
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

int dp_check(bp_handle *h)
{
    if (h->dp && *(volatile unsigned *)h->dp->err) {
        const unsigned e = *(volatile unsigned *)h->dp->err;
        if (e / 1000u == 4)
            return fail(BP_ERR_STATE, "data-parallel exchange timed out on the device: this rank's weight-gradient tiles were never all counted "
                                      "(exchange stream and main stream not running concurrently?)");
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
static int dp_update_grid(const DpReduceArgs &a, int layer)
{
    const size_t n4 = (a.hi - a.lo) / 4;
    // few, deep workgroups: at 2048 workgroups the exchange kernel crowds the GEMMs it runs beside out of the CUs'
    // memory pipes (C2 step through the exchange path on one GPU: 0.51 ms at 2048, 0.34 at 512, 0.28 at 128, 0.31 at 64).
    // Layer 1's update runs beside the second half of the weight-gradient launch (MFMA-bound, not a GEMM waiting on memory),
    // and every later update and the next forward queue behind it: it takes more workgroups (round 5, same box: 0.2506-0.2508
    // ms per step at 224 against 0.2540-0.2564 at 128; 160 / 192: 0.2517-0.2533; 256+: slower again)
    int max_grid = layer == 1 ? 224 : 128;
#ifdef BP_DEV
    max_grid = dev_int("BP_DP_GRID", max_grid);
    if (layer == 1) max_grid = dev_int("BP_DP_GRID1", max_grid);
#endif
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
    for (int p = 0; p < d->world; ++p) {
        // pull form: slice `rank` of every rank's gradient segment, read over the fabric | push form: this rank's `world` LOCAL
        // receive slots of the layer (the kernel indexes grads[p] + lo; same summation order 0..world-1 either way)
        // (bf16 segments: the kernel treats grads[p] as an array of 2-byte elements with the same indices)
        a.grads[p] = d->gbf16 ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(d->recv) + d->roff[l] + (size_t)p * 4 * d->per4[l] - a.lo)
                              : (d->push ? d->recv + d->roff[l] + (size_t)p * 4 * d->per4[l] - a.lo : d->p_grad[p]);
        a.params[p] = d->p_params[p];
    }
    const int grid = dp_update_grid(a, l);
    if (d->gbf16) { hipLaunchKernelGGL((bp_dp_reduce_update<0, true>), dim3(grid), dim3(256), 0, d->comm, a); return hipGetLastError(); }
    switch (d->world) {
    case 1: hipLaunchKernelGGL(bp_dp_reduce_update<1>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 2: hipLaunchKernelGGL(bp_dp_reduce_update<2>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 4: hipLaunchKernelGGL(bp_dp_reduce_update<4>, dim3(grid), dim3(256), 0, d->comm, a); break;
    case 8: hipLaunchKernelGGL(bp_dp_reduce_update<8>, dim3(grid), dim3(256), 0, d->comm, a); break;
    default: hipLaunchKernelGGL(bp_dp_reduce_update<0>, dim3(grid), dim3(256), 0, d->comm, a); break;
    }
    return hipGetLastError();
}

// comm stream (push form): this rank's gradient segment of layer l, slice by slice, into the owners' receive buffers; raises GRAD(l, rank) there
static hipError_t dp_push_layer(bp_handle *h, int l)
{
    bp_dp *d = h->dp;
    DpPushArgs a; memset(&a, 0, sizeof(a));
    a.grad = h->grad; a.seg = h->g_off[l]; a.n4 = h->g_cnt[l] / 4; a.per4 = d->per4[l]; a.roff = d->roff[l];
    for (int p = 0; p < d->world; ++p) a.recv[p] = d->p_recv[p];
    a.world = d->world; a.rank = d->rank; a.arrive = d->arrive_push + l;
    a.peers = dp_peers(d); a.flag_index = bp_dp_flag_index(BP_DP_FLAG_GRAD, l, d->rank); a.epoch = d->epoch;
    int grid = (int)((a.n4 + 256 * 4 - 1) / (256 * 4));
    const int max_grid = l == 1 ? 224 : 128;                   // (the update kernel's sizes: few, deep workgroups beside the GEMMs)
    if (grid > max_grid) grid = max_grid;
    if (d->gbf16) hipLaunchKernelGGL(bp_dp_push<true>, dim3(grid < 1 ? 1 : grid), dim3(256), 0, d->comm, a);
    else hipLaunchKernelGGL(bp_dp_push<false>, dim3(grid < 1 ? 1 : grid), dim3(256), 0, d->comm, a);
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
        if (d->push) { for (int i = 0; i < n; ++i) if ((er = dp_push_layer(h, ls[i])) != hipSuccess) return er; }   // (each push raises its own flags)
        else hipLaunchKernelGGL(bp_dp_signal_n, dim3(1), dim3(64), 0, d->comm, peers, d->world, sig, d->epoch);
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
            hipLaunchKernelGGL(bp_dp_reduce_update<1>, dim3(dp_update_grid(a, l)), dim3(256), 0, d->comm, a);
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
hipError_t dp_bunch(bp_handle *h, int first)
{
    bp_dp *d = h->dp;
    const int L = h->L;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    const float *x0, *tg;
    CKE(step_inputs(h, first, &x0, &tg));
    const unsigned prev_epoch = d->epoch;
    d->epoch++;
    for (int l = 1; l < L; ++l) {
        CKE(dp_wait_weights(h, l, prev_epoch));
        if (prev_epoch) CKE(step_shadow(h, l));                 // bf16 mode: bf16 copy of the gathered fp32 weights
        CKE(step_forward(h, l, x0, tg));
    }
    for (int l = L - 1; l >= 2; --l) CKE(step_dgrad(h, l));
    int all[BP_MAXLAYER], nall = 0;
    for (int l = 1; l < L; ++l) all[nall++] = l;
    // (the predicate comes from the step itself: only the launch that really counts its tiles may be waited on that way)
    if (d->counters_ok && d->backend != BP_DP_TRANSPORT_RCCL && step_wgrads_count(h)) {
        // ONE grouped weight-gradient launch, layer 1's tiles first; every tile counts itself into done[l] (bp_wgrad_dma.h), and
        // the exchange stream -- queued right here, running beside the launch -- picks each layer up as soon as its count is
        // complete: no event on this stream (each cost it a ~7 us bubble), no split of the launch, and layer 1's exchange
        // overlaps the other layers' tiles instead of waiting behind a launch boundary.
        unsigned *done[BP_MAXLAYER] = {nullptr};
        const DpPeers peers = dp_peers(d);
        for (int l = 1; l < L; ++l) {
            done[l] = d->done + l;
            d->done_target[l] += step_wgrad_tiles(h, l);
            if (d->push) {
                // wait for the local tiles only (world 0: nothing signalled, nobody waited for), push the slices (raises GRAD at the owners),
                // wait for every source's contribution to this rank's slice
                hipLaunchKernelGGL(bp_dp_sync, dim3(1), dim3(64), 0, d->comm, d->done + l, d->done_target[l], peers, d->flags, 0,
                                   bp_dp_flag_index(BP_DP_FLAG_GRAD, l, d->rank), bp_dp_flag_index(BP_DP_FLAG_GRAD, l, 0), d->epoch, d->budget_ticks, d->err, 1u);
                CKE(hipGetLastError());
                CKE(dp_push_layer(h, l));
                hipLaunchKernelGGL(bp_dp_wait, dim3(1), dim3(64), 0, d->comm, d->flags, bp_dp_flag_index(BP_DP_FLAG_GRAD, l, 0), d->world, d->epoch, d->budget_ticks, d->err, 1u);
            } else
            hipLaunchKernelGGL(bp_dp_sync, dim3(1), dim3(64), 0, d->comm, d->done + l, d->done_target[l], peers, d->flags, d->world,
                               bp_dp_flag_index(BP_DP_FLAG_GRAD, l, d->rank), bp_dp_flag_index(BP_DP_FLAG_GRAD, l, 0), d->epoch, d->budget_ticks, d->err, 1u);
            CKE(hipGetLastError());
            CKE(dp_reduce_layer(h, l));
        }
        CKE(step_wgrads_store(h, all, nall, x0, done));
    } else {
        // layer 1 (the largest segment, needed first by the next forward) goes out alone; the rest as ONE grouped launch:
        // its exchange queues behind layer 1's on the comm stream anyway, and one launch + one event replace L-2 of each
        CKE(step_wgrads_store(h, all, 1, x0, nullptr));
        CKE(dp_exchange_layers(h, all, 1));
        if (nall > 1) {
            CKE(step_wgrads_store(h, all + 1, nall - 1, x0, nullptr));
            CKE(dp_exchange_layers(h, all + 1, nall - 1));
        }
    }
#undef CKE
    return hipSuccess;
}

// After the last minibatch of a call: the main stream waits until every layer's weights have been gathered (and
// therefore every peer has finished reading this rank's gradients), so that stream order again covers everything.
hipError_t dp_flush(bp_handle *h)
{
    bp_dp *d = h->dp;
    hipError_t er;
    for (int l = 1; l < h->L; ++l) {
        if ((er = dp_wait_weights(h, l, d->epoch)) != hipSuccess) return er;
        if (d->epoch && (er = step_shadow(h, l)) != hipSuccess) return er;
    }
    if ((er = hipEventRecord(d->ev_comm, d->comm)) != hipSuccess) return er;
    return hipStreamWaitEvent(h->stream, d->ev_comm, 0);
}

// bp_get_deltas on an attached handle: the momentum state is sharded; pull the peers' slices into the local arena.
bool dp_gathers_deltas(const bp_handle *h) { return h->dp && h->dp->world > 1; }
int dp_gather_deltas(bp_handle *h)
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
extern "C" int bp_dp_handoff(bp_handle *h, int *in_kernel)
{
    if (!h || !h->dp || !in_kernel) return fail(BP_ERR_STATE, "bp_dp_handoff: handle is not attached / null argument");
    *in_kernel = h->dp->counters_ok && h->dp->backend != BP_DP_TRANSPORT_RCCL && step_wgrads_count(h) ? 1 : 0;
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
