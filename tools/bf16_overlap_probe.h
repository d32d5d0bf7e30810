// tools/bf16_overlap_probe.h -- NOT part of the library.  Round-5 experiment: the bf16 step's update
// launches on a SECOND stream beside the dgrad GEMMs, ordered by device-side counters and one-wave gate kernels instead of
// events.  This is the code as it was compiled into the development library (bp_step.hip, behind BP_BF16_OVERLAP=1; the handle
// carried `struct { hipStream_t stream; unsigned *cnt; unsigned *err; unsigned steps; bool on; } ov;`, bunch() called
// bf_bunch_overlapped() for fused bf16 steps and bp_train_resident() called ov_join() before recording its end event).
// bf16 parity suite green with it.  MEASURED SLOWER (profiles/r05_bf16_overlap.txt): configs[4] 0.647-0.670 ms per step against
// 0.620-0.623 on the same box.  The timeline says why: the update launch of layer l starts ~4 us ahead of dgrad(l-1), fills every
// CU with its three 48 KB workgroups and keeps refilling freed slots; a dgrad workgroup needs 96 KB of LDS (4-stage operand
// ring), so it is not placed until the update launch has no more workgroups to dispatch -- dgrad takes 98-103 us beside a
// 75 us update launch instead of 30 us: the two run back to back, plus five signal kernels on the main stream.  Equal stream
// priorities change nothing.  Co-residency on a CU is bounded by LDS capacity (160 KB): one GEMM workgroup (96 KB) + ONE update
// workgroup, and a 48 KB (2-stage) GEMM ring costs the GEMM 53 us instead of 30 (profiles/r04_bf16_gemm_probe.txt).
#pragma once
#ifdef BP_DEV
// ------------------------------------------------------------------ experiment: update launches beside the dgrad GEMMs
// BP_BF16_OVERLAP=1 (development build only).  The bf16 step is two halves that never overlap: the HBM-bound update launch
// (MFMA ~11 % busy) and eleven GEMM launches that read their weights from the Infinity Cache and leave HBM idle (VERDICT r4
// item 1).  Here the update of layer l runs on a SECOND stream as soon as dgrad(l) -- the last reader of Wb_l, and the
// producer of nothing it needs later than dEdX_l from dgrad(l+1) -- has finished, beside dgrad(l-1) .. dgrad(2); the next
// step's forward of layer l waits for update(l).  Ordering is by device-side counters and one-wave gate kernels (an event
// fork/join costs ~20 us on this stack, profiles/r03_overlap_two_streams.txt), spins bounded.
__global__ void bp_gate_signal(unsigned *c, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store(c, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void bp_gate_wait(const unsigned *c, unsigned v, unsigned long long budget_ticks, unsigned *err)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if ((int)(__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) >= 0) break;
        if (wall_clock64() - t0 > budget_ticks) { atomicExch(err, 1u); break; }
        __builtin_amdgcn_s_sleep(16);
    }
}
static int ov_init(bp_handle *h)
{
    if (h->ov.stream) return BP_OK;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&h->ov.stream, hipStreamNonBlocking, dev_flag("BP_BF16_OVERLAP_SAMEPRIO") ? hi : lo));
    HIPCHK(hipMalloc((void **)&h->ov.cnt, 2 * BP_MAXLAYER * sizeof(unsigned)));
    HIPCHK(hipMemset(h->ov.cnt, 0, 2 * BP_MAXLAYER * sizeof(unsigned)));
    HIPCHK(hipHostMalloc((void **)&h->ov.err, sizeof(unsigned), hipHostMallocMapped));
    *h->ov.err = 0u;
    h->ov.steps = 0;
    return BP_OK;
}
static hipError_t ov_wait(bp_handle *h, hipStream_t st, int idx, unsigned v)
{
    if (v == 0) return hipSuccess;
    hipLaunchKernelGGL(bp_gate_wait, dim3(1), dim3(64), 0, st, h->ov.cnt + idx, v, 200000000ull /* 2 s */, h->ov.err);
    return hipGetLastError();
}
static hipError_t ov_signal(bp_handle *h, hipStream_t st, int idx, unsigned v)
{
    hipLaunchKernelGGL(bp_gate_signal, dim3(1), dim3(64), 0, st, h->ov.cnt + idx, v);
    return hipGetLastError();
}
// counters: [l] = "dgrad l of step n done" (n = value), [BP_MAXLAYER + l] = "update l of step n done"
static hipError_t bf_bunch_overlapped(bp_handle *h, const float *x0, const float *tg)
{
    const int L = h->L;
    const unsigned n = h->ov.steps, n1 = n + 1;
    hipError_t er;
#define CKE(x) do { er = (x); if (er != hipSuccess) return er; } while (0)
    hipStream_t main_st = h->stream, upd = h->ov.stream;
    for (int l = 1; l < L; ++l) {
        CKE(ov_wait(h, main_st, BP_MAXLAYER + l, n));          // update(l) of the previous step (l == 1: also the last reader of the input bunch's bf16 copy)
        if (l == 1) CKE(bf_input(h, x0, h->B));
        CKE(bf_fwd(h, l, h->B, tg, nullptr, true, 1.0f));
    }
    for (int l = L - 1; l >= 2; --l) { CKE(bf_dgrad(h, l)); CKE(ov_signal(h, main_st, l, n1)); }
    for (int l = L - 1; l >= 1; --l) {
        CKE(ov_wait(h, upd, l >= 2 ? l : 2, n1));
        h->stream = upd;
        er = bf_wgrads_dma(h, &l, 1, true);
        h->stream = main_st;
        if (er != hipSuccess) return er;
        CKE(ov_signal(h, upd, BP_MAXLAYER + l, n1));
    }
    h->ov.steps = n1;
#undef CKE
    return hipSuccess;
}
// the main stream catches up with the update stream (end of a bp_train_resident call: stream order covers everything again)
static hipError_t ov_join(bp_handle *h)
{
    if (!h->ov.stream || !h->ov.steps) return hipSuccess;
    return ov_wait(h, h->stream, BP_MAXLAYER + 1, h->ov.steps);
}
#endif

