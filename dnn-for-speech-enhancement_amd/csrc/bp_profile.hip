// bp_profile.hip -- C-ABI implementation (include/bp_c_api.h), part 3 of 3: measurement entry points used by bench.py --
// the in-step event profile (bp_profile_step), the device's measured peaks (bp_measure_peaks) and back-to-back timing of
// single kernels of the step (bp_time_kernel).  No reference counterpart (the reference times whole passes on the host,
// BPtrain.cc:25-26,91-92).
#include <hip/hip_runtime.h>
#include <string>

#include "bp_handle.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    h->prof = &prof;
    hipError_t er = prof_mark(h, -1);                        // origin
    for (int i = 0; er == hipSuccess && i < n_bunches; ++i) {
        h->next_first = (step_stages(h) && i + 1 < n_bunches) ? first_frame + (i + 1) * h->B : -1;   // (as bp_train_resident)
        er = bunch(h, first_frame + i * h->B, true);
        h->step++;
    }
    h->next_first = -1; h->pre.valid = false;
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
