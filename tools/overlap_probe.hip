// overlap_probe.hip -- what does it cost to order / un-order two kernels on MI355X with the HIP runtime?
//   hipcc --offload-arch=gfx950 -O2 tools/overlap_probe.hip -o tools/overlap_probe.bin && tools/overlap_probe.bin
// Kernel = every workgroup waits `us` microseconds on the wall clock (s_memrealtime, 100 MHz): a fixed-duration kernel
// that occupies 1 wave per CU, so two of them CAN run side by side.  Reported per pattern: average time of one
// repetition (events around 200 repetitions on the first stream).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int us, int *sink)
{
    const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    const unsigned long long w0 = wall_clock64();                  // 100 MHz
    while (wall_clock64() - w0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
    if (us < 0) *sink = (int)(__builtin_readcyclecounter() - t0);
}
int main()
{
    const int REP = 200, US = 40;
    int *sink; CK(hipMalloc(&sink, 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const unsigned flagsets[4] = {hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence,
                                  hipEventDisableTiming | hipEventReleaseToDevice};
    const char *fname[4] = {"default", "DisableTiming", "DisableTiming|DisableSystemFence", "DisableTiming|ReleaseToDevice"};
    auto timed = [&](const char *what, auto body) -> int {
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(t0, s1));
            for (int i = 0; i < REP; ++i) body();
            CK(hipEventRecord(t1, s1));
            CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%-72s %8.2f us per repetition\n", what, ms * 1e3 / REP);
        return 0;
    };
    dim3 g(256), b(64);
    timed("1 kernel (40 us)", [&] { hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); });
    timed("2 kernels, same stream, ordered", [&] { hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); });
    timed("2 kernels, same stream, second hipExtAnyOrderLaunch", [&] {
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        hipExtLaunchKernelGGL(spin, g, b, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, US, sink); });
    timed("2 kernels, same stream, second hipExtLaunchKernelGGL flags=0", [&] {
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        hipExtLaunchKernelGGL(spin, g, b, 0, s1, nullptr, nullptr, 0, US, sink); });
    timed("3 kernels: A, B any-order, C ordered (expect 80 if honoured, 120 if not)", [&] {
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        hipExtLaunchKernelGGL(spin, g, b, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, US, sink);
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); });
    for (int f = 0; f < 4; ++f) {
        hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, flagsets[f])); CK(hipEventCreateWithFlags(&e2, flagsets[f]));
        char nm[160];
        snprintf(nm, sizeof nm, "A(s1); record; B(s1)            [%s]", fname[f]);
        timed(nm, [&] { hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); hipEventRecord(e1, s1); hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); });
        snprintf(nm, sizeof nm, "A(s1); fork B(s2) || C(s1); join; (expect 80) [%s]", fname[f]);
        timed(nm, [&] {
            hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
            hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
            hipLaunchKernelGGL(spin, g, b, 0, s2, US, sink);
            hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
            hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); });
        CK(hipEventDestroy(e1)); CK(hipEventDestroy(e2));
    }
    // a graph with the same fork/join
    {
        hipGraph_t gr; hipGraphExec_t ge; hipEvent_t e1, e2;
        CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));
        hipLaunchKernelGGL(spin, g, b, 0, s2, US, sink);
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0));
        hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        CK(hipStreamEndCapture(s1, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        timed("graph: A; B || C; D  (expect 120)", [&] { hipGraphLaunch(ge, s1); });
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink);
        CK(hipStreamEndCapture(s1, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        timed("graph: A; B; C; D linear (expect 160 + gaps)", [&] { hipGraphLaunch(ge, s1); });
        timed("stream: A; B; C; D linear", [&] { for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(spin, g, b, 0, s1, US, sink); });
    }
    return 0;
}
