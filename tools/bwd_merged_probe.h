// bwd_merged_probe.h (development probe, NOT part of the library; measured and not adopted, profiles/r04_merged_dgrad_wgrad.txt) -- one launch = the dgrad of layer l AND the wgrad+update of layer l+1 (independent once dEdX_l+1
// exists: the dgrad reads W_l, the update writes W_l+1; BP_GPU.cu:611-652 orders them the same way).
//
// Why: a hidden dgrad at 256 frames is 256 workgroups = ONE per CU with one wave per SIMD, so every barrier and every
// LDS round trip of that wave is an idle matrix pipe (k-loop at 80 %, profiles/r03_gemm_trace_fwd.txt), and the grouped
// wgrad launch at the end of the step has a 12 us tail at falling occupancy (profiles/r03_wgrad_trace.txt).  Here the
// dgrad workgroups come FIRST in block order (dispatched first: they are the critical path of the step), the wgrad tiles
// of the layer above fill the remaining residency of every CU and run in the dgrad's pipe bubbles.
#pragma once
#include "../dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h"

template <class DG, class WG, int MINWG>
__global__ __launch_bounds__(256, MINWG) void bp_dgrad_wgrad(const GemmArgs dg, const EpiArgs de, const int n_dg, const MultiArgs w)
{
    constexpr int SM = DG::SMEM > WG::SMEM ? DG::SMEM : WG::SMEM;
    __shared__ __attribute__((aligned(16))) float smem[SM];
    const int b = blockIdx.x;
    if (b < n_dg) { DG::run(dg, de, b, n_dg, 0, smem); return; }
    const int bw = b - n_dg;
    int p = 0;
    while (p + 1 < w.n && bw >= w.first_tile[p + 1]) ++p;
    WG::run(w.g[p], w.e[p], bw - w.first_tile[p], w.first_tile[p + 1] - w.first_tile[p], smem);
}

/* How it was wired into bp_engine.hip for the measurement (git diff of the experiment, env switch BP_MERGE):

diff --git a/dnn-for-speech-enhancement_amd/csrc/bp_engine.hip b/dnn-for-speech-enhancement_amd/csrc/bp_engine.hip
index 91444fe..ea76056 100644
--- a/dnn-for-speech-enhancement_amd/csrc/bp_engine.hip
+++ b/dnn-for-speech-enhancement_amd/csrc/bp_engine.hip
@@ -23,6 +23,7 @@
 #include "bp_dp.h"
 #include "bp_rdv.h"
 #include "bp_wgrad_dma.h"
+#include "bp_bwd_merged.h"
 #include "bp_wgrad_dma_bf16.h"
 
 #include <dlfcn.h>
@@ -520,6 +521,24 @@ static hipError_t run_wgrads(hipStream_t st, Prepared *ps, int n, bool grouped)
     return hipSuccess;
 }
 
+// dgrad of layer l + the fused wgrad problems ws[0..n) in ONE launch (bp_bwd_merged.h); 256-frame bunches, wide layers
+template <class DG, int MINWG>
+static hipError_t run_dgrad_wgrad(hipStream_t st, Prepared &d, Prepared *ws, int n)
+{
+    d.g.tiles_m = (d.M + 31) / 32; d.g.tiles_n = (d.N + 63) / 64;
+    const int n_dg = (d.g.tiles_m * d.g.tiles_n + 7) & ~7;
+    MultiArgs a; memset(&a, 0, sizeof(a));
+    int t = 0;
+    for (int i = 0; i < n; ++i) {
+        ws[i].g.tiles_m = (ws[i].M + 63) / 64; ws[i].g.tiles_n = (ws[i].N + 63) / 64;
+        a.g[i] = ws[i].g; a.e[i] = ws[i].e; a.first_tile[i] = t;
+        t += (ws[i].g.tiles_m * ws[i].g.tiles_n + 7) & ~7;
+    }
+    a.first_tile[n] = t; a.n = n;
+    hipLaunchKernelGGL((bp_dgrad_wgrad<DG, WgradDma<16, 4, MINWG, 256>, MINWG>), dim3(n_dg + t), dim3(256), 0, st, d.g, d.e, n_dg, a);
+    return hipGetLastError();
+}
+
 static hipError_t launch_dgrad(bp_handle *h, hipStream_t st, int l, int M)
 {
     Prepared p = prep_dgrad(h, l, M);
@@ -752,6 +771,27 @@ static hipError_t bunch(bp_handle *h, int first, bool fused)
     // layer and the lower layers' updates come later), so the wgrad+update problems can all wait until
     // the last dgrad and share grouped launches (bp_gemm_multi), layer 1 (the largest) first.
     Prepared ws[BP_MAXLAYER]; int nw = 0;
+    static const int merge = getenv("BP_MERGE") ? atoi(getenv("BP_MERGE")) : 0;      // development A/B switch
+    if (merge && fused && h->grouped && B == 256) {
+        Prepared pend; bool have = false;                   // wgrad of the layer above, waiting for the next wide dgrad
+        for (int l = L - 1; l >= 1; --l) {
+            if (l != 1) {
+                Prepared d = prep_dgrad(h, l, B);
+                if (have && d.cfg != CFG_DGRAD_NARROW) {
+                    if (merge == 1) CKE((run_dgrad_wgrad<KDgradWide, 3>(h->stream, d, &pend, 1)));
+                    else CKE((run_dgrad_wgrad<KDgradWide, 2>(h->stream, d, &pend, 1)));
+                    have = false;
+                } else CKE(launch_dgrad(h, h->stream, l, B));
+                CKE(prof_mark(h, l == L - 1 ? BP_PROF_DGRAD_OUT : BP_PROF_DGRAD_HIDDEN));
+            }
+            if (have) { ws[nw++] = pend; have = false; }
+            pend = prep_wgrad(h, l, B, l == 1 ? x0 : h->y[l - 1], true); have = true;
+        }
+        if (have) ws[nw++] = pend;
+        for (int i = 0; i < nw / 2; ++i) { Prepared t = ws[i]; ws[i] = ws[nw - 1 - i]; ws[nw - 1 - i] = t; }
+        if (nw) { CKE(run_wgrads(h->stream, ws, nw, true)); CKE(prof_mark(h, BP_PROF_WGRAD)); }
+        return hipSuccess;
+    }
     for (int l = L - 1; l >= 1; --l) {
         if (l != 1) { CKE(launch_dgrad(h, h->stream, l, B)); CKE(prof_mark(h, l == L - 1 ? BP_PROF_DGRAD_OUT : BP_PROF_DGRAD_HIDDEN)); }
         if (h->grouped) ws[nw++] = prep_wgrad(h, l, B, l == 1 ? x0 : h->y[l - 1], fused);

*/
