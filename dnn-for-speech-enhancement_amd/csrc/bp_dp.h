// bp_dp.h -- in-library data-parallel exchange (semantics donor: the reference's
// commented-out train_bunch_multi, BP_GPU.cu:775-908: per-GPU forward/backward, gradient SUM,
// one update with n = global bunch, identical parameters on every GPU afterwards).
//
// One process per GPU.  Every rank exports three device allocations through hipIpc and maps its
// peers': the flat parameter arena [W_1|b_1|W_2|b_2|...], the flat gradient buffer (same layout)
// and a small fine-grained flag array.  Per layer and minibatch the exchange is
//     reduce-scatter : rank r READS slice r of every peer's gradient segment (all xGMI links at once)
//     sharded update : delta/W of slice r only (kernUpdatedelta + kernAccSum, DevFunc.cu:313-318,
//                      270-277) -- the momentum state is never replicated
//     all-gather     : the new W slice is WRITTEN into every peer's parameter arena
// in ONE kernel (bp_dp_reduce_update), ordered against the peers' kernels by device-side epoch flags:
// no host synchronisation and no collective library on the data path.  The reference moved the same
// data through GPU 0 with cublasSaxpy / cublasScopy over P2P (BP_GPU.cu:863-904).
//
// Memory-model contract (gfx950; LLVM AMDGPU memory model, system scope):
//   producer : payload stores -> every wave drains vmcnt -> one lane per workgroup system-scope
//              release (buffer_wbl2 sc0 sc1) -> arrival counter -> the LAST workgroup stores the
//              epoch into every peer's flag word (system-scope atomic store)
//   consumer : a one-wave wait kernel polls its OWN flag words (system-scope relaxed loads, bounded
//              spin) -> kernel boundary -> the reading kernel's workgroups each execute one
//              system-scope acquire (buffer_inv sc0 sc1) before their first peer load.
// Spins are bounded by a wall-clock budget; a timeout raises the error word instead of hanging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { BP_DP_MAXRANKS = 8 };
// flag words of one rank: [kind][layer][source rank]
enum { BP_DP_FLAG_GRAD = 0, BP_DP_FLAG_W = 1, BP_DP_FLAG_KINDS = 2 };
#define BP_DP_FLAG_WORDS (BP_DP_FLAG_KINDS * 16 * BP_DP_MAXRANKS)
__host__ __device__ inline int bp_dp_flag_index(int kind, int layer, int src) { return (kind * 16 + layer) * BP_DP_MAXRANKS + src; }

struct DpPeers { unsigned *flags[BP_DP_MAXRANKS]; };

// One wave: after everything earlier on the stream is complete (kernel boundary) and a system-scope
// release, lane p stores `epoch` into word `index` of peer p's flag array.
__global__ void bp_dp_signal(DpPeers peers, int world, int index, unsigned epoch)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int p = threadIdx.x;
    if (p < world) __hip_atomic_store(peers.flags[p] + index, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave: lane p waits until own flag word base+p has reached `epoch` (signed distance, so the
// 32-bit epoch may wrap).  On timeout the error word gets 1000*kind_code + p and the wait gives up
// (the step then produces garbage, which the host reports through bp_dp_check).
__global__ void bp_dp_wait(const unsigned *flags, int base, int world, unsigned epoch, unsigned long long budget_ticks,
                           unsigned *err, unsigned code)
{
    const int p = threadIdx.x;
    if (p < world) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const unsigned v = __hip_atomic_load(flags + base + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - epoch) >= 0) break;
            if (wall_clock64() - t0 > budget_ticks) { atomicExch(err, code * 1000u + (unsigned)p + 1u); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

struct DpReduceArgs {
    const float *grads[BP_DP_MAXRANKS];   // every rank's flat gradient buffer (own entry = local pointer)
    float *params[BP_DP_MAXRANKS];        // every rank's flat parameter arena
    float *delta;                         // own flat momentum arena (only this rank's slices are live)
    unsigned long long lo, hi;            // this rank's slice of the layer segment, flat float indices (multiples of 4)
    unsigned long long w_end;             // flat index where the W part of the segment ends (weight cost applies below it)
    int world, rank;
    float mom, c1, wc, ndiv;              // c1 = (1-m)*lr or lr; ndiv = (float)global bunch
    unsigned *arrive;                     // device counter (zero between launches): last-arriver election
    DpPeers peers; int flag_index; unsigned epoch;
};

// Slice [lo, hi) of one layer: g = sum over ranks (fixed order 0..world-1, so the result does not depend on
// which rank owns the slice), momentum update of delta/W, new W to every rank.  One thread = 4 floats.
template <int WORLD>
__global__ __launch_bounds__(256) void bp_dp_reduce_update(const DpReduceArgs a)
{
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // peers' gradients: drop stale lines
    __syncthreads();
    const int world = WORLD > 0 ? WORLD : a.world;
    const unsigned long long n4 = (a.hi - a.lo) >> 2, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const unsigned long long i = a.lo + 4 * q;
        float4 g[BP_DP_MAXRANKS];
#pragma unroll
        for (int p = 0; p < BP_DP_MAXRANKS; ++p)
            if (p < world) g[p] = *reinterpret_cast<const float4 *>(a.grads[p] + i);
        const float4 w = *reinterpret_cast<const float4 *>(a.params[a.rank] + i);
        const float4 d = *reinterpret_cast<const float4 *>(a.delta + i);
        float4 s = g[0];
#pragma unroll
        for (int p = 1; p < BP_DP_MAXRANKS; ++p)
            if (p < world) { s.x += g[p].x; s.y += g[p].y; s.z += g[p].z; s.w += g[p].w; }
        const float wc = i < a.w_end ? a.wc : 0.0f;                     // (segments are multiples of 64 floats: a float4 never straddles)
        float4 dn, wn;
        dn.x = a.mom * d.x - a.c1 * (s.x / a.ndiv + wc * w.x); wn.x = dn.x + 1.0f * w.x;   // kernUpdatedelta, kernAccSum
        dn.y = a.mom * d.y - a.c1 * (s.y / a.ndiv + wc * w.y); wn.y = dn.y + 1.0f * w.y;
        dn.z = a.mom * d.z - a.c1 * (s.z / a.ndiv + wc * w.z); wn.z = dn.z + 1.0f * w.z;
        dn.w = a.mom * d.w - a.c1 * (s.w / a.ndiv + wc * w.w); wn.w = dn.w + 1.0f * w.w;
        *reinterpret_cast<float4 *>(a.delta + i) = dn;
#pragma unroll
        for (int p = 0; p < BP_DP_MAXRANKS; ++p)
            if (p < world) *reinterpret_cast<float4 *>(a.params[p] + i) = wn;
    }
    // publish: drain, release at system scope once per workgroup, then the last workgroup raises the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (last) {
        if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)threadIdx.x < world)
            __hip_atomic_store(a.peers.flags[threadIdx.x] + a.flag_index, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Momentum slices of the peers into the local arena (bp_get_deltas on a data-parallel handle): plain copy.
__global__ void bp_dp_copy(float *dst, const float *src, unsigned long long n4)
{
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride)
        reinterpret_cast<float4 *>(dst)[q] = reinterpret_cast<const float4 *>(src)[q];
}
