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
// (round 6: or, selectable per group, the PUSH form -- every rank WRITES slice r of its segment into rank r's receive buffer,
// bp_dp_push below, and the owner sums its local slots in the same order: the same bits, posted writes instead of round trips)
// in ONE kernel (bp_dp_reduce_update), ordered against the peers' kernels by device-side epoch flags:
// no host synchronisation and no collective library on the data path.  The reference moved the same
// data through GPU 0 with cublasSaxpy / cublasScopy over P2P (BP_GPU.cu:863-904).
//
// Memory-model contract (gfx950; LLVM AMDGPU memory model).  No per-workgroup ACQUIRE and no fence in the exchange kernel:
// a system-scope acquire (buffer_inv sc0 sc1) drops the whole L2 of the executing XCD, and a release + acquire per
// workgroup made the exchange kernel 5x slower and evicted the L2 under the GEMMs that run beside it (rocprofv3, round 2).
// Both directions move their data with system-scope WRITE-THROUGH accesses instead, so that nothing is ever left dirty in a
// writer's L2 and nothing stale can be read through a reader's:
//   * gradient tiles are written with system-scope (sc0 sc1) write-through stores by the kernel that counts its tiles
//     (bp_kernels.h epilogue_block<EPI_WGRAD_STORE>; the event path's kernels end with the kernel-boundary release), into
//     FINE-GRAINED device memory (never cached dirty; peers' mappings of it are uncached), and are read with system-scope
//     16-byte loads, so a reader can neither see a stale L2 line nor leave one;
//   * new weights are written with system-scope WRITE-THROUGH (sc0 sc1) 16-byte stores into the (cacheable) parameter
//     arenas: nothing stays dirty in the writer's L2; the owner's L2 is kept coherent for its local memory by the
//     fabric's probes, and its L1s are invalidated at the next kernel boundary;
//   * ordering: every storing wave drains vmcnt (write-through stores are complete when acknowledged) ->
//     __syncthreads -> agent-scope arrival counter -> the LAST workgroup stores the epoch into every peer's flag word
//     (system-scope atomic store; flag words are fine-grained memory);
//   * consumer: a one-wave wait kernel polls its OWN flag words (system-scope relaxed loads, bounded spin), and the
//     kernel boundary behind it orders the reading kernel after the poll.
// Spins are bounded by a wall-clock budget; a timeout raises the error word instead of hanging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { BP_DP_MAXRANKS = 8 };
// flag words of one rank: [kind][layer][source rank]
enum { BP_DP_FLAG_GRAD = 0, BP_DP_FLAG_W = 1, BP_DP_FLAG_PROBE = 2 /* attach-time self-test: layer index = direction */, BP_DP_FLAG_KINDS = 3 };
#define BP_DP_FLAG_WORDS (BP_DP_FLAG_KINDS * 16 * BP_DP_MAXRANKS)
__host__ __device__ inline int bp_dp_flag_index(int kind, int layer, int src) { return (kind * 16 + layer) * BP_DP_MAXRANKS + src; }

struct DpPeers { unsigned *flags[BP_DP_MAXRANKS]; };

// One wave: after everything earlier on the stream is complete (kernel boundary: the producing kernel's L2 write-back
// has happened, and the gradient buffer is write-through fine-grained memory anyway), lane p stores `epoch` into word
// `index` of peer p's flag array.
__global__ void bp_dp_signal(DpPeers peers, int world, int index, unsigned epoch)
{
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
}

// The same two kernels for SEVERAL flag words at once (the layers of one grouped weight-gradient launch).
struct DpIdx { int n; int index[16]; };
__global__ void bp_dp_signal_n(DpPeers peers, int world, DpIdx idx, unsigned epoch)
{
    const int p = threadIdx.x;
    if (p < world)
        for (int i = 0; i < idx.n; ++i) __hip_atomic_store(peers.flags[p] + idx.index[i], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void bp_dp_wait_n(const unsigned *flags, DpIdx bases, int world, unsigned epoch, unsigned long long budget_ticks, unsigned *err, unsigned code)
{
    const int p = threadIdx.x;
    if (p < world) {
        const unsigned long long t0 = wall_clock64();
        for (int i = 0; i < bases.n; ++i)
            for (;;) {
                const unsigned v = __hip_atomic_load(flags + bases.index[i] + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((int)(v - epoch) >= 0) break;
                if (wall_clock64() - t0 > budget_ticks) { atomicExch(err, code * 1000u + (unsigned)p + 1u); return; }
                __builtin_amdgcn_s_sleep(16);
            }
    }
}

// Exchange stream, per layer: (1) wait until the local weight-gradient launch has counted `target` tiles of this layer's
// segment (EpiArgs::done, bp_wgrad_dma.h; signed distance, the counter only grows) -- every counted tile was written with
// write-through stores that its waves drained before counting, so it is in memory --, (2) tell every rank, (3) wait for every
// rank.  One wave, one launch: it replaces an event record on the main stream (which cost it a ~7 us bubble), a stream-wait, a
// signal kernel and a wait kernel.
__global__ void bp_dp_sync(const unsigned *done, unsigned target, DpPeers peers, const unsigned *flags, int world, int sig_index, int wait_base,
                           unsigned epoch, unsigned long long budget_ticks, unsigned *err, unsigned code)
{
    const int p = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    if (p == 0 && done) {
        for (;;) {
            const unsigned v = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - target) >= 0) break;
            if (wall_clock64() - t0 > budget_ticks) { atomicExch(err, 4000u + 1u); ok = 0; break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __builtin_amdgcn_s_barrier();          // (one wave: orders lane 0's poll in front of the other lanes' stores)
    // the local segment never completed: do NOT tell the peers it is ready (they would reduce an incomplete segment before the
    // error surfaces); their own waits time out and report this rank instead
    if (!__builtin_amdgcn_readfirstlane(ok)) return;
    if (p < world) {
        __hip_atomic_store(peers.flags[p] + sig_index, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (;;) {
            const unsigned v = __hip_atomic_load(flags + wait_base + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - epoch) >= 0) break;
            if (wall_clock64() - t0 > budget_ticks) { atomicExch(err, code * 1000u + (unsigned)p + 1u); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
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

#ifndef BP_DP_UNROLL
#define BP_DP_UNROLL 4
#endif
// 16-byte system-scope accesses through a buffer descriptor (aux: sc0 = 1, sc1 = 16)
typedef float bp_f32x4 __attribute__((ext_vector_type(4)));
#define BP_AUX_SYS (1 | 16)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bp_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// Slice [lo, hi) of one layer: g = sum over ranks (fixed order 0..world-1, so the result does not depend on
// which rank owns the slice), momentum update of delta/W, new W to every rank.  One thread = 4 floats per pass.
// round to nearest even, NaN stays NaN (as bp_bf16.h f2bf) / back
__device__ __forceinline__ unsigned bp_dp_f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// GBF16: the sources are bf16 receive slots (push form with bf16 gradient segments): grads[p] are then bf16 arrays indexed like the
// float ones (element i of the slice at 2*i bytes)
template <int WORLD, bool GBF16 = false>
__global__ __launch_bounds__(256) void bp_dp_reduce_update(const DpReduceArgs a)
{
    const int world = WORLD > 0 ? WORLD : a.world;
    const unsigned bytes = (unsigned)((a.hi - a.lo) * 4);
    __amdgpu_buffer_rsrc_t rg[BP_DP_MAXRANKS], rw[BP_DP_MAXRANKS];
#pragma unroll
    for (int p = 0; p < BP_DP_MAXRANKS; ++p)
        if (p < world) {
            rg[p] = GBF16 ? bp_rsrc(reinterpret_cast<const unsigned short *>(a.grads[p]) + a.lo, bytes / 2) : bp_rsrc(a.grads[p] + a.lo, bytes);
            rw[p] = bp_rsrc(a.params[p] + a.lo, bytes);
        }
    const float *w_own = a.params[a.rank] + a.lo;
    float *d_own = a.delta + a.lo;
    const unsigned long long n4 = (a.hi - a.lo) >> 2, stride = (unsigned long long)gridDim.x * blockDim.x;
    // U float4 per thread and pass: U * world peer loads in flight per lane (xGMI latency), few workgroups resident
    // (the exchange runs beside the next minibatch's GEMMs and must not crowd them out of the CUs' memory pipes)
    constexpr int U = BP_DP_UNROLL;
    for (unsigned long long q0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += U * stride) {
        float4 w[U], d[U];
        bp_f32x4 g[U][BP_DP_MAXRANKS];
        typedef unsigned bp_u32x2 __attribute__((ext_vector_type(2)));
        bp_u32x2 gh[GBF16 ? U : 1][BP_DP_MAXRANKS];       // bf16 sources: kept as loaded until the sum (a conversion next to its load makes the compiler wait for every load in turn)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long q = q0 + u * stride < n4 ? q0 + u * stride : q0;     // (clamped: loads stay unconditional)
            w[u] = *reinterpret_cast<const float4 *>(w_own + 4 * q);
            d[u] = *reinterpret_cast<const float4 *>(d_own + 4 * q);
#pragma unroll
            for (int p = 0; p < BP_DP_MAXRANKS; ++p)
                if (p < world) {
                    if constexpr (GBF16) gh[u][p] = __builtin_bit_cast(bp_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rg[p], (unsigned)(q * 8), 0, BP_AUX_SYS));
                    else
                    g[u][p] = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg[p], (unsigned)(q * 16), 0, BP_AUX_SYS));
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long q = q0 + u * stride;
            if (q >= n4) break;
            if constexpr (GBF16) {
#pragma unroll
                for (int p = 0; p < BP_DP_MAXRANKS; ++p)
                    if (p < world) {
                        const bp_u32x2 h = gh[u][p];
                        g[u][p].x = __uint_as_float(h.x << 16); g[u][p].y = __uint_as_float(h.x & 0xFFFF0000u);
                        g[u][p].z = __uint_as_float(h.y << 16); g[u][p].w = __uint_as_float(h.y & 0xFFFF0000u);
                    }
            }
            bp_f32x4 s = g[u][0];
#pragma unroll
            for (int p = 1; p < BP_DP_MAXRANKS; ++p)
                if (p < world) s += g[u][p];
            const float wc = a.lo + 4 * q < a.w_end ? a.wc : 0.0f;      // (segments are multiples of 64 floats: a float4 never straddles)
            float4 dn; bp_f32x4 wn;
            dn.x = a.mom * d[u].x - a.c1 * (s.x / a.ndiv + wc * w[u].x); wn.x = dn.x + 1.0f * w[u].x;   // kernUpdatedelta, kernAccSum
            dn.y = a.mom * d[u].y - a.c1 * (s.y / a.ndiv + wc * w[u].y); wn.y = dn.y + 1.0f * w[u].y;
            dn.z = a.mom * d[u].z - a.c1 * (s.z / a.ndiv + wc * w[u].z); wn.z = dn.z + 1.0f * w[u].z;
            dn.w = a.mom * d[u].w - a.c1 * (s.w / a.ndiv + wc * w[u].w); wn.w = dn.w + 1.0f * w[u].w;
            *reinterpret_cast<float4 *>(d_own + 4 * q) = dn;      // (nontemporal accesses for the momentum stream: 0.2657 vs 0.2537 ms, profiles/r05_dp_world1.txt)
#pragma unroll
            for (int p = 0; p < BP_DP_MAXRANKS; ++p)
                if (p < world)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, wn), rw[p], (unsigned)(q * 16), 0, BP_AUX_SYS);
        }
    }
    // publish: every wave drains its write-through stores, then the last workgroup to arrive raises the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {
        if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)threadIdx.x < world)
            __hip_atomic_store(a.peers.flags[threadIdx.x] + a.flag_index, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Push form of the reduce-scatter (BP_DP_TRANSPORT_NATIVE_PUSH): this rank's gradient segment of one layer, slice by slice, into the
// owners' receive buffers -- slot `rank` of the layer's region there -- with system-scope write-through 16-byte stores (posted
// writes over xGMI: nothing waits for a round trip, which is what a reduce-scatter by peer READS does per request); the last
// workgroup then raises GRAD(layer, rank) in every owner's flag array: "my contribution to your slice has landed".  The owner's
// bp_dp_reduce_update sums its `world` LOCAL slots in the fixed order 0..world-1 (the pull form's order: same bits).
struct DpPushArgs {
    const float *grad;                    // own flat gradient buffer
    float *recv[BP_DP_MAXRANKS];          // every rank's receive buffer
    unsigned long long seg;               // flat index of the layer's segment in `grad`
    unsigned long long n4, per4;          // float4 count of the segment / of one slice (the last slices may be short or empty)
    unsigned long long roff;              // float index of the layer's region in a receive buffer: `world` slots of 4*per4 floats
    int world, rank;
    unsigned *arrive;
    DpPeers peers; int flag_index; unsigned epoch;
};
template <bool GBF16 = false>
__global__ __launch_bounds__(256) void bp_dp_push(const DpPushArgs a)
{
    const __amdgpu_buffer_rsrc_t rg = bp_rsrc(a.grad + a.seg, (unsigned)(a.n4 * 16));
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (int r = 0; r < a.world; ++r) {
        const unsigned long long lo = a.per4 * r < a.n4 ? a.per4 * r : a.n4, hi = a.per4 * (r + 1) < a.n4 ? a.per4 * (r + 1) : a.n4;
        if (hi <= lo) continue;
        // (bf16 segments: the same float indices address 2-byte elements, i.e. a slot of half the bytes at half the offset)
        const __amdgpu_buffer_rsrc_t rw = GBF16 ? bp_rsrc(reinterpret_cast<unsigned short *>(a.recv[r]) + a.roff + (unsigned long long)a.rank * 4 * a.per4, (unsigned)((hi - lo) * 8))
                                                : bp_rsrc(a.recv[r] + a.roff + (unsigned long long)a.rank * 4 * a.per4, (unsigned)((hi - lo) * 16));
        constexpr int U = 4;
        for (unsigned long long q0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q0 < hi - lo; q0 += U * stride) {
            bp_f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long q = q0 + u * stride < hi - lo ? q0 + u * stride : q0;
                v[u] = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (unsigned)((lo + q) * 16), 0, BP_AUX_SYS));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long q = q0 + u * stride;
                if (q >= hi - lo) break;
                if constexpr (GBF16) {
                    typedef unsigned bp_u32x2 __attribute__((ext_vector_type(2)));
                    bp_u32x2 h;
                    h.x = bp_dp_f2bf(v[u].x) | (bp_dp_f2bf(v[u].y) << 16); h.y = bp_dp_f2bf(v[u].z) | (bp_dp_f2bf(v[u].w) << 16);
                    __builtin_amdgcn_raw_buffer_store_b64(h, rw, (unsigned)(q * 8), 0, BP_AUX_SYS);
                } else
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v[u]), rw, (unsigned)(q * 16), 0, BP_AUX_SYS);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {
        if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)threadIdx.x < a.world)
            __hip_atomic_store(a.peers.flags[threadIdx.x] + a.flag_index, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Momentum slices of the peers into the local arena (bp_get_deltas on a data-parallel handle): plain copy.
__global__ void bp_dp_copy(float *dst, const float *src, unsigned long long n4)
{
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (rare call: one system-scope acquire per workgroup is fine here)
    __syncthreads();
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride)
        reinterpret_cast<float4 *>(dst)[q] = reinterpret_cast<const float4 *>(src)[q];
}


// ------------------------------------------------------------------ attach-time self-test of the memory-model contract
// The exchange above leans on two properties of the platform that cannot be derived from the programming model alone
// (header comment): (W) weights a PEER writes into this rank's cacheable arena with system-scope write-through stores are
// what this rank's next kernels read with plain cached loads once its flag has risen; (G) gradients this rank's kernels
// wrote with plain stores into fine-grained memory are what a peer's system-scope loads return once ITS flag has risen.
// bp_dp_attach checks both on the actual devices of the group, with the product's own access flavours and ordering
// recipe, on probe buffers of the same allocation kinds, before any training step relies on them (bp_dp.hip,
// dp_selftest).  If (W) fails as is, the group falls back to an explicit system-scope acquire on every XCD behind each
// wait (bp_dp_l2_invalidate); if it still fails, bp_dp_attach fails.
#define BP_DP_PROBE_FLOATS (64 * 1024)            /* 256 KB per probe buffer */
__device__ __forceinline__ float bp_probe_value(unsigned round, unsigned writer, unsigned idx)
{
    return (float)((round * 131u + writer * 17u + idx * 3u) & 0xFFFFFu);
}
// plain cached reads of the whole probe from every XCD (leaves its lines in the L2s / L1s of this device)
__global__ void bp_dp_probe_touch(const float *probe, float *sink)
{
    float s = 0.f;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < BP_DP_PROBE_FLOATS; i += gridDim.x * blockDim.x) s += probe[i];
    if (s == 1.2345e-30f) *sink = s;
}
// writer `rank` fills ITS slice of every rank's probe with system-scope write-through 16-byte stores, drains, and raises
// word (PROBE, 0, rank) of every rank's flag array: the product's all-gather recipe (bp_dp_reduce_update).  One workgroup.
__global__ __launch_bounds__(256) void bp_dp_probe_push(DpReduceArgs a, unsigned round)
{
    const unsigned per = BP_DP_PROBE_FLOATS / BP_DP_MAXRANKS, lo = a.rank * per;
    for (int p = 0; p < a.world; ++p) {
        __amdgpu_buffer_rsrc_t rw = bp_rsrc(a.params[p] + lo, per * 4);
        for (unsigned q = threadIdx.x; q < per / 4; q += blockDim.x) {
            bp_f32x4 v;
            v.x = bp_probe_value(round, a.rank, lo + 4 * q); v.y = bp_probe_value(round, a.rank, lo + 4 * q + 1);
            v.z = bp_probe_value(round, a.rank, lo + 4 * q + 2); v.w = bp_probe_value(round, a.rank, lo + 4 * q + 3);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rw, q * 16, 0, BP_AUX_SYS);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < a.world)
        __hip_atomic_store(a.peers.flags[threadIdx.x] + a.flag_index, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// owner: plain cached loads of the whole probe; counts words that are not what writer (idx / per) stored in `round`
__global__ void bp_dp_probe_check(const float *probe, int world, unsigned round, unsigned *bad)
{
    const unsigned per = BP_DP_PROBE_FLOATS / BP_DP_MAXRANKS;
    unsigned n = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per * (unsigned)world; i += gridDim.x * blockDim.x)
        if (probe[i] != bp_probe_value(round, i / per, i)) ++n;
    if (n) atomicAdd(bad, n);
}
// owner: plain stores of a fresh pattern into its own fine-grained probe (what the wgrad kernels do to the gradient buffer)
__global__ void bp_dp_probe_fill(float *probe, unsigned round, unsigned rank)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < BP_DP_PROBE_FLOATS; i += gridDim.x * blockDim.x)
        probe[i] = bp_probe_value(round, rank, i);
}
// the same fill with the product's IN-KERNEL hand-off (bp_wgrad_dma.h, EpiArgs::done): plain stores, every wave drains, one
// lane per workgroup counts -- no kernel boundary between these stores and the readers
__global__ __launch_bounds__(256) void bp_dp_probe_fill_count(float *probe, unsigned round, unsigned rank, unsigned *done)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < BP_DP_PROBE_FLOATS; i += gridDim.x * blockDim.x)
        __hip_atomic_store(probe + i, bp_probe_value(round, rank, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // write-through, as the tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// reader: system-scope 16-byte loads of slice `rank` of every rank's fine-grained probe (the reduce-scatter read)
__global__ void bp_dp_probe_check_remote(DpReduceArgs a, unsigned round, unsigned *bad)
{
    const unsigned per = BP_DP_PROBE_FLOATS / BP_DP_MAXRANKS, lo = a.rank * per;
    unsigned n = 0;
    for (int p = 0; p < a.world; ++p) {
        __amdgpu_buffer_rsrc_t rg = bp_rsrc(a.grads[p] + lo, per * 4);
        for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < per / 4; q += gridDim.x * blockDim.x) {
            const bp_f32x4 v = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, q * 16, 0, BP_AUX_SYS));
            n += v.x != bp_probe_value(round, p, lo + 4 * q) || v.y != bp_probe_value(round, p, lo + 4 * q + 1) ||
                 v.z != bp_probe_value(round, p, lo + 4 * q + 2) || v.w != bp_probe_value(round, p, lo + 4 * q + 3);
        }
    }
    if (n) atomicAdd(bad, n);
}
// Fallback acquire (dp "acquire mode" 1): one system-scope acquire (buffer_inv sc0 sc1) per workgroup; the launch has
// enough workgroups to land on every XCD, so every L2 of this device drops its non-coherent lines before the next kernel.
__global__ void bp_dp_l2_invalidate()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
