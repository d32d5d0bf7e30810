// tools/wgrad_sk_probe.h -- DEVELOPMENT PROBE, not part of the product (measured round 4: SLOWER than the shipped kernel, see the end of this comment): the grouped
// weight gradient + fused momentum update as ONE PERSISTENT launch whose last round is k-balanced.
//
//   G_l = Y_{l-1}^T . dEdX_l for every layer of the step (SgemmNT, DevFunc.h:57-67; BP_GPU.cu:642), then kernUpdatedelta +
//   kernAccSum (DevFunc.cu:313-318, 270-277) on the tile while it is in registers, bias gradient by m-tile 0 (kernAccSumrow,
//   DevFunc.cu:224-242) -- the work of bp_wgrad_dma.h, cut differently.
//
// Why: bp_wgrad_dma launches one workgroup per 64x64 tile.  C2 has 3648 tiles for 1024 resident workgroup slots (4 per CU) =
// 3.56 rounds, and the in-kernel trace (profiles/r03_wgrad_trace.txt) shows the last 0.56 round draining for ~16 us at
// 3.0 -> 0.1 workgroups per CU.  Here the launch is 4 workgroups per CU that never exit.  Every XCD owns the tile list the old
// XCD-aware map gave it (so the operand panels keep their L2) and its WX = grid/8 workgroups walk it the way the dispatcher
// would have -- workgroup i takes tiles i, i+WX, i+2WX, ... for as many FULL rounds as the list has (neighbours work on
// neighbouring tiles at the same time: the activation / dEdX panels they share are L2 hits; a first version that gave every
// workgroup one contiguous run of the list was 6 % SLOWER than one workgroup per tile for exactly that reason).  What is left
// (C2: 72 tiles per XCD for 128 workgroups) is cut into EQUAL runs of consecutive k-tiles (16 frames of one output tile; C2:
// 9 each), so all workgroups finish together.  A run that does not hold a tile's first k-tile computes its piece FIRST and
// publishes the partial sums (write-through stores, one flag); the workgroup that holds the tile's first k-tiles (always the
// LAST thing of its run) adds its successors' pieces in list order and finishes the tile.  Summation order of a split tile:
// pieces in k order, each (two chains) -- fixed by the geometry, independent of timing.
// The operand pipeline of bp_wgrad_dma is kept (global_load_lds_dwordx4 into a 4-stage ring, three k-tiles ahead, one raw
// s_barrier per k-tile, counted vmcnt) and now runs ACROSS tile boundaries: no per-tile prologue.
//
// Two bodies.  A whole tile with at least three more k-tiles of the workgroup's sequence behind it runs the fully unrolled
// body (`mid`).  Everything else -- pieces of shared tiles, the last whole tile, the few m-tile-0 tiles that also sum the bias
// gradient -- runs the loop body (`seg`), where hipcc's own waits around the W/delta loads are conservative (vmcnt(0)).
//
// MEASURED (round 4, profiles/r04_wgrad_streamk_probe.txt): results equal the shipped kernel's to fp32 summation order, and the launch
// is 6-10 % SLOWER (C2: 88.1 vs 83.7 us back to back; 93 vs 84 on C3).  The in-kernel trace shows why equal shares do not end
// together: (1) the SIMD issue arbiter prefers the OLDEST wave, so of the 4 workgroups of a CU the first-started finishes its
// share at 64 us and the last-started at 82 (mean by start rank 64.0 / 71.7 / 75.9 / 81.7); rotating s_setprio through four
// levels offset by that rank equalises the ranks (74.6 / 73.4 / 74.8 / 80.6) but (2) whole CUs differ by +-15 % as well (ends
// 62..90 us), and a static partition has nothing left to hand to the fast ones.  The hardware dispatcher IS the dynamic
// scheduler this would need; the 16 us tail of one-workgroup-per-tile costs less than any static cut.  Not in the product.
#pragma once
#include "../dnn-for-speech-enhancement_amd/csrc/bp_kernels.h"
#include "../dnn-for-speech-enhancement_amd/csrc/bp_wgrad_dma.h"

#define BP_SK_MAXP 4
#ifndef BP_SK_DELAY
#define BP_SK_DELAY 1
#endif
#ifndef BP_SK_PRIO_STEP
#define BP_SK_PRIO_STEP 4       // k-tiles per priority level
#endif
struct SkArgs {
    GemmArgs g[BP_SK_MAXP];
    EpiArgs e[BP_SK_MAXP];
    int cum[BP_SK_MAXP + 1];     // XCD-local tile index ranges: problem p owns [cum[p], cum[p+1]) on every XCD (tiles_p / 8 each)
    int n;                       // problems
    int rounds;                  // full rounds: tiles [0, rounds * grid/8) of every XCD's list, one per workgroup and round
    float *ws;                   // [grid][WS] partial tiles handed to the finishing workgroup
    unsigned *flags;             // [grid] epoch of the last partial published by that workgroup
    unsigned epoch;              // this launch (flags are never reset: the value only grows)
    unsigned *err;               // raised when a finisher gives up waiting (never hang the device)
    unsigned long long budget;   // ... after this many wall_clock64 ticks (100 MHz)
#ifdef BP_SK_TRACE               // development only (tools/wgrad_sk_probe.hip): per-workgroup timestamps, 16 slots each
    unsigned long long *trace;
#endif
};

template <int KTOT, bool STORE>
struct WgradSk {
    using Base = WgradDma<16, 4, 4, KTOT, STORE>;
    static constexpr int EPI = STORE ? EPI_WGRAD_STORE : EPI_WGRAD_UPDATE;
    static constexpr int BM = 64, BN = 64, BK = 16, ST = 4, D = 3, NT = KTOT / BK;
    static constexpr int STAGE = Base::STAGE, A_STAGE = Base::A_STAGE, RING = ST * STAGE;
    static constexpr int SMEM = RING + 4 * BN;                     // + bias-gradient scratch OUTSIDE the ring (the ring never drains)
    static constexpr int WS = BM * BN + 256;                       // floats per workgroup: accumulator image + per-thread bias sums
    // VMEM operations a wave issues besides the 2 DMA instructions per k-tile, as groups that later operand waits must count:
    // the W/delta prefetch (32 loads, fused update only) and the tile's stores (32 W/delta, or 16 gradient)
    static constexpr int XN = STORE ? 16 : 32;
    static constexpr int WD_T = NT - D;                            // k-tile whose iteration issues the W/delta prefetch (behind the request of the tile's last k-tile)
    static_assert(NT >= 8 && (NT & (NT - 1)) == 0, "k-tiles per tile: power of two");
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;

    // A workgroup's sequence of k-tiles, indexed v = 0 .. nv-1: first its `rounds` whole tiles (tile wi + r*WX of the XCD's
    // list), then its run [rb, re) of the k-tiles of the remaining tiles (counted from the first remaining tile).
    struct Ctx {
        const SkArgs &a; float *smem;
        int tid, lane, wave, x, a_off, b_off, kh, r_row, c_col;
        int wi, WX, nfull, rb, nv;                                 // nfull = rounds * NT
        int iv; const float *ia, *ib; int ilda, ildb;              // issue cursor: next k-tile to request, its tile's operand panels
        int hot;                                                   // operand waits that still have a group of XN younger accesses in their window
        int pbase;                                                 // priority rotation (BP_SK_PRIO): rank among the CU's workgroups + tiles done
    };
    // XCD-local tile index and k-tile of sequence position v
    static __device__ __forceinline__ void locate(const Ctx &c, int v, int &j, int &kt)
    {
        if (v < c.nfull) { j = c.wi + c.WX * (v / NT); kt = v & (NT - 1); }
        else { const int u = c.rb + (v - c.nfull); j = c.a.rounds * c.WX + u / NT; kt = u & (NT - 1); }
    }
    static __device__ __forceinline__ void decode(const SkArgs &a, int x, int j, int &p, int &tile_m, int &tile_n)
    {
        p = 0;
        while (p + 1 < a.n && j >= a.cum[p + 1]) ++p;
        const int jj = j - a.cum[p], tn = a.g[p].tiles_n, tm = a.g[p].tiles_m;
        if ((tn & 7) == 0) { const int per = tn >> 3; tile_n = x * per + jj % per; tile_m = jj / per; }
        else { const int b = 8 * jj + x; tile_m = b % tm; tile_n = b / tm; }
    }
    // The SIMD's issue arbiter prefers the OLDEST wave at equal priority, so of the 4 workgroups that share a CU the one that
    // started first runs fastest and equal shares of work do not end together (in-kernel trace: mean end 64.8 / 72.5 / 76.6 /
    // 82.5 us by start rank).  Rotating s_setprio through the four levels, offset by the rank, gives every workgroup every
    // level for the same share of its k-tiles.
    static __device__ __forceinline__ void set_prio(int p)
    {
#if defined(BP_SK_PRIO) && BP_SK_PRIO
        switch (p & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#else
        (void)p;
#endif
    }
    template <int N> static __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

    static __device__ __forceinline__ void issue_decode(Ctx &c)
    {
        int j, kt, p, tm, tn;
        locate(c, c.iv, j, kt);
        decode(c.a, c.x, j, p, tm, tn);
        c.ilda = c.a.g[p].lda; c.ildb = c.a.g[p].ldb;
        c.ia = c.a.g[p].A + tm * BM; c.ib = c.a.g[p].B + tn * BN;
    }
    // request k-tile KT of the issue cursor's tile into ring stage (iv & 3).  CHECK: 0 = the NEXT request stays in this tile (no
    // test), 1 = it may start another tile (test), 2 = it does.
    template <int CHECK = 1>
    static __device__ __forceinline__ void issue(Ctx &c, int kt)
    {
        const int st = c.iv & (ST - 1);
        const float *ga = c.ia + (size_t)(kt * BK + c.r_row) * c.ilda + c.c_col;
        const float *gb = c.ib + (size_t)(kt * BK + c.r_row) * c.ildb + c.c_col;
        __builtin_amdgcn_global_load_lds((glb_ptr)ga, (lds_ptr)(c.smem + st * STAGE + c.wave * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)gb, (lds_ptr)(c.smem + st * STAGE + A_STAGE + c.wave * 256), 16, 0, 0);
        ++c.iv;
        if constexpr (CHECK == 1) { if (c.iv < c.nv && (kt == NT - 1 || c.iv == c.nfull)) issue_decode(c); }
        if constexpr (CHECK == 2) issue_decode(c);
    }
    // (loop body) the k-tile index of the issue cursor is not known statically
    static __device__ __forceinline__ void issue_any(Ctx &c)
    {
        int j, kt;
        locate(c, c.iv, j, kt);
        issue<1>(c, kt);
    }
    static __device__ __forceinline__ void bias_rows(const Ctx &c, int st, float &bsum)
    {
        const float *bs = c.smem + st * STAGE + A_STAGE + (c.tid >> 6) * (BK / 4) * BN + (c.tid & 63);
#pragma unroll
        for (int k = 0; k < BK / 4; ++k) bsum += bs[k * BN];
    }

    // ---- whole tile, fully unrolled: sequence positions v .. v+NT-1 are k-tiles 0 .. NT-1 of one tile, and at least D more
    // positions follow, so every iteration requests the position D ahead unconditionally: k-tile T+D of this tile, or k-tile
    // T+D-NT of the next one (iteration NT-D-1 makes the last request of this tile and moves the cursor on).
    template <int T>
    static __device__ __forceinline__ void mid(Ctx &c, int v, const EpiArgs &e, int mb, int nb, f32x16 (&acc)[2], EpiPre &pre)
    {
        if constexpr (T < NT) {
            if constexpr (T < D) { if (c.hot > T) vmwait<2 * (D - 1) + XN>(); else vmwait<2 * (D - 1)>(); }    // the previous tile's stores
            else if constexpr (!STORE && T > WD_T) vmwait<2 * (D - 1) + 32>();                                    // this tile's W/delta prefetch
            else vmwait<2 * (D - 1)>();
            if constexpr (T % BP_SK_PRIO_STEP == 0) set_prio(c.pbase + T / BP_SK_PRIO_STEP);
            __builtin_amdgcn_s_barrier();
            // The LAST iteration's request is made by the caller behind the tile's update instead: hipcc waits vmcnt(0) in front of
            // the first use of the W/delta registers (it does not count past the DMA instructions), i.e. for the youngest request --
            // which then is one whole k-tile old instead of brand new
            if constexpr (T < NT - D) issue<(T == NT - D - 1) ? 2 : 0>(c, T + D);       // k-tile T+D of this tile (the last one moves the cursor on)
            else if constexpr (T != NT - 1 || !BP_SK_DELAY) issue_any(c);                // the first k-tiles of whatever follows in the sequence
            if constexpr (!STORE && T == WD_T) epilogue_fetch<EPI_WGRAD_UPDATE, 0, 16>(e, mb, nb, c.lane, pre);
            const int st = (v + T) & (ST - 1);
            Base::multiply(c.smem, st, c.a_off, c.b_off, c.kh, acc);
            mid<T + 1>(c, v, e, mb, nb, acc, pre);
        }
    }

    // ---- bias gradient (kernAccSumrow) of the dEdX panel this tile streamed, then the tile's update / store.  Called from
    // straight-line code behind the unrolled body.
    static __device__ __forceinline__ void finish(const Ctx &c, const EpiArgs &e, int n0, int mb, int nb, bool do_bias, float bsum,
                                                  const f32x16 &acc, const EpiPre &pre)
    {
        if (do_bias) {
            float *red = c.smem + RING;                            // scratch behind the ring
            red[(c.tid >> 6) * BN + (c.tid & 63)] = bsum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c.tid < BN && n0 + c.tid < e.n_limit) {
                const float s = (red[c.tid] + red[BN + c.tid]) + (red[2 * BN + c.tid] + red[3 * BN + c.tid]);
                const int n = n0 + c.tid;
                if constexpr (STORE) {
                    e.bias_g[n] = s;
                } else {
                    const float d = e.mom * e.bias_d[n] - e.c1 * (s / e.ndiv + 0.0f * e.bias_w[n]);
                    e.bias_d[n] = d;
                    e.bias_w[n] = d + 1.0f * e.bias_w[n];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // (red is rewritten by the next bias tile)
        }
        // epilogue_block<EPI> of bp_kernels.h without its "block outside the matrix" exit: the host only takes this kernel
        // when every extent is a multiple of 64, and a path on which the W/delta registers are never read would make the
        // compiler guard them with a vmcnt(0) at the top of every tile
        float *cw = uniform_ptr(e.C + (size_t)mb * e.ldc + nb);
        const unsigned ldc = __builtin_amdgcn_readfirstlane(e.ldc);
        const unsigned lob = 4u * ((unsigned)(4 * (c.lane >> 5)) * (unsigned)e.ldc + (unsigned)(c.lane & 31));   // bytes
        if constexpr (STORE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * ldc;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cw + ro)) + lob) = acc[r];
            }
        } else {
            float *cd = uniform_ptr(e.aux2 + (size_t)mb * e.ldc + nb);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * ldc;
                const float w = pre.p0[r];
                const float d = e.mom * pre.p1[r] - e.c1 * (acc[r] / e.ndiv + e.wc * w);          // kernUpdatedelta (DevFunc.cu:313-318)
                *reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cd + ro)) + lob) = d;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(sgpr_row_base(cw + ro)) + lob) = d + 1.0f * w;   // kernAccSum (DevFunc.cu:270-277)
            }
        }
    }

    // ---- k-tiles [kt0, kt1) of one tile by the loop body.  FIN: this workgroup finishes the tile (the W/delta prefetch goes
    // behind the request of the piece's last k-tile).  Two instantiations, so that on every path the compiler sees the
    // prefetch followed by its use before the tile loop closes (otherwise it protects the registers with a vmcnt(0) per tile).
    template <bool FIN>
    static __device__ __forceinline__ void seg(Ctx &c, int &v, int kt0, int kt1, const EpiArgs &e, int mb, int nb, bool do_bias, float &bsum,
                                               f32x16 (&acc)[2], EpiPre &pre)
    {
        const int wd_kt = (kt1 - D > kt0) ? kt1 - D : kt0;
        for (int kt = kt0; kt < kt1; ++kt, ++v) {
            // this wave's two DMA pieces of position v have landed: everything it issued later may still be in flight -- the
            // DMAs of positions v+1, v+2 (fewer at the end of the sequence) and at most ONE counted group of XN plain accesses
            const int ahead = c.nv - 1 - v;
            if (c.hot > 0) {
                if (ahead >= 2) vmwait<4 + XN>(); else if (ahead == 1) vmwait<2 + XN>(); else vmwait<XN>();
                --c.hot;
            } else {
                if (ahead >= 2) vmwait<4>(); else if (ahead == 1) vmwait<2>(); else vmwait<0>();
            }
            if ((kt & (BP_SK_PRIO_STEP - 1)) == 0 || kt == kt0) set_prio(c.pbase + kt / BP_SK_PRIO_STEP);
            __builtin_amdgcn_s_barrier();                           // ... everybody's; and stage (v-1)&3 is no longer read
            if (c.iv < c.nv) issue_any(c);
            if constexpr (FIN && !STORE) {
                if (kt == wd_kt) { epilogue_fetch<EPI_WGRAD_UPDATE, 0, 16>(e, mb, nb, c.lane, pre); c.hot = D; }
            }
            const int st = v & (ST - 1);
            if (do_bias) bias_rows(c, st, bsum);
            Base::multiply(c.smem, st, c.a_off, c.b_off, c.kh, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
    }

    static __device__ __forceinline__ void run(const SkArgs &a, float *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave >> 1, wn = wave & 1;
        const int wi = blockIdx.x >> 3, WX = gridDim.x >> 3;
        const int rem_units = (a.cum[a.n] - a.rounds * WX) * NT;   // k-tiles of the tiles behind the full rounds, shared out evenly
        const int rb = (int)(((long long)wi * rem_units) / WX), re = (int)(((long long)(wi + 1) * rem_units) / WX);
        const int nfull = a.rounds * NT, nv = nfull + (re - rb);
        if (nv <= 0) return;
        Ctx c{a, smem, tid, lane, wave, (int)(blockIdx.x & 7), wm * 32 + (lane & 31), wn * 32 + (lane & 31), lane >> 5,
              wave * 4 + (lane >> 4), (lane & 15) * 4, wi, WX, nfull, rb, nv, 0, nullptr, nullptr, 0, 0, 0,
              (int)(blockIdx.x / (gridDim.x >> 2))};
        issue_decode(c);
#pragma unroll
        for (int d = 0; d < D; ++d) if (c.iv < nv) issue_any(c);

#ifdef BP_SK_TRACE
        int tslot = 0;
#define SKT(code) do { if (tid == 0 && a.trace && tslot < 15) { a.trace[(size_t)blockIdx.x * 16 + tslot] = (wall_clock64() << 4) | (unsigned)(code); ++tslot; } } while (0)
#else
#define SKT(code) ((void)0)
#endif
        SKT(0);
#ifdef BP_SK_TRACE
        if (tid == 0 && a.trace) a.trace[(size_t)blockIdx.x * 16 + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
#endif
        int v = 0;
        while (v < nv) {
            int j, kt0;
            locate(c, v, j, kt0);
            c.pbase += NT / BP_SK_PRIO_STEP + 1;                       // (a whole tile advances the rotation by one level more than a full cycle)
            const int kt1 = (nv - v) < (NT - kt0) ? kt0 + (nv - v) : NT;
            int p, tile_m, tile_n;
            decode(a, c.x, j, p, tile_m, tile_n);
            const EpiArgs &e = a.e[p];
            const int n0 = tile_n * BN, mb = tile_m * BM + wm * 32, nb = n0 + wn * 32;
            const bool do_bias = tile_m == 0;
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            float bsum = 0.f;

            if (kt0 > 0) {
                // ---- a piece of a tile whose first k-tiles belong to a predecessor in this XCD's list: compute, hand the partial
                // sums over (the predecessor finishes the tile as the LAST thing of its run), go on
                EpiPre none;
                seg<false>(c, v, kt0, kt1, e, mb, nb, do_bias, bsum, acc, none);
                float *wsp = uniform_ptr(a.ws + (size_t)blockIdx.x * WS + wave * 1024);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(wsp + r * 64 + lane, acc[0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
                if (do_bias) __hip_atomic_store(wsp + (BM * BN - wave * 1024 + wave * 64) + lane, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains (this also lands the DMAs in flight: once per run)
                __builtin_amdgcn_s_barrier();
                if (tid == 0) __hip_atomic_store(a.flags + blockIdx.x, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.hot = 0;
                SKT(1);
                continue;
            }
            EpiPre pre;
            // (defined here as far as the compiler can tell: the loop body's prefetch is conditional, and registers that might be
            // read undefined are kept alive around the tile loop -- every tile would end with copies of them behind a vmcnt(0))
            asm volatile("" : "=v"(pre.p0), "=v"(pre.p1));
            if (kt1 == NT && !do_bias && v + NT + D <= nv) {
                mid<0>(c, v, e, mb, nb, acc, pre);
                v += NT;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
                finish(c, e, n0, mb, nb, false, 0.f, acc[0], pre);
                if constexpr (BP_SK_DELAY) { issue_any(c); c.hot = D - 1; }   // (the stores are now OLDER than this request: one window less)
                else c.hot = D;
                SKT(2);
                continue;
            }
            seg<true>(c, v, kt0, kt1, e, mb, nb, do_bias, bsum, acc, pre);
            if (kt1 < NT) {
                // ---- the START of a tile of the last round: the following workgroups of the list hold [kt1, NT) -- one of them, or
                // two when a whole run lies inside the tile -- and computed it FIRST; add their pieces in list order
                const int tile_end = (j - a.rounds * WX + 1) * NT;                 // in k-tiles of the last round
                for (int q = 1;; ++q) {
                    const int qb = (int)(((long long)(wi + q) * rem_units) / WX), qe = (int)(((long long)(wi + q + 1) * rem_units) / WX);
                    if (qb >= tile_end) break;
                    if (qe == qb) continue;                                        // (an empty run publishes nothing)
                    const unsigned pw = blockIdx.x + 8 * q;
                    if (tid == 0) {
                        const unsigned long long t0 = wall_clock64();
                        while (__hip_atomic_load(a.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                            __builtin_amdgcn_s_sleep(8);
                            if (wall_clock64() - t0 > a.budget) { __hip_atomic_store(a.err, 1u + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                    const float *wsq = uniform_ptr(a.ws + (size_t)pw * WS + wave * 1024);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[0][r] += __hip_atomic_load(wsq + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // L1-bypassing
                    if (do_bias) bsum += __hip_atomic_load(wsq + (BM * BN - wave * 1024 + wave * 64) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            finish(c, e, n0, mb, nb, do_bias, bsum, acc[0], pre);
            c.hot = D;
            SKT(kt1 < NT ? 4 : 3);
        }
#undef SKT
    }
};

template <int KTOT, bool STORE = false>
__global__ __launch_bounds__(256, 4) void bp_wgrad_sk(const SkArgs a)
{
    using K = WgradSk<KTOT, STORE>;
    __shared__ __attribute__((aligned(16))) float smem[K::SMEM];
    K::run(a, smem);
}
