// tools/bf16_update_persist_probe.h -- NOT part of the library.  Round-5 experiment on the bf16 update launch (configs[4]):
// a PERSISTENT form whose update waves hold two register images of W / delta, so that tile i+1's HBM reads are in flight
// through all of tile i.  Built into a development library (it was `#include`d at the end of bp_wgrad_dma_bf16.h and launched
// from bf_wgrads_dma under BP_BF16_UPD_PERSIST=<workgroups>), parity suite green (bit-identical results), MEASURED SLOWER:
//
//   configs[4] per-GPU shape, ms per step (tools/bench_bf16.py c5bf16, same box, alternating runs; profiles/r05_bf16_persist.txt)
//     shipped one-tile-per-workgroup launch (3 x 6 waves per CU)                  0.600 / 0.603 / 0.610 / 0.615
//     persistent, 2 update waves (168 VGPRs), 256 / 512 / 768 workgroups         0.807 / 0.803 / 0.798   (ONE workgroup per CU resident)
//     persistent, 4 update waves (92 VGPRs),  256 / 512 / 768 workgroups         0.737 / 0.629-0.632 / 0.651
//   rocprofv3: bp_wgrad_dma_bf16_persist<512,64,3,4> 332 us per launch against 301 us for bp_wgrad_dma_bf16_six<512,64,3>.
//
// Why it does not pay: the premise was that the shipped launch is bound by how long a workgroup keeps HBM requests in flight.
// It is not -- three staggered workgroups per CU already keep the memory system busy: 301 us for 1.63 GB is 3.96 us per tile
// and CU = 5.4 TB/s, against 3.3 us at the 6.5 TB/s copy rate; two persistent workgroups per CU deliver a tile every 4.35 us.
// What separates the launch from the copy rate is the operand side (128 KB of L2 -> LDS traffic per tile), not the W / delta
// latency.  Kept for the record; the library has ONE update kernel.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------------
// PERSISTENT six-wave form: the same tile body, but a workgroup walks the tile list (its index + k * gridDim, which keeps the
// XCD bits) and its two update waves hold TWO register images of W / delta: while tile i runs its k-loop they already have
// the loads of tile i+1 in flight.  Why: the one-tile-per-workgroup launch keeps a workgroup's 32 KB of HBM reads in flight only
// from its start until they land (~2.5 us of a ~12 us tile slot); the rest of the slot (k-loop, update, stores) that workgroup
// has nothing outstanding, and three staggered workgroups per CU average less than one tile's loads in flight -- the launch is
// bound by that concurrency, not by bandwidth (forcing 128 / 256 / 512 frames: 250 / 259 / 296 us; 250 us is the copy rate).
// Here every workgroup always has a whole tile's W / delta on its way: 2 workgroups per CU (168 VGPRs for the two images) x
// 32 KB.  Barriers: all six waves meet NT + 2 times per tile; the update waves use RAW s_barrier (a __syncthreads() would
// drain their prefetch with vmcnt(0)), the MFMA waves wait lgkmcnt(0) by hand where their LDS writes must have landed.  The
// gradient tile is out of the way of the next prologue, so the MFMA waves start the NEXT tile's operand DMA while
// the update waves still read this one (it sits in the last ring stage, which that DMA does not touch).  Results are bit-identical to the one-tile-per-workgroup form (same expressions,
// same k order).
template <int KTOT, int BKX = 32, int STX = 4, int NU = 4>
struct WgradDmaBf6P {
    using M = WgradDmaBf<KTOT, BKX, STX>;
    static constexpr int GLD = 64;                         // floats per row of the staged gradient tile [64][64] (16 KB; the 2-way
                                                           // conflict of the accumulator-layout ds_write_b32 is free, guide LDS section)
    // where the gradient tile lives: the LAST ring stage when it is large enough (64-frame k-tiles: 16 KB) -- the next tile's
    // prologue fills stages 0 .. ST-2 only, and stage ST-1 is refilled behind the next tile's first barrier, which the update
    // waves reach only after they have read this tile -- else its own region behind the ring
    static constexpr bool SG_IN_RING = M::STAGE * 2 >= 64 * GLD * 4;
    static constexpr int SG_OFF = SG_IN_RING ? (M::ST - 1) * M::STAGE : M::SMEM;                 // halfs
    static constexpr int SMEM_HALFS = SG_IN_RING ? M::SMEM : M::SMEM + 64 * GLD * 2;
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    // NU update waves share the tile's 64 rows: 64 / NU rows = NP pieces of 16 bytes per lane and array.  NU = 4 (eight waves per
    // workgroup): two register images are 4 x 16 VGPRs, the kernel stays under 128 VGPRs and two workgroups fit a CU whatever SIMDs
    // their waves land on (NU = 2: 168 VGPRs, and in the measurement 256 / 512 / 768 workgroups took the same time, i.e. ONE
    // workgroup per CU was resident: 0.80 ms per configs[4] step against 0.61).
    static constexpr int NP = 16 / NU, RPW = 64 / NU;
    static_assert(NU == 2 || NU == 4, "update waves");
    struct Tile { int p, m0, n0, tile_m; };

    // first real tile at or behind b on this workgroup's walk (padding slots of the grouped list are skipped); -1: none
    static __device__ __forceinline__ int seek(const BfWgradMulti &a, int b, int stride, Tile &t)
    {
        const int total = a.first_tile[a.n];
        for (; b < total; b += stride) {
            int p = 0;
            while (p + 1 < a.n && b >= a.first_tile[p + 1]) ++p;
            const int rb = b - a.first_tile[p], tm = a.p[p].tiles_m, tn = a.p[p].tiles_n;
            if (rb >= tm * tn) continue;
            int tile_m, tile_n;
            if ((tn & 7) == 0) { const int xcd = rb & 7, j = rb >> 3, per = tn >> 3; tile_n = xcd * per + j % per; tile_m = j / per; }
            else { tile_m = rb % tm; tile_n = rb / tm; }
            t.p = p; t.m0 = tile_m * 64; t.n0 = tile_n * 64; t.tile_m = tile_m;
            return b;
        }
        return -1;
    }
    // An update wave's 8 pieces of a tile: chunk c = lane + 64 i -> row 32u + (c >> 4), 4 floats at column 4 (c & 15); as BYTE
    // offsets from the (uniform) base of W / delta: off0 + i * (4 rows), the same for both arrays and half of it for the shadow
    // (32-bit offsets against scalar bases: the two register images of W / delta leave no room for 64-bit addresses per piece)
    static __device__ __forceinline__ unsigned piece0(const EpiArgs &e, const Tile &t, int u, int lane)
    {
        return 4u * ((unsigned)(t.m0 + RPW * u + (lane >> 4)) * (unsigned)e.ldc + (unsigned)(t.n0 + (lane & 15) * 4));
    }
    static __device__ __forceinline__ void fetch(const BfWgradProblem &g, const Tile &t, int u, int lane, float4 (&w4)[NP], float4 (&d4)[NP])
    {
        const EpiArgs &e = g.e;
        const char *bw = reinterpret_cast<const char *>(e.C), *bd = reinterpret_cast<const char *>(e.aux2);
        const unsigned off0 = piece0(e, t, u, lane), step = 16u * (unsigned)e.ldc;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const nt_f4 a = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(bw + (off0 + i * step)));
            const nt_f4 d = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(bd + (off0 + i * step)));
            w4[i] = make_float4(a[0], a[1], a[2], a[3]); d4[i] = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    static __device__ __forceinline__ void apply(const BfWgradProblem &g, const Tile &t, int u, int lane, const float *sg, const float4 (&w4)[NP], const float4 (&d4)[NP])
    {
        const EpiArgs &e = g.e;
        char *bw = reinterpret_cast<char *>(e.C), *bd = reinterpret_cast<char *>(e.aux2), *bs = reinterpret_cast<char *>(g.Wb);
        const unsigned off0 = piece0(e, t, u, lane), step = 16u * (unsigned)e.ldc;
        const float *sgp = sg + (RPW * u + (lane >> 4)) * GLD + (lane & 15) * 4;
        // (the scheduler must not hoist all eight gradient reads in front of the arithmetic: with both register images of W / delta
        // live there is no room for them -- one piece ahead is enough to cover the LDS latency)
        float4 gn = *reinterpret_cast<const float4 *>(sgp);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float4 g4 = gn;
            if (i + 1 < NP) gn = *reinterpret_cast<const float4 *>(sgp + 4 * (i + 1) * GLD);
            __builtin_amdgcn_sched_barrier(0);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, wv[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w}, dv[4] = {d4[i].x, d4[i].y, d4[i].z, d4[i].w};
            float dn[4], wn_[4]; bf16_t hb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dn[j] = e.mom * dv[j] - e.c1 * (gv[j] / e.ndiv + e.wc * wv[j]);      // kernUpdatedelta
                wn_[j] = dn[j] + 1.0f * wv[j];                                        // kernAccSum
                hb[j] = f2bf(wn_[j]);
            }
            const unsigned o = off0 + i * step;
            const nt_f4 dst = {dn[0], dn[1], dn[2], dn[3]}, wst = {wn_[0], wn_[1], wn_[2], wn_[3]};
            __builtin_nontemporal_store(dst, reinterpret_cast<nt_f4 *>(bd + o));
            __builtin_nontemporal_store(wst, reinterpret_cast<nt_f4 *>(bw + o));
            *reinterpret_cast<uint2 *>(bs + (o >> 1)) =                               // (ldwb == ldc: one shadow in the weights' own layout)
                make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the barriers of one tile as an update wave sees them: the k-loop's, "ring free", "gradient tile in LDS"
    static __device__ __forceinline__ void sit_through_tile()
    {
#pragma unroll 1
        for (int t = 0; t < M::NT + 2; ++t) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    static __device__ __forceinline__ void run(const BfWgradMulti &a, bf16_t *smem)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int stride = gridDim.x;
        float *sg = reinterpret_cast<float *>(smem + SG_OFF);
        Tile t0, t1;
        int b = seek(a, blockIdx.x, stride, t0);
        if (b < 0) return;
        if (wave >= 4) {
            const int u = wave - 4;
            float4 wA[NP], dA[NP];
            fetch(a.p[t0.p], t0, u, lane, wA, dA);
            for (;;) {
                // the NEXT tile's W / delta go into a second register image now and stay in flight through this tile's k-loop; they
                // move into the first image behind this tile's stores (vmcnt retires in order: that copy waits for loads that
                // are a whole tile old, never for the stores just issued)
                float4 wB[NP], dB[NP];
                const int bn = seek(a, b + stride, stride, t1);
                if (bn >= 0) fetch(a.p[t1.p], t1, u, lane, wB, dB);
                sit_through_tile();
                apply(a.p[t0.p], t0, u, lane, sg, wA, dA);
                if (bn < 0) return;
#pragma unroll
                for (int i = 0; i < NP; ++i) { wA[i] = wB[i]; dA[i] = dB[i]; }
                t0 = t1; b = bn;
            }
        }
        // ---- MFMA waves
        const int wm = wave >> 1, wn = wave & 1;
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int t = 0; t < M::D; ++t) M::issue_tile(a.p[t0.p], t0.m0, t0.n0, t * M::BK, smem, t, wave, lane);
        for (;;) {
            const BfWgradProblem &g = a.p[t0.p];
            const EpiArgs &e = g.e;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bool do_bias = t0.tile_m == 0;
            const int n_bias = t0.n0 + (tid >> 2);
            float bsum = 0.f;
            M::template iter<0>(g, t0.m0, t0.n0, smem, wave, lane, tid, ra, rb, kh, do_bias, bsum, acc);
            __builtin_amdgcn_s_barrier();                       // every MFMA wave is past its last fragment read (the MFMAs consumed them)
            // the ring is free: the NEXT tile's first k-tiles start their way now (stages 0 .. D-1; the gradient tile below goes to
            // the last stage, or behind the ring), under the gradient hand-over and the update waves' work
            b = seek(a, b + stride, stride, t1);
            if (b >= 0) {
#pragma unroll
                for (int t = 0; t < M::D; ++t) M::issue_tile(a.p[t1.p], t1.m0, t1.n0, t * M::BK, smem, t, wave, lane);
            }
            {   // gradient block -> LDS: lane -> column, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5) of the wave's 32x32 block
                const int nl = wn * 32 + (lane & 31), ml = wm * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) sg[(ml + (r & 3) + 8 * (r >> 2)) * GLD + nl] = acc[r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // the gradient tile is in LDS
            if (do_bias) {
                bsum += __shfl_xor(bsum, 1);
                bsum += __shfl_xor(bsum, 2);
                if ((tid & 3) == 0 && n_bias < e.n_limit) {
                    const float d = e.mom * e.bias_d[n_bias] - e.c1 * (bsum / e.ndiv + 0.0f * e.bias_w[n_bias]);
                    e.bias_d[n_bias] = d;
                    e.bias_w[n_bias] = d + 1.0f * e.bias_w[n_bias];
                }
            }
            if (b < 0) return;
            t0 = t1;
        }
    }
};

template <int KTOT, int BKX = 32, int STX = 4, int NU = 4>
__global__ __launch_bounds__(256 + 64 * NU, NU == 4 ? 4 : 3) void bp_wgrad_dma_bf16_persist(const BfWgradMulti a)
{
    __shared__ __attribute__((aligned(16))) bf16_t smem[WgradDmaBf6P<KTOT, BKX, STX, NU>::SMEM_HALFS];
    WgradDmaBf6P<KTOT, BKX, STX, NU>::run(a, smem);
}
