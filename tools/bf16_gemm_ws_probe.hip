// tools/bf16_gemm_ws_probe.hip -- development probe (round 6; not part of the library): WAVE-SPECIALISED forms of the bf16 GEMM of
// configs[4] (C[512][4096] = A[512][4096] . B^T, B as [n][k] = the dgrad form or as [k][n] = the forward's form), against the
// loop the library shipped in rounds 4-5 (`v2` below = bp_gemm_bf16<.,128,.,DMA>: four waves that each issue their share of the
// LDS-DMA pieces BETWEEN their own MFMAs).
//
// Why: one global_load_lds_dwordx4 (1 KiB) costs the ISSUING wave 60-185 cycles (MI355X_MICROARCH.md, constants table), and with
// one wave per SIMD nothing else can issue on that SIMD meanwhile: 6 pieces per wave and k-tile are 360-600 cycles next to 256
// cycles of MFMA -- the 832 cycles per k-tile that round 4 measured.  Here the workgroup has 4 CONSUMER waves (fragment reads +
// MFMAs, nothing else) and NPW PRODUCER waves (all 24 DMA pieces of a k-tile, the counted vmcnt wait); they meet at ONE raw
// s_barrier per k-tile.  The barrier of tile t certifies tile t+1 as landed too, so the consumers' fragment read-ahead runs across
// tile boundaries and no LDS latency is exposed behind a barrier.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/bf16_gemm_ws_probe tools/bf16_gemm_ws_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <stdint.h>
typedef uint16_t bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
template <int N> struct VmWait { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); } };
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef const __attribute__((address_space(1))) void *glb_ptr;
template <int N> __device__ __forceinline__ void lgkm_wait3(f32x4v &a, f32x4v &b, f32x4v &c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait4(f32x4v &a, f32x4v &b, v4s &c, v4s &d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); }

static __host__ __device__ inline float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

__global__ void ref_gemm(const bf16_t *A, const bf16_t *B, float *C, int M, int N, int K)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf2f(A[(size_t)m * K + k]) * bf2f(B[(size_t)n * K + k]);
    C[(size_t)m * N + n] = s;
}

// ------------------------------------------------------------------ baseline: the loop of rounds 4-5 (4 waves, everybody does everything)
__device__ __forceinline__ void pf_wave(const bf16_t *B, int ldb, int n0, int nt, int tile_m, int tiles_m, int lane, bool bkn, int pfd);
// PFD > 0: a FIFTH wave that only prefetches (pf_wave, below) and joins the per-tile barrier
template <int ST, bool BKN, int RA, int PFD = 0>
__global__ __launch_bounds__(PFD ? 320 : 256) void gemm_dma_v2(const bf16_t *A, const bf16_t *B, float *C, int lda, int ldb, int ldc, int K, int tiles_m, int tiles_n, int stag)
{
    constexpr int STAGE = 192 * 128, D = ST - 1;
    constexpr int NRD = BKN ? 4 : 3;
    __shared__ __attribute__((aligned(1024))) char smem[ST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    { const int b = blockIdx.x, xcd = b & 7, jj = b >> 3, per = tiles_n >> 3; tile_n = xcd * per + jj / tiles_m; tile_m = jj % tiles_m; }
    const int m0 = tile_m * 128, n0 = tile_n * 64;
    if (PFD && wave == 4) { pf_wave(B, ldb, n0, K / 64, tile_m, tiles_m, lane, BKN, PFD); return; }   // (not with krep)
    const bf16_t *src[6]; size_t step[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int r = 8 * (wave * 6 + i) + (lane >> 3);
        if (r < 128 || !BKN) { const int c = (lane & 7) ^ ((r >> 1) & 7); src[i] = (r < 128 ? A + (size_t)(m0 + r) * lda : B + (size_t)(n0 + r - 128) * ldb) + c * 8; step[i] = 64; }
        else { const int k = r - 128, c = (lane & 7) ^ (4 * ((k >> 1) & 1)); src[i] = B + (size_t)k * ldb + n0 + c * 8; step[i] = (size_t)64 * ldb; }
    }
    const int krep = stag >= 1000 ? stag / 1000 : 1; stag %= 1000;
    const int ntr = K / 64, nt = ntr * krep;
    const int rot = (tile_m * stag) % ntr;
    auto issue_piece = [&](int i, int t, int st) {
        int tt = (t < nt ? t : nt - 1) % ntr;
        tt += rot; if (tt >= ntr) tt -= ntr;
        __builtin_amdgcn_global_load_lds((glb_ptr)(src[i] + (size_t)tt * step[i]), (lds_ptr)(smem + st * STAGE + (wave * 6 + i) * 1024), 16, 0, 0);
    };
    const int ra = wm * 64 + (lane & 31), rb = 128 + wn * 32 + (lane & 31), h = lane >> 5;
    const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)smem;
    unsigned aq[4], bq[4];
    const int j = (lane & 15) >> 2, cb = 4 * wn + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        aq[q] = lds0 + ra * 128 + 16 * (h ^ (sa & 1)) + 32 * (q ^ (sa >> 1));
        bq[q] = BKN ? lds0 + 128 * 128 + (16 * q + 8 * h + j) * 128 + 16 * (cb ^ (4 * ((j >> 1) & 1))) + 8 * (lane & 1)
                    : lds0 + rb * 128 + 16 * (h ^ (sb & 1)) + 32 * (q ^ (sb >> 1));
    }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < D; ++t)
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i, t, t);
    f32x4v a0[4], a1[4], bb[4]; v4s blo[4], bhi[4];
    auto reads = [&](unsigned so, int q) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(a0[q]) : "v"(aq[q] + so));
        if constexpr (BKN) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[q]) : "v"(bq[q] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(bhi[q]) : "v"(bq[q] + so));
        } else asm volatile("ds_read_b128 %0, %1" : "=v"(bb[q]) : "v"(bq[q] + so));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a1[q]) : "v"(aq[q] + so));
    };
#define ARRIVED(q, N) do { if constexpr (BKN) lgkm_wait4<(N)>(a0[q], a1[q], blo[q], bhi[q]); else lgkm_wait3<(N)>(a0[q], a1[q], bb[q]); } while (0)
    auto bfrag = [&](int q) {
        if constexpr (BKN) return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[q], bhi[q], 0, 1, 2, 3, 4, 5, 6, 7));
        else return __builtin_bit_cast(bf16x8_t, bb[q]);
    };
    for (int t = 0; t < nt; ++t) {
        const unsigned so = (unsigned)((t % ST) * STAGE);
        const int stn = (t + D) % ST;
        VmWait<(D - 1) * 6>::go();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < RA; ++q) reads(so, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + RA - 1 <= 3) ARRIVED(q, (RA - 1) * NRD);
            else if (q == 3) ARRIVED(q, 0);
            else ARRIVED(q, NRD);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8_t bf = bfrag(q);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0[q]), bf, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 2) issue_piece(2 * q, t + D, stn); else issue_piece(2 + q, t + D, stn);
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1[q]), bf, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 2) issue_piece(2 * q + 1, t + D, stn);
            if (q + RA < 4) reads(so, q + RA);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            C[(size_t)m * ldc + n] = acc[i][r];
        }
}

// ------------------------------------------------------------------ wave-specialised: 4 consumer waves + NPW producer waves
// Ring of ST stages of one 192-row x 128-byte k-tile image (same image and source swizzles as v2).  Tile u lives in stage u % ST.
//   barrier #t (t = 0 .. nt-1) certifies: tiles <= t+1 have landed; every consumer is past its last read of tile t-1.
//   producer, iteration t : s_waitcnt vmcnt((ST-3)*PP)  [own pieces of tile t+1 landed]  ->  s_barrier #t  ->  issue its PP pieces of tile
//                           t+ST-1 into the stage tile t-1 just left (ST-2 tiles in flight behind the two landed ones)
//   consumer              : s_barrier #t, then the four k-steps of tile t; the fragments of step g+2 (global step index, crossing into
//                           tile t+1 at q = 2, 3) are issued behind the MFMAs of step g.  RA = 2 fragment sets in flight.
// ABL: 0 = full kernel | 1 = producers issue no DMA in the loop (wrong result) | 2 = consumers read no fragments (wrong result)
// PRIO: s_setprio value of the consumer waves (0 = leave alone)
// PFD > 0: one more wave, the PREFETCHER: behind barrier #t it touches the 64 lines of B's k-tile t+PFD (one 4-byte load per lane and
// 128-byte line, result never used; its own vmcnt) if that tile is this m-tile's turn (t+PFD mod tiles_m): the m-tiles that share a
// B panel sit on one XCD, so each line is pulled into that L2 ONCE, PFD tiles before the ring's DMA asks for it.  Why: the ring
// keeps 3 tiles = 24 KB of B per CU in flight and the sharers of a panel ask for the SAME lines, i.e. 1.5 MB of unique weight bytes
// in flight chip-wide -- at ~1 us of HBM latency that is the 1.2-1.5 TB/s the cold launches measure.
__device__ __forceinline__ void pf_wave(const bf16_t *B, int ldb, int n0, int nt, int tile_m, int tiles_m, int lane, bool bkn, int pfd)
{
    const bf16_t *line = bkn ? B + (size_t)lane * ldb + n0 : B + (size_t)(n0 + lane) * ldb;    // line `lane` of k-tile 0
    const size_t step = bkn ? (size_t)64 * ldb : 64;
    // The load's destination register is written when the data RETURNS: it must stay allocated to `junk` until the final wait
    // ("+v" chains every touch through the same register; a plain "=v" output is dead at once and the register gets reused).
    unsigned junk = 0;
    auto touch = [&](int t) {
        if (t < nt && t % tiles_m == tile_m)
            asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(line + (size_t)t * step) : "memory");
    };
    for (int t = 3; t < pfd; ++t) touch(t);                     // (tiles 0..2 are the producers' prologue)
    for (int t = 0; t < nt; ++t) {
        __builtin_amdgcn_s_barrier();
        touch(t + pfd);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(junk) :: "memory");
}
template <int ST, bool BKN, int NPW, int PRIO, int ABL = 0, int PFD = 0>
__global__ __launch_bounds__(256 + 64 * NPW + (PFD ? 64 : 0), 1) void gemm_ws(const bf16_t *A, const bf16_t *B, float *C, int lda, int ldb, int ldc, int K, int tiles_m, int tiles_n, int stag)
{
    static_assert(24 % NPW == 0 && ST >= 4, "producer waves must divide the 24 pieces of a k-tile; ring of at least 4");
    constexpr int STAGE = 192 * 128, PP = 24 / NPW;
    constexpr int NRD = BKN ? 4 : 3;
    __shared__ __attribute__((aligned(1024))) char smem[ST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n;
    { const int b = blockIdx.x, xcd = b & 7, jj = b >> 3, per = tiles_n >> 3; tile_n = xcd * per + jj / tiles_m; tile_m = jj % tiles_m; }
    const int m0 = tile_m * 128, n0 = tile_n * 64;
    const int nt = K / 64;
    if (PFD && wave == 4 + NPW) { pf_wave(B, ldb, n0, nt, tile_m, tiles_m, lane, BKN, PFD); return; }
    if (wave >= 4) {
        // ---------------- producer
        const int pw = wave - 4;
        const bf16_t *src[PP]; size_t step[PP];
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int r = 8 * (pw * PP + i) + (lane >> 3);
            if (r < 128 || !BKN) { const int c = (lane & 7) ^ ((r >> 1) & 7); src[i] = (r < 128 ? A + (size_t)(m0 + r) * lda : B + (size_t)(n0 + r - 128) * ldb) + c * 8; step[i] = 64; }
            else { const int k = r - 128, c = (lane & 7) ^ (4 * ((k >> 1) & 1)); src[i] = B + (size_t)k * ldb + n0 + c * 8; step[i] = (size_t)64 * ldb; }
        }
        const int rot = (tile_m * stag) % nt;
        auto issue_tile = [&](int t, int st) {
            int tt = t < nt ? t : nt - 1;                      // past the end: a duplicate into a stage nobody reads again (keeps every count static)
            tt += rot; if (tt >= nt) tt -= nt;
#pragma unroll
            for (int i = 0; i < PP; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src[i] + (size_t)tt * step[i]), (lds_ptr)(smem + st * STAGE + (pw * PP + i) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < ST - 1; ++t) issue_tile(t, t);
        int stn = ST - 1;                                      // stage of tile t + ST - 1
        for (int t = 0; t < nt; ++t) {
            VmWait<(ST - 3) * PP>::go();
            __builtin_amdgcn_s_barrier();
            if (ABL != 1) issue_tile(t + ST - 1, stn);
            stn = stn + 1 == ST ? 0 : stn + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ---------------- consumer
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    const int wm = wave >> 1, wn = wave & 1;
    const int ra = wm * 64 + (lane & 31), rb = 128 + wn * 32 + (lane & 31), h = lane >> 5;
    const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)smem;
    unsigned aq[4], bq[4];
    const int j = (lane & 15) >> 2, cb = 4 * wn + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        aq[q] = lds0 + ra * 128 + 16 * (h ^ (sa & 1)) + 32 * (q ^ (sa >> 1));
        bq[q] = BKN ? lds0 + 128 * 128 + (16 * q + 8 * h + j) * 128 + 16 * (cb ^ (4 * ((j >> 1) & 1))) + 8 * (lane & 1)
                    : lds0 + rb * 128 + 16 * (h ^ (sb & 1)) + 32 * (q ^ (sb >> 1));
    }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    f32x4v a0[4], a1[4], bb[4]; v4s blo[4], bhi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a0[q] = 0.f; a1[q] = 0.f; bb[q] = 0.f; blo[q] = 0; bhi[q] = 0; }
    auto reads = [&](unsigned so, int q) {
        if (ABL == 2) return;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a0[q]) : "v"(aq[q] + so));
        if constexpr (BKN) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[q]) : "v"(bq[q] + so));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(bhi[q]) : "v"(bq[q] + so));
        } else asm volatile("ds_read_b128 %0, %1" : "=v"(bb[q]) : "v"(bq[q] + so));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a1[q]) : "v"(aq[q] + so));
    };
    auto bfrag = [&](int q) {
        if constexpr (BKN) return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(blo[q], bhi[q], 0, 1, 2, 3, 4, 5, 6, 7));
        else return __builtin_bit_cast(bf16x8_t, bb[q]);
    };
    __builtin_amdgcn_s_barrier();                               // #0: tiles 0 and 1 have landed
    __builtin_amdgcn_sched_barrier(0);
    reads(0u, 0); reads(0u, 1);
    __builtin_amdgcn_sched_barrier(0);
    unsigned so = 0u;                                           // stage offset of tile t
    for (int t = 0; t < nt; ++t) {
        const unsigned son = so + STAGE == ST * STAGE ? 0u : so + STAGE;   // ... of tile t+1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ABL != 2) ARRIVED(q, NRD);                      // in flight behind step q: the fragments of the next step
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8_t bf = bfrag(q);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0[q]), bf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1[q]), bf, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 2) reads(so, q + 2); else reads(son, q - 2);   // (behind the last tile: a stage that exists, contents never used)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < nt) __builtin_amdgcn_s_barrier();           // #(t+1)
        __builtin_amdgcn_sched_barrier(0);
        so = son;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            C[(size_t)m * ldc + n] = acc[i][r];
        }
}

// ------------------------------------------------------------------ wave-specialised, consumers split k INSIDE the k-tile
// Consumer wave w takes k-step w (16 of the tile's 64 k) for the WHOLE 128 x 64 tile: 8 accumulator blocks (128 registers), per k-tile
// 4 A fragments + 2 B fragments for 8 MFMAs -- half the fragment bytes out of LDS that the 2 x 2 arrangement reads (each fragment
// feeds 2 resp. 4 MFMAs instead of 1 resp. 2), eight independent accumulator chains.  The four partial tiles meet once, after
// the k-loop, through the (then free) ring: wave w hands the six blocks it does not own to their owners and finishes blocks
// 2w, 2w+1 -- every wave runs a quarter of the epilogue.  Fragment registers are double buffered by name: the reads of tile t+1
// (certified by barrier #t) are issued before the MFMAs of tile t.
template <int ST, bool BKN, int NPW, int PFD = 0, int ABL = 0>
__global__ __launch_bounds__(256 + 64 * NPW, 1) void gemm_ws2(const bf16_t *A, const bf16_t *B, float *C, int lda, int ldb, int ldc, int K, int tiles_m, int tiles_n, int stag)
{
    static_assert(24 % NPW == 0 && ST >= 4, "producer waves must divide the 24 pieces of a k-tile; ring of at least 4");
    constexpr int STAGE = 192 * 128, PP = 24 / NPW;
    __shared__ __attribute__((aligned(1024))) char smem[ST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n;
    { const int b = blockIdx.x, xcd = b & 7, jj = b >> 3, per = tiles_n >> 3; tile_n = xcd * per + jj / tiles_m; tile_m = jj % tiles_m; }
    const int m0 = tile_m * 128, n0 = tile_n * 64;
    // timing aid: stag = 1000*krep + stag walks the k range krep times (wrong result; slope per k-tile = (T(2) - T(1)) / (K/64))
    const int krep = stag >= 1000 ? stag / 1000 : 1; stag %= 1000;
    const int ntr = K / 64, nt = ntr * krep;
    if (wave >= 4) {
        const int pw = wave - 4;
        const bf16_t *src[PP]; size_t step[PP];
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int r = 8 * (pw * PP + i) + (lane >> 3);
            if (r < 128 || !BKN) { const int c = (lane & 7) ^ ((r >> 1) & 7); src[i] = (r < 128 ? A + (size_t)(m0 + r) * lda : B + (size_t)(n0 + r - 128) * ldb) + c * 8; step[i] = 64; }
            else { const int k = r - 128, c = (lane & 7) ^ (4 * ((k >> 1) & 1)); src[i] = B + (size_t)k * ldb + n0 + c * 8; step[i] = (size_t)64 * ldb; }
        }
        const int rot = (tile_m * stag) % ntr;
        auto issue_tile = [&](int t, int st) {
            int tt = (t < nt ? t : nt - 1) % ntr;
            tt += rot; if (tt >= ntr) tt -= ntr;
#pragma unroll
            for (int i = 0; i < PP; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src[i] + (size_t)tt * step[i]), (lds_ptr)(smem + st * STAGE + (pw * PP + i) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < ST - 1; ++t) issue_tile(t, t);
        int stn = ST - 1;
        for (int t = 0; t < nt; ++t) {
            VmWait<(ST - 3) * PP>::go();
            if (ABL != 3 && ABL != 5) __builtin_amdgcn_s_barrier();
            if (ABL == 0 || ABL == 2 || ABL == 6) issue_tile(t + ST - 1, stn);
            stn = stn + 1 == ST ? 0 : stn + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ---------------- consumer `wave`: k-step `wave` of every k-tile
    const int h = lane >> 5, rl = lane & 31, sa = (rl >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)smem;
    // A fragment of block i (rows 32 i + rl): + 4096 i bytes.  Chunk 2 wave + h of the row, in slot chunk ^ sa (source swizzle of the DMA).
    const unsigned aqa = lds0 + rl * 128 + 16 * (((2 * wave + h) ^ sa) & 7);
    unsigned bqa[2];
    if constexpr (BKN) {
        const int j4 = (lane & 15) >> 2;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int cb = 4 * jb + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
            bqa[jb] = lds0 + 128 * 128 + (16 * wave + 8 * h + j4) * 128 + 16 * (cb ^ (4 * ((j4 >> 1) & 1))) + 8 * (lane & 1);
        }
    } else {
        bqa[0] = lds0 + (128 + rl) * 128 + 16 * (((2 * wave + h) ^ sa) & 7);     // rows 128 + 32 jb + rl: the same swizzle (32 jb >> 1 is a multiple of 8)
        bqa[1] = bqa[0] + 4096;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jb][r] = 0.f;
    struct Frag { f32x4v a[4]; f32x4v b[2]; v4s blo[2], bhi[2]; };
    Frag f0, f1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f0.a[i] = 0.f; f1.a[i] = 0.f; }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) { f0.b[jb] = 0.f; f1.b[jb] = 0.f; f0.blo[jb] = 0; f0.bhi[jb] = 0; f1.blo[jb] = 0; f1.bhi[jb] = 0; }
    constexpr int NRD = BKN ? 8 : 6;
#define WS2_READS(F, so)                                                                                             \
    do { if (ABL != 2 && ABL != 3 && ABL != 4) {                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(F.a[0]) : "v"(aqa + (so)));                                        \
        if constexpr (BKN) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(F.blo[0]) : "v"(bqa[0] + (so)));        \
                             asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(F.bhi[0]) : "v"(bqa[0] + (so))); } \
        else asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[0]) : "v"(bqa[0] + (so)));                                \
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(F.a[1]) : "v"(aqa + (so)));                            \
        if constexpr (BKN) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(F.blo[1]) : "v"(bqa[1] + (so)));        \
                             asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(F.bhi[1]) : "v"(bqa[1] + (so))); } \
        else asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[1]) : "v"(bqa[1] + (so)));                                \
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(F.a[2]) : "v"(aqa + (so)));                            \
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(F.a[3]) : "v"(aqa + (so)));                           \
    } } while (0)
// wait until at most N LDS reads are outstanding; everything of F issued before them has then arrived
#define WS2_ARRIVED(F, N)                                                                                            \
    do { if (ABL != 2 && ABL != 3 && ABL != 4) { if constexpr (BKN) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.blo[0]), "+v"(F.bhi[0]), "+v"(F.blo[1]), "+v"(F.bhi[1]) : "n"(N)); \
                         else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.b[0]), "+v"(F.b[1]) : "n"(N)); } } while (0)
#define WS2_MFMAS(F)                                                                                                 \
    do { bf16x8_t b0_, b1_;                                                                                          \
         if constexpr (BKN) { b0_ = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(F.blo[0], F.bhi[0], 0, 1, 2, 3, 4, 5, 6, 7)); \
                              b1_ = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(F.blo[1], F.bhi[1], 0, 1, 2, 3, 4, 5, 6, 7)); } \
         else { b0_ = __builtin_bit_cast(bf16x8_t, F.b[0]); b1_ = __builtin_bit_cast(bf16x8_t, F.b[1]); }            \
         _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
             acc[i_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, F.a[i_]), b0_, acc[i_][0], 0, 0, 0); \
             acc[i_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, F.a[i_]), b1_, acc[i_][1], 0, 0, 0); } } while (0)
    // PFD: consumer wave 0 is also the prefetcher (pf_wave above, folded in: the consumers never wait on vmcnt, so these loads cost
    // one issue slot every tiles_m-th tile and nothing else; a ninth wave would put three waves on one SIMD = 168 registers)
    const bf16_t *pf_line = BKN ? B + (size_t)lane * ldb + n0 : B + (size_t)(n0 + lane) * ldb;
    const size_t pf_step = BKN ? (size_t)64 * ldb : 64;
    unsigned junk = 0;                                          // (kept live until the final vmcnt(0): see pf_wave)
    auto touch = [&](int t) {
        if (PFD && wave == 0 && t < nt && t % tiles_m == tile_m)
            asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(pf_line + (size_t)t * pf_step) : "memory");
    };
    for (int t = 3; t < PFD; ++t) touch(t);
// ABL 6 / 7 (full kernel / no DMA): the reads of the NEXT tile one by one behind the MFMAs of this one (a ds_read issued in the shadow
// of a 32-cycle MFMA costs nothing; six of them in a row idle the matrix pipe: +53 ns per k-tile, slope measurement of this probe)
#define WS2_RD1(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define WS2_TR1(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define WS2_TILE_IL(FC, FN, son)                                                                                     \
    do { WS2_ARRIVED(FC, 0);                                                                                         \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         bf16x8_t b0_, b1_;                                                                                          \
         if constexpr (BKN) { b0_ = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(FC.blo[0], FC.bhi[0], 0, 1, 2, 3, 4, 5, 6, 7)); \
                              b1_ = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(FC.blo[1], FC.bhi[1], 0, 1, 2, 3, 4, 5, 6, 7)); } \
         else { b0_ = __builtin_bit_cast(bf16x8_t, FC.b[0]); b1_ = __builtin_bit_cast(bf16x8_t, FC.b[1]); }          \
         acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[0]), b0_, acc[0][0], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         WS2_RD1(FN.a[0], aqa + (son), 0);                                                                           \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[0]), b1_, acc[0][1], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         if constexpr (BKN) { WS2_TR1(FN.blo[0], bqa[0] + (son), 0); WS2_TR1(FN.bhi[0], bqa[0] + (son), 512); } else WS2_RD1(FN.b[0], bqa[0] + (son), 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[1]), b0_, acc[1][0], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         WS2_RD1(FN.a[1], aqa + (son), 4096);                                                                        \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[1]), b1_, acc[1][1], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         if constexpr (BKN) { WS2_TR1(FN.blo[1], bqa[1] + (son), 0); WS2_TR1(FN.bhi[1], bqa[1] + (son), 512); } else WS2_RD1(FN.b[1], bqa[1] + (son), 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[2]), b0_, acc[2][0], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         WS2_RD1(FN.a[2], aqa + (son), 8192);                                                                        \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[2]), b1_, acc[2][1], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         WS2_RD1(FN.a[3], aqa + (son), 12288);                                                                       \
         __builtin_amdgcn_sched_barrier(0);                                                                          \
         acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[3]), b0_, acc[3][0], 0, 0, 0); \
         acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, FC.a[3]), b1_, acc[3][1], 0, 0, 0); \
         __builtin_amdgcn_sched_barrier(0); } while (0)
    __builtin_amdgcn_s_barrier();                               // #0: tiles 0 and 1 have landed
    __builtin_amdgcn_sched_barrier(0);
    WS2_READS(f0, 0u);
    unsigned so = 0u;
    if constexpr (ABL == 6 || ABL == 7) {
        for (int t = 0; t < nt; t += 2) {
            const unsigned s1 = so + STAGE == ST * STAGE ? 0u : so + STAGE, s2 = s1 + STAGE == ST * STAGE ? 0u : s1 + STAGE;
            touch(t + PFD);
            WS2_TILE_IL(f0, f1, s1);                            // tile t; fetches the fragments of tile t+1 (certified by barrier #t)
            __builtin_amdgcn_s_barrier();                       // #(t+1)
            __builtin_amdgcn_sched_barrier(0);
            touch(t + 1 + PFD);
            WS2_TILE_IL(f1, f0, s2);
            if (t + 2 < nt) __builtin_amdgcn_s_barrier();       // #(t+2)
            __builtin_amdgcn_sched_barrier(0);
            so = s2;
        }
    } else
    // two tiles per trip (fragment sets by name); nt is even (K a multiple of 128)
    for (int t = 0; t < nt; t += 2) {
        const unsigned s1 = so + STAGE == ST * STAGE ? 0u : so + STAGE, s2 = s1 + STAGE == ST * STAGE ? 0u : s1 + STAGE;
        __builtin_amdgcn_sched_barrier(0);
        WS2_READS(f1, s1);                                      // tile t+1: certified by barrier #t
        touch(t + PFD);
        WS2_ARRIVED(f0, NRD);
        __builtin_amdgcn_sched_barrier(0);
        WS2_MFMAS(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 3 && ABL != 5) __builtin_amdgcn_s_barrier();   // #(t+1)
        __builtin_amdgcn_sched_barrier(0);
        WS2_READS(f0, s2);                                      // tile t+2 (behind the last tile: a stage that exists, contents unused)
        touch(t + 1 + PFD);
        WS2_ARRIVED(f1, NRD);
        __builtin_amdgcn_sched_barrier(0);
        WS2_MFMAS(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt && ABL != 3 && ABL != 5) __builtin_amdgcn_s_barrier();           // #(t+2)
        so = s2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PFD) asm volatile("s_waitcnt vmcnt(0)" : "+v"(junk) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- the four partial tiles meet: block b = 2 i + jb is finished by wave b >> 1.  Exchange area (the ring; the producers are
    // done: their last pieces are duplicates nobody reads, and they drained vmcnt before leaving): [owner][giver slot 0..2][2 blocks][16 regs][64 lanes]
    __builtin_amdgcn_s_barrier();                               // every consumer is past its last fragment read
    float *xa = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i == wave) continue;
        const int slot = wave < i ? wave : wave - 1;            // this giver's slot at owner i
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4v v; v[0] = acc[i][jb][4 * r4]; v[1] = acc[i][jb][4 * r4 + 1]; v[2] = acc[i][jb][4 * r4 + 2]; v[3] = acc[i][jb][4 * r4 + 3];
                *reinterpret_cast<f32x4v *>(xa + ((((i * 3 + slot) * 2 + jb) * 4 + r4) * 64 + lane) * 4) = v;
            }
    }
    __syncthreads();
    // (acc[wave][.] indexed by a wave-uniform RUNTIME value would go through scratch: select by name)
#define WS2_FINISH(W)                                                                                                \
    if (wave == (W)) {                                                                                               \
        _Pragma("unroll") for (int jb = 0; jb < 2; ++jb) {                                                           \
            _Pragma("unroll") for (int sl = 0; sl < 3; ++sl)                                                         \
                _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4) {                                                   \
                    const f32x4v v = *reinterpret_cast<const f32x4v *>(xa + (((((W) * 3 + sl) * 2 + jb) * 4 + r4) * 64 + lane) * 4); \
                    acc[(W)][jb][4 * r4] += v[0]; acc[(W)][jb][4 * r4 + 1] += v[1]; acc[(W)][jb][4 * r4 + 2] += v[2]; acc[(W)][jb][4 * r4 + 3] += v[3]; \
                }                                                                                                    \
            const int n = n0 + jb * 32 + (lane & 31);                                                                \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                         \
                const int m = m0 + (W) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);                              \
                C[(size_t)m * ldc + n] = acc[(W)][jb][r];                                                            \
            }                                                                                                        \
        }                                                                                                            \
    }
    WS2_FINISH(0) WS2_FINISH(1) WS2_FINISH(2) WS2_FINISH(3)
#undef WS2_FINISH
#undef WS2_READS
#undef WS2_ARRIVED
#undef WS2_MFMAS
}

int main()
{
    const int M = 512, N = 4096, K = 4096;
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto &v : hA) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto &v : hB) v = f2bf(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f);
    bf16_t *A, *B; float *C, *R;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_gemm, dim3(N / 256, M), dim3(256), 0, 0, A, B, R, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> hR((size_t)M * N), hC((size_t)M * N);
    CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost));
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<bf16_t> hBt((size_t)K * N);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) hBt[(size_t)k * N + n] = hB[(size_t)n * K + k];
    bf16_t *Bt; CK(hipMalloc(&Bt, hBt.size() * 2)); CK(hipMemcpy(Bt, hBt.data(), hBt.size() * 2, hipMemcpyHostToDevice));
    // cold weights: 8 copies used round-robin (8 x 33.5 MB > the 256 MB Infinity Cache): every launch streams B from HBM
    bf16_t *Bs[8], *Bts[8];
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&Bs[i], hB.size() * 2)); CK(hipMemcpy(Bs[i], B, hB.size() * 2, hipMemcpyDeviceToDevice));
                                  CK(hipMalloc(&Bts[i], hBt.size() * 2)); CK(hipMemcpy(Bts[i], Bt, hBt.size() * 2, hipMemcpyDeviceToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int tm = M / 128, tn = N / 64;
    int rr = 0;
    auto run = [&](const char *name, auto launch) {
        CK(hipMemsetAsync(C, 0, (size_t)M * N * 4, st));
        launch();
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < hC.size(); ++i) { md = std::max(md, (double)fabsf(hC[i] - hR[i])); mx = std::max(mx, (double)fabsf(hR[i])); }
        std::vector<float> ts;
        for (int r = 0; r < 7; ++r) {
            launch(); launch();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 20 * 1000.f);
        }
        std::sort(ts.begin(), ts.end());
        const double fl = 2.0 * M * N * K;
        printf("%-64s med %7.2f us  %6.1f TF (%.3f of 2.5 PF)   max|diff| %.2e (max|ref| %.2e)\n", name, ts[3], fl / ts[3] * 1e-6, fl / ts[3] * 1e-6 / 2500.0, md, mx);
        fflush(stdout);
    };
#define WS(ST, BKN, NPW, PRIO, ABL, Bp, ldb_, stag) hipLaunchKernelGGL((gemm_ws<ST, BKN, NPW, PRIO, ABL>), dim3(tm * tn), dim3(256 + 64 * NPW), 0, st, A, Bp, C, K, ldb_, N, K, tm, tn, stag)
#define WS2(ST, BKN, NPW, PFD, ABL, Bp, ldb_, stag) hipLaunchKernelGGL((gemm_ws2<ST, BKN, NPW, PFD, ABL>), dim3(tm * tn), dim3(256 + 64 * NPW), 0, st, A, Bp, C, K, ldb_, N, K, tm, tn, stag)
#define WSP(ST, BKN, NPW, PFD, Bp, ldb_, stag) hipLaunchKernelGGL((gemm_ws<ST, BKN, NPW, 0, 0, PFD>), dim3(tm * tn), dim3(256 + 64 * NPW + (PFD ? 64 : 0)), 0, st, A, Bp, C, K, ldb_, N, K, tm, tn, stag)
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d: correctness + absolute times of the interleaved form\n", pass);
        run("v2 (rounds 4-5) [n][k] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, B, C, K, K, N, K, tm, tn, 0); });
        run("v2 (rounds 4-5) [k][n] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bt, C, K, N, N, K, tm, tn, 0); });
        run("ws2 interleaved [n][k] warm", [&] { WS2(5, false, 4, 0, 6, B, K, 0); });
        run("ws2 interleaved [k][n] warm", [&] { WS2(5, true, 4, 0, 6, Bt, N, 0); });
        run("ws2 interleaved, prefetch 16 [n][k] warm", [&] { WS2(5, false, 4, 16, 6, B, K, 0); });
        run("v2 (rounds 4-5) [n][k] COLD, sharers 2 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 2); });
        run("v2 (rounds 4-5) [k][n] COLD, sharers 4 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bts[rr++ & 7], C, K, N, N, K, tm, tn, 4); });
        run("v2 + prefetch wave 16 [n][k] COLD", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2, 16>), dim3(tm * tn), dim3(320), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 0); });
        run("ws2 interleaved, no prefetch [n][k] COLD", [&] { WS2(5, false, 4, 0, 6, Bs[rr++ & 7], K, 0); });
        run("ws2 interleaved, prefetch 8 [n][k] COLD", [&] { WS2(5, false, 4, 8, 6, Bs[rr++ & 7], K, 0); });
        run("ws2 interleaved, prefetch 16 [n][k] COLD", [&] { WS2(5, false, 4, 16, 6, Bs[rr++ & 7], K, 0); });
        run("ws2 interleaved, prefetch 16 [k][n] COLD", [&] { WS2(5, true, 4, 16, 6, Bts[rr++ & 7], N, 0); });
        run("ws2 interleaved, no prefetch [k][n] COLD", [&] { WS2(5, true, 4, 0, 6, Bts[rr++ & 7], N, 0); });
    }
    if (getenv("WS_ROUND2"))
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d (round 2 of the probe: prefetch wave, k-split consumers)\n", pass);
        run("v2 (rounds 4-5) [n][k] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, B, C, K, K, N, K, tm, tn, 0); });
        run("v2 (rounds 4-5) [k][n] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bt, C, K, N, N, K, tm, tn, 0); });
        run("ws2 (k-split consumers) ring 5, 4 producers [n][k] warm", [&] { WS2(5, false, 4, 0, 0, B, K, 0); });
        run("ws2 ring 6, 4 producers [n][k] warm", [&] { WS2(6, false, 4, 0, 0, B, K, 0); });
        run("ws2 ring 5, 4 producers [k][n] warm", [&] { WS2(5, true, 4, 0, 0, Bt, N, 0); });
        run("  ws2 ring 5, 4 producers: no DMA in the loop (wrong result)", [&] { WS2(5, false, 4, 0, 1, B, K, 0); });
        run("  ws2 ring 5, 4 producers: no fragment reads (wrong result)", [&] { WS2(5, false, 4, 0, 2, B, K, 0); });
        run("ws2 ring 5, 4 producers, prefetch 16 [n][k] warm", [&] { WS2(5, false, 4, 16, 0, B, K, 0); });
        run("v2 (rounds 4-5) [n][k] COLD, sharers 2 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 2); });
        run("v2 (rounds 4-5) [k][n] COLD, sharers 4 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bts[rr++ & 7], C, K, N, N, K, tm, tn, 4); });
        run("v2 + prefetch wave 8 [n][k] COLD", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2, 8>), dim3(tm * tn), dim3(320), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 0); });
        run("v2 + prefetch wave 16 [n][k] COLD", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2, 16>), dim3(tm * tn), dim3(320), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 0); });
        run("v2 + prefetch wave 24 [n][k] COLD", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2, 24>), dim3(tm * tn), dim3(320), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 0); });
        run("v2 + prefetch wave 16 [k][n] COLD", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2, 16>), dim3(tm * tn), dim3(320), 0, st, A, Bts[rr++ & 7], C, K, N, N, K, tm, tn, 0); });
        run("v2 + prefetch wave 16 [n][k] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2, 16>), dim3(tm * tn), dim3(320), 0, st, A, B, C, K, K, N, K, tm, tn, 0); });
        run("ws ring 5, 4 producers, no prefetch [n][k] COLD", [&] { WSP(5, false, 4, 0, Bs[rr++ & 7], K, 0); });
        run("ws ring 5, 4 producers, prefetch 8 [n][k] COLD", [&] { WSP(5, false, 4, 8, Bs[rr++ & 7], K, 0); });
        run("ws ring 5, 4 producers, prefetch 16 [n][k] COLD", [&] { WSP(5, false, 4, 16, Bs[rr++ & 7], K, 0); });
        run("ws ring 5, 4 producers, prefetch 24 [n][k] COLD", [&] { WSP(5, false, 4, 24, Bs[rr++ & 7], K, 0); });
        run("ws ring 5, 4 producers, prefetch 16 [k][n] COLD", [&] { WSP(5, true, 4, 16, Bts[rr++ & 7], N, 0); });
        run("ws2 ring 5, 4 producers, no prefetch [n][k] COLD", [&] { WS2(5, false, 4, 0, 0, Bs[rr++ & 7], K, 0); });
        run("ws2 ring 5, 4 producers, prefetch 8 [n][k] COLD", [&] { WS2(5, false, 4, 8, 0, Bs[rr++ & 7], K, 0); });
        run("ws2 ring 5, 4 producers, prefetch 16 [n][k] COLD", [&] { WS2(5, false, 4, 16, 0, Bs[rr++ & 7], K, 0); });
        run("ws2 ring 5, 4 producers, prefetch 24 [n][k] COLD", [&] { WS2(5, false, 4, 24, 0, Bs[rr++ & 7], K, 0); });
        run("ws2 ring 5, 4 producers, prefetch 16 [k][n] COLD", [&] { WS2(5, true, 4, 16, 0, Bts[rr++ & 7], N, 0); });
    }
    {
        printf("---- per-k-tile slopes: T(k range walked twice) - T(once), / 64 tiles; warm\n");
        auto timeit = [&](auto launch) {
            std::vector<float> ts;
            for (int r = 0; r < 7; ++r) {
                launch(); launch();
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 20; ++i) launch();
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 20 * 1000.f);
            }
            std::sort(ts.begin(), ts.end());
            return ts[3];
        };
        auto slope = [&](const char *name, auto l1, auto l2) {
            const float t1 = timeit(l1), t2 = timeit(l2);
            printf("%-64s T1 %6.2f us  T2 %6.2f us  -> %6.1f ns per k-tile, fixed part %5.2f us\n", name, t1, t2, (t2 - t1) / 64 * 1000.f, t1 - (t2 - t1));
            fflush(stdout);
        };
        slope("v2 [n][k]", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, B, C, K, K, N, K, tm, tn, 1000); },
                           [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, B, C, K, K, N, K, tm, tn, 2000); });
        slope("ws2 full [n][k]", [&] { WS2(5, false, 4, 0, 0, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 0, B, K, 2000); });
        slope("ws2 INTERLEAVED reads, full [n][k]", [&] { WS2(5, false, 4, 0, 6, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 6, B, K, 2000); });
        slope("ws2 INTERLEAVED reads, full [k][n]", [&] { WS2(5, true, 4, 0, 6, Bt, N, 1000); }, [&] { WS2(5, true, 4, 0, 6, Bt, N, 2000); });
        slope("ws2 INTERLEAVED reads, no DMA", [&] { WS2(5, false, 4, 0, 7, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 7, B, K, 2000); });
        slope("ws2 INTERLEAVED, ring 4", [&] { WS2(4, false, 4, 0, 6, B, K, 1000); }, [&] { WS2(4, false, 4, 0, 6, B, K, 2000); });
        slope("ws2 INTERLEAVED, ring 6", [&] { WS2(6, false, 4, 0, 6, B, K, 1000); }, [&] { WS2(6, false, 4, 0, 6, B, K, 2000); });
        slope("ws2 INTERLEAVED, 2 producers", [&] { WS2(5, false, 2, 0, 6, B, K, 1000); }, [&] { WS2(5, false, 2, 0, 6, B, K, 2000); });
        slope("ws2 no DMA", [&] { WS2(5, false, 4, 0, 1, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 1, B, K, 2000); });
        slope("ws2 no fragment reads", [&] { WS2(5, false, 4, 0, 2, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 2, B, K, 2000); });
        slope("ws2 MFMAs only (no DMA, no reads, no barriers)", [&] { WS2(5, false, 4, 0, 3, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 3, B, K, 2000); });
        slope("ws2 MFMAs + barriers", [&] { WS2(5, false, 4, 0, 4, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 4, B, K, 2000); });
        slope("ws2 MFMAs + reads (no DMA, no barriers)", [&] { WS2(5, false, 4, 0, 5, B, K, 1000); }, [&] { WS2(5, false, 4, 0, 5, B, K, 2000); });
    }
    if (getenv("WS_ROUND1"))
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d\n", pass);
        run("v2 (rounds 4-5) [n][k] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, B, C, K, K, N, K, tm, tn, 0); });
        run("v2 (rounds 4-5) [k][n] warm", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bt, C, K, N, N, K, tm, tn, 0); });
        run("ws ring 5, 4 producers, prio 0 [n][k] warm", [&] { WS(5, false, 4, 0, 0, B, K, 0); });
        run("ws ring 5, 4 producers, prio 1 [n][k] warm", [&] { WS(5, false, 4, 1, 0, B, K, 0); });
        run("ws ring 5, 8 producers, prio 0 [n][k] warm", [&] { WS(5, false, 8, 0, 0, B, K, 0); });
        run("ws ring 5, 8 producers, prio 1 [n][k] warm", [&] { WS(5, false, 8, 1, 0, B, K, 0); });
        run("ws ring 6, 4 producers, prio 1 [n][k] warm", [&] { WS(6, false, 4, 1, 0, B, K, 0); });
        run("ws ring 6, 8 producers, prio 1 [n][k] warm", [&] { WS(6, false, 8, 1, 0, B, K, 0); });
        run("ws ring 4, 4 producers, prio 1 [n][k] warm", [&] { WS(4, false, 4, 1, 0, B, K, 0); });
        run("ws ring 5, 12 producers, prio 1 [n][k] warm", [&] { WS(5, false, 12, 1, 0, B, K, 0); });
        run("ws ring 5, 4 producers, prio 1 [k][n] warm", [&] { WS(5, true, 4, 1, 0, Bt, N, 0); });
        run("ws ring 5, 8 producers, prio 1 [k][n] warm", [&] { WS(5, true, 8, 1, 0, Bt, N, 0); });
        run("  ws ring 5, 4 producers: no DMA in the loop (wrong result)", [&] { WS(5, false, 4, 1, 1, B, K, 0); });
        run("  ws ring 5, 4 producers: no fragment reads (wrong result)", [&] { WS(5, false, 4, 1, 2, B, K, 0); });
        run("  ws ring 5, 8 producers: no fragment reads (wrong result)", [&] { WS(5, false, 8, 1, 2, B, K, 0); });
        run("v2 (rounds 4-5) [n][k] COLD, sharers 2 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, false, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bs[rr++ & 7], C, K, K, N, K, tm, tn, 2); });
        run("v2 (rounds 4-5) [k][n] COLD, sharers 4 tiles apart", [&] { hipLaunchKernelGGL((gemm_dma_v2<4, true, 2>), dim3(tm * tn), dim3(256), 0, st, A, Bts[rr++ & 7], C, K, N, N, K, tm, tn, 4); });
        run("ws ring 5, 4 producers, prio 1 [n][k] COLD", [&] { WS(5, false, 4, 1, 0, Bs[rr++ & 7], K, 0); });
        run("ws ring 5, 4 producers, prio 1 [n][k] COLD, sharers 2 apart", [&] { WS(5, false, 4, 1, 0, Bs[rr++ & 7], K, 2); });
        run("ws ring 6, 4 producers, prio 1 [n][k] COLD, sharers 2 apart", [&] { WS(6, false, 4, 1, 0, Bs[rr++ & 7], K, 2); });
        run("ws ring 5, 8 producers, prio 1 [n][k] COLD, sharers 2 apart", [&] { WS(5, false, 8, 1, 0, Bs[rr++ & 7], K, 2); });
        run("ws ring 6, 8 producers, prio 1 [n][k] COLD, sharers 2 apart", [&] { WS(6, false, 8, 1, 0, Bs[rr++ & 7], K, 2); });
        run("ws ring 5, 4 producers, prio 1 [k][n] COLD, sharers 4 apart", [&] { WS(5, true, 4, 1, 0, Bts[rr++ & 7], N, 4); });
        run("ws ring 6, 8 producers, prio 1 [k][n] COLD, sharers 4 apart", [&] { WS(6, true, 8, 1, 0, Bts[rr++ & 7], N, 4); });
    }
    return 0;
}
