// tools/tile_load_probe.hip -- development probe: how fast can 256 workgroups pull the operand tiles of the
// M=256 forward GEMM (32x64 of A + 64x64 of B per k-tile, 32 k-tiles, XCD-aware tile map) when NOTHING else is
// done with them (no LDS, no MFMA)?  Compares the row-major layout (rows 8 KB apart) with a tile-major one
// (each tile's bytes contiguous).  usage: tile_load_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// MODE 0: row-major A[m][lda] (k contiguous), B[k][ldb] (n contiguous).  MODE 1: tile-major: A tile (tile_m, kt) =
// 8 KB contiguous, B tile (kt, tile_n) = 16 KB contiguous.  DEPTH = k-tiles whose loads are in flight.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void tiles(const float *A, const float *B, int ld, int K, float *out)
{
    const int tid = threadIdx.x;
    const int b = blockIdx.x, xcd = b & 7, jj = b >> 3;
    const int tile_n = xcd * 4 + jj / 8, tile_m = jj % 8;
    const int nt = K / 64;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r[DEPTH][6];
    auto load = [&](float4 (&q)[6], int kt) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {       // A: 32 rows x 16 float4
                const int f = tid + i * 256, row = f / 16, c4 = f % 16;
                q[i] = *reinterpret_cast<const float4 *>(A + (size_t)(tile_m * 32 + row) * ld + kt * 64 + c4 * 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // B: 64 k-rows x 16 float4
                const int f = tid + i * 256, k = f / 16, c4 = f % 16;
                q[2 + i] = *reinterpret_cast<const float4 *>(B + (size_t)(kt * 64 + k) * ld + tile_n * 64 + c4 * 4);
            }
        } else {
            const float *ta = A + ((size_t)tile_m * (K / 64) + kt) * 2048, *tb = B + ((size_t)kt * 32 + tile_n) * 4096;
#pragma unroll
            for (int i = 0; i < 2; ++i) q[i] = *reinterpret_cast<const float4 *>(ta + (tid + i * 256) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[2 + i] = *reinterpret_cast<const float4 *>(tb + (tid + i * 256) * 4);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load(r[d], d);
    for (int t = 0; t < nt; t += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int tl = t + u + DEPTH - 1;
            load(r[(u + DEPTH - 1) % DEPTH], tl < nt ? tl : nt - 1);
#pragma unroll
            for (int i = 0; i < 6; ++i) { acc.x += r[u][i].x; acc.y += r[u][i].y; acc.z += r[u][i].z; acc.w += r[u][i].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int MODE, int DEPTH>
static void run(const char *name, const float *A, const float *B, float *out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tiles<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, A, B, 2048, 2048, out);
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tiles<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, A, B, 2048, 2048, out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 20 * 1000.f);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    const double bytes = 256.0 * 32 * 24576;
    printf("%-44s med %6.2f us   %5.2f TB/s into the CUs\n", name, t[t.size() / 2], bytes / t[t.size() / 2] * 1e-6);
}

int main()
{
    float *A, *B, *out;
    CK(hipMalloc(&A, (size_t)256 * 2048 * 4 + 65536)); CK(hipMalloc(&B, (size_t)2048 * 2048 * 4 + 65536)); CK(hipMalloc(&out, 256));
    CK(hipMemset(A, 0, (size_t)256 * 2048 * 4)); CK(hipMemset(B, 0, (size_t)2048 * 2048 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 2>("row-major tiles, 2 k-tiles in flight", A, B, out);
        run<0, 4>("row-major tiles, 4 k-tiles in flight", A, B, out);
        run<1, 2>("tile-major (contiguous) tiles, 2 in flight", A, B, out);
        run<1, 4>("tile-major (contiguous) tiles, 4 in flight", A, B, out);
    }
    return 0;
}
