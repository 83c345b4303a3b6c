// gather_layout_probe.hip — does a TILED cell-record layout make k_perceive's gathers cheaper (round 3)?
// k_perceive gathers one 16-byte record per perceived cell: 49 cells of a rotated 7x7 patch (spacing 1.1) per ant.  With the
// row-major layout (cell = x * H + y, 8 cells per 128-byte line along y) a patch touches ~10 different x rows = 10 places
// 4 KB apart (different DRAM pages), ~19-21 lines.  A tile of TX x 8 cells stored contiguously puts the lines a patch needs
// into 2-4 contiguous 0.5-2 KB pieces.  This probe issues exactly those gathers (no stores, trivial use of the values) for
// E x N ants at random positions / headings and reports time per launch for each layout:
//   --layout 0 row-major   1: 8x8 tiles (1 KB)   2: 4x8 tiles   3: 2x8   4: 16x8   5: 8x8 tiles whose lines are 2x4 blocks
//   --spread S   ants uniform in a (2S)^2 window around the grid centre (S = 128: whole 256^2 grid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int L>
__device__ __forceinline__ uint32_t slot(const int x, const int y)
{
    if (L == 0) return (uint32_t)(x * 256 + y);
    if (L == 1) return (uint32_t)((((x >> 3) * 32 + (y >> 3)) << 6) + ((x & 7) << 3) + (y & 7));
    if (L == 2) return (uint32_t)((((x >> 2) * 32 + (y >> 3)) << 5) + ((x & 3) << 3) + (y & 7));
    if (L == 3) return (uint32_t)((((x >> 1) * 32 + (y >> 3)) << 4) + ((x & 1) << 3) + (y & 7));
    if (L == 4) return (uint32_t)((((x >> 4) * 32 + (y >> 3)) << 7) + ((x & 15) << 3) + (y & 7));
    return (uint32_t)((((x >> 3) * 32 + (y >> 3)) << 6) + (((x & 7) >> 1) << 4) + (((y & 7) >> 2) << 3) + ((x & 1) << 2) + (y & 3));
}

template <int L>
__global__ void __launch_bounds__(256) k_gather(const vf4 *__restrict__ table, const float4 *__restrict__ ants, float *__restrict__ out,
                                                const int N, const int run)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nseg = N / (4 * run);
    const int b = blockIdx.x;
    const int e = ((b >> 3) / nseg) * 8 + (b & 7), seg = (b >> 3) % nseg;
    const int i0 = (seg * 4 + wave) * run;
    const int q = lane < 49 ? lane : 48;
    const float ox = (float)(q % 7 - 3) * 1.1f, oy = (float)(q / 7 - 3) * 1.1f;
    const vf4 *cells = table + (size_t)e * 65536;
    vf4 acc = {0, 0, 0, 0};
    vf4 cur[2], nxt[2];
    auto fetch = [&](int j, vf4 *dst) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 a = ants[(size_t)e * N + min(i0 + j + u, i0 + run - 1)]; // x, y, cos, sin (wave-uniform)
            const float rx = a.z * ox - a.w * oy, ry = a.w * ox + a.z * oy;
            const int ix = (int)rintf(rx + a.x) & 255, iy = (int)rintf(ry + a.y) & 255;
            dst[u] = cells[slot<L>(ix, iy)];
        }
    };
    fetch(0, cur);
    for (int j = 0; j < run; j += 2) {
        fetch(j + 2, nxt);
#pragma unroll
        for (int u = 0; u < 2; ++u) { acc += cur[u]; cur[u] = nxt[u]; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char **argv)
{
    int E = 1024, N = 512, layout = 0, spread = 128, run = 8, reps = 10;
    for (int i = 1; i + 1 < argc; i += 2) {
        const char *k = argv[i];
        const int v = atoi(argv[i + 1]);
        if (!strcmp(k, "--envs")) E = v; else if (!strcmp(k, "--layout")) layout = v; else if (!strcmp(k, "--spread")) spread = v;
        else if (!strcmp(k, "--run")) run = v;
    }
    vf4 *table;
    float4 *ants, *h = (float4 *)malloc((size_t)E * N * 16);
    float *out;
    CK(hipMalloc(&table, (size_t)E * 65536 * 16));
    CK(hipMalloc(&ants, (size_t)E * N * 16));
    CK(hipMalloc(&out, 256));
    CK(hipMemset(table, 0, (size_t)E * 65536 * 16));
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (size_t i = 0; i < (size_t)E * N; ++i) {
        const double th = rnd() * 6.283185307179586;
        h[i] = make_float4((float)(128 + (rnd() * 2 - 1) * spread), (float)(128 + (rnd() * 2 - 1) * spread), (float)cos(th), (float)sin(th));
    }
    CK(hipMemcpy(ants, h, (size_t)E * N * 16, hipMemcpyHostToDevice));
    const dim3 grid(E * (N / (4 * run))), block(256);
    auto launch = [&]() {
        switch (layout) {
        case 0: hipLaunchKernelGGL((k_gather<0>), grid, block, 0, 0, table, ants, out, N, run); break;
        case 1: hipLaunchKernelGGL((k_gather<1>), grid, block, 0, 0, table, ants, out, N, run); break;
        case 2: hipLaunchKernelGGL((k_gather<2>), grid, block, 0, 0, table, ants, out, N, run); break;
        case 3: hipLaunchKernelGGL((k_gather<3>), grid, block, 0, 0, table, ants, out, N, run); break;
        case 4: hipLaunchKernelGGL((k_gather<4>), grid, block, 0, 0, table, ants, out, N, run); break;
        default: hipLaunchKernelGGL((k_gather<5>), grid, block, 0, 0, table, ants, out, N, run); break;
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("envs=%d layout=%d spread=%d run=%d : %.4f ms per launch\n", E, layout, spread, run, ms / reps);
    return 0;
}
