// Box diagnosis (profiles/r04/box_diag.sh): idle HBM latency by pointer chase, and the rate of scattered 16-byte gathers
// with / without a streaming nt write beside them — the two ingredients of k_perceive, stand-alone.
//   hipcc --offload-arch=gfx950 -O3 profiles/r04/lat_probe.hip -o profiles/r04/bin/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_chase(const uint32_t *__restrict__ next, uint32_t start, int hops, uint32_t *out, long long *ticks)
{
    uint32_t i = start;
    const long long t0 = wall_clock64();
    for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(next + (size_t)i * 32); // one 128-byte line per hop
    const long long t1 = wall_clock64();
    *out = i;
    *ticks = t1 - t0;
}

// every lane gathers 16 bytes from a pseudo-random line of `src` (n16 16-byte elements), `iters` times, 4 in flight
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ src, size_t n16, int iters, uint4 *sink, uint4 *__restrict__ wr, int do_write)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = s * 1664525u + 1013904223u;
            v[k] = src[(size_t)(s >> 4) % n16];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
        if (do_write) { // a contiguous 1 KiB per wave and iteration, nt: the observation stream's shape
            uint4 *p = wr + ((size_t)it * gridDim.x * blockDim.x + tid);
            __builtin_nontemporal_store(acc.x, &p->x); __builtin_nontemporal_store(acc.y, &p->y);
            __builtin_nontemporal_store(acc.z, &p->z); __builtin_nontemporal_store(acc.w, &p->w);
        }
    }
    if (acc.x == 0x12345678u) sink[tid] = acc;
}

int main()
{
    const size_t lines = (size_t)1 << 23; // 1 GiB of 128-byte lines
    std::vector<uint32_t> perm(lines);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    for (size_t i = lines - 1; i > 0; --i) std::swap(perm[i], perm[rng() % (i + 1)]);
    std::vector<uint32_t> host(lines * 32, 0u);
    for (size_t i = 0; i < lines; ++i) host[(size_t)perm[i] * 32] = perm[(i + 1) % lines];
    uint32_t *d; uint32_t *out; long long *ticks;
    CK(hipMalloc(&d, lines * 128)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticks, 64));
    CK(hipMemcpy(d, host.data(), lines * 128, hipMemcpyHostToDevice));
    const int hops = 20000;
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, perm[(size_t)r * 2500000 + 17], hops, out, ticks);
        CK(hipDeviceSynchronize());
        long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
        printf("chase: %.1f ns per dependent line (1 GiB, %d hops)\n", t * 10.0 / hops, hops);
    }
    // gathers over the same 1 GiB, with and without the write stream (0.7 GB)
    const int blocks = 256 * 16, iters = 32;
    uint4 *wr; CK(hipMalloc(&wr, (size_t)blocks * 256 * iters * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d, lines * 8, iters, (uint4 *)out, wr, w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double g = (double)blocks * 256 * iters * 4;
            if (r) printf("gather%s: %.3f ms  %.1f G gathers/s (%.2f TB/s of lines)%s\n", w ? " + nt write" : "", ms, g / ms / 1e6, g * 128 / ms / 1e9,
                          w ? "" : "");
        }
    return 0;
}
