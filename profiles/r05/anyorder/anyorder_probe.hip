// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950?  Kernel A (8 workgroups) spins for ~100 us; kernel B, launched
// behind it in the SAME stream, stamps its start.  In order: B starts after A's end.  Any order: B starts while A spins.
//   hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe && ./anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_spin(uint64_t *t, uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
__global__ void k_stamp(uint64_t *t)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64();
}

int main()
{
    uint64_t *d, h[3];
    hipMalloc(&d, 3 * sizeof(uint64_t));
    hipStream_t st;
    hipStreamCreate(&st);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0); // kHz
    const uint64_t ticks = (uint64_t)rate / 10; // 100 us
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 3 * sizeof(uint64_t), st);
            hipStreamSynchronize(st);
            hipLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, st, d, ticks);
            hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags, d);
            hipStreamSynchronize(st);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("flags %d: A ran %.1f us; B started %.1f us after A's start (%s A's end)\n", flags, (h[1] - h[0]) * 1e3 / rate,
                   ((double)h[2] - (double)h[0]) * 1e3 / rate, h[2] < h[1] ? "BEFORE" : "after");
        }
    }
    return 0;
}
