// How much dynamic LDS may a 256-thread workgroup ask for and still fit n per CU?  (hipOccupancyMaxActiveBlocksPerMultiprocessor
// on a trivial kernel: the LDS allocation granularity and the per-CU total of gfx950, by experiment.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *p) { extern __shared__ float s[]; s[threadIdx.x] = 1.0f; __syncthreads(); if (p) p[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    int prev = -1;
    for (int lds = 16 * 1024; lds <= 34 * 1024; lds += 16) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds) != hipSuccess) { printf("error at %d\n", lds); return 1; }
        if (n != prev) { printf("lds %6d B: %d workgroups per CU\n", lds, n); prev = n; }
    }
    return 0;
}
