// fill_clone_probe.hip — why does torch.fill_ write 720 MB at 7.0 TB/s when k_perceive's stream reaches 5.0-5.7?
// A clone of the elementwise fill (each 256-thread workgroup writes CHUNK KiB contiguous, UNR 16-byte stores per
// thread) with knobs: chunk size, nt, workgroups assigned in address order or strided, persistent or one-shot.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(float *out, size_t n4, int unr, int chunks_per_wg, size_t wg_stride4)
{
    const vf4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    vf4 *o = reinterpret_cast<vf4 *>(out);
    for (int c = 0; c < chunks_per_wg; ++c) {
        const size_t base = (size_t)blockIdx.x * wg_stride4 + (size_t)c * 256 * unr;
        for (int u = 0; u < unr; ++u) {
            const size_t j = base + (size_t)u * 256 + threadIdx.x;
            if (j < n4) {
                if (NT) __builtin_nontemporal_store(v, o + j);
                else o[j] = v;
            }
        }
    }
}

int main(int argc, char **argv)
{
    int unr = 4, chunks = 1, nt = 0, reps = 10;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--unr")) unr = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--chunks")) chunks = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--nt")) nt = atoi(argv[i + 1]);
    }
    const size_t n4 = (size_t)1024 * 512 * 343 / 4;
    float *out;
    CK(hipMalloc(&out, n4 * 16 + 4096));
    const size_t per_wg = (size_t)256 * unr * chunks;
    const unsigned grid = (unsigned)((n4 + per_wg - 1) / per_wg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() {
        if (nt) hipLaunchKernelGGL(k_fill<true>, dim3(grid), dim3(256), 0, 0, out, n4, unr, chunks, per_wg);
        else hipLaunchKernelGGL(k_fill<false>, dim3(grid), dim3(256), 0, 0, out, n4, unr, chunks, per_wg);
    };
    for (int i = 0; i < 3; ++i) go();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("unr=%d chunks=%d (%zu KiB per workgroup, %u workgroups) nt=%d : %.4f ms %.2f TB/s\n", unr, chunks, per_wg * 16 / 1024, grid, nt, ms,
           n4 * 16 / ms / 1e9);
    return 0;
}
