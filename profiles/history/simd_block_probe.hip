// simd_block_probe.hip — does a wave that is stalled at the ISSUE of an HBM-bound global store keep the
// other waves of its SIMD from issuing VALU work?  (k_perceive: 72 us of VALU per SIMD + 165 us of
// observation stores take 250 us, not 165.)
//
// 1024 workgroups x 512 threads (8 waves, 2 per SIMD, 2 workgroups per CU).  ROWS rows of 2 KiB per wave are
// streamed with nt stores (0.75 GB per launch), WORK x 8 FMAs per lane per row are the compute.
//   mode 0  stores only                     mode 1  compute only
//   mode 2  every wave: compute a row, store a row (what k_perceive does)
//   mode 3  split BY SIMD: waves on SIMD 3 store everything, waves on SIMDs 0-2 compute everything
//   mode 4  split WITHIN each SIMD: of the two waves of a SIMD one stores (two waves' rows), one computes
//   mode 5  like 2, but a wave waits for its previous row's stores (vmcnt(0)) before issuing the next ones
//           (parks in s_waitcnt instead of stalling at issue)
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/simd_block_probe profiles/simd_block_probe.hip && /tmp/simd_block_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define ROWS 90 // per wave

__device__ __forceinline__ void compute(float &a, float &b, float &c, float &d, const float k, const int work)
{
    for (int i = 0; i < work; ++i) { // 8 independent-enough FMAs per iteration (4 chains x 2)
        a = fmaf(a, k, 1.0f); b = fmaf(b, k, 1.0f); c = fmaf(c, k, 1.0f); d = fmaf(d, k, 1.0f);
        a = fmaf(a, k, 0.5f); b = fmaf(b, k, 0.5f); c = fmaf(c, k, 0.5f); d = fmaf(d, k, 0.5f);
    }
}

__device__ __forceinline__ void store_row(float *row, int lane, float a, float b)
{
    __builtin_nontemporal_store(vf4{a, b, 1.0f, 2.0f}, reinterpret_cast<vf4 *>(row) + lane);
    __builtin_nontemporal_store(vf4{b, a, 3.0f, 4.0f}, reinterpret_cast<vf4 *>(row) + 64 + lane);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_probe(float *out, const int work, const float seed, int *simd_census)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int simd = (__builtin_amdgcn_s_getreg(4 | (31 << 11)) >> 4) & 3; // HW_REG_HW_ID.SIMD_ID
    if (blockIdx.x == 0 && lane == 0) simd_census[wave] = simd;
    float *base = out + (size_t)blockIdx.x * 8 * ROWS * 512; // 8 waves x ROWS rows x 512 floats
    float a = seed + lane, b = seed + 1, c = seed + 2, d = seed + 3;
    const float k = seed * 0.25f;
    if (MODE == 0) {
        for (int r = 0; r < ROWS; ++r) store_row(base + ((size_t)wave * ROWS + r) * 512, lane, a, b);
    } else if (MODE == 1) {
        for (int r = 0; r < ROWS; ++r) compute(a, b, c, d, k, work);
    } else if (MODE == 2 || MODE == 5) {
        for (int r = 0; r < ROWS; ++r) {
            compute(a, b, c, d, k, work);
            if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_row(base + ((size_t)wave * ROWS + r) * 512, lane, a, b);
        }
    } else if (MODE == 3) {
        // roles by the SIMD the wave sits on: a 512-thread workgroup has two waves on each SIMD
        __shared__ int n_store, n_comp, idx_store[8], idx_comp[8];
        if (threadIdx.x == 0) { n_store = 0; n_comp = 0; }
        __syncthreads();
        int me = -1;
        if (lane == 0) me = simd == 3 ? atomicAdd(&n_store, 1) : atomicAdd(&n_comp, 1);
        me = __builtin_amdgcn_readfirstlane(me);
        __syncthreads();
        const int ns = n_store, nc = n_comp;
        if (simd == 3) { // this wave stores rows [me*total/ns, (me+1)*total/ns)
            const int total = 8 * ROWS, lo = me * total / ns, hi = (me + 1) * total / ns;
            for (int r = lo; r < hi; ++r) store_row(base + (size_t)r * 512, lane, a, b);
        } else {
            const int total = 8 * ROWS, lo = me * total / nc, hi = (me + 1) * total / nc;
            for (int r = lo; r < hi; ++r) compute(a, b, c, d, k, work);
        }
    } else if (MODE == 4) {
        // roles within each SIMD: the first wave to arrive on a SIMD stores, the second computes
        __shared__ int seen[4];
        if (threadIdx.x < 4) seen[threadIdx.x] = 0;
        __syncthreads();
        int me = -1;
        if (lane == 0) me = atomicAdd(&seen[simd], 1);
        me = __builtin_amdgcn_readfirstlane(me);
        if ((me & 1) == 0) {
            for (int r = 0; r < 2 * ROWS; ++r) store_row(base + ((size_t)(wave >> 1) * 2 * ROWS + r) * 512, lane, a, b);
        } else {
            for (int r = 0; r < 2 * ROWS; ++r) compute(a, b, c, d, k, work);
        }
    }
    if (a + b + c + d == 12345.678f) out[0] = a;
}

template <int MODE>
static float run(float *out, int work, int *census, int E)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_probe<MODE>, dim3(E), dim3(512), 0, 0, out, work, 1.0f, census);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_probe<MODE>, dim3(E), dim3(512), 0, 0, out, work, 1.0f, census);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const int E = 1024;
    const size_t bytes = (size_t)E * 8 * ROWS * 512 * 4;
    float *out;
    int *census;
    CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&census, 64));
    CK(hipMemset(out, 0, bytes));
    printf("%.2f GB per launch, %d workgroups x 512 threads\n", bytes / 1e9, E);
    const float ts = run<0>(out, 0, census, E);
    int h[8];
    CK(hipMemcpy(h, census, 32, hipMemcpyDeviceToHost));
    printf("SIMD of waves 0..7 of workgroup 0: %d %d %d %d %d %d %d %d\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    printf("stores only: %.4f ms (%.2f TB/s)\n", ts, bytes / ts / 1e9);
    for (int work : {8, 16, 32, 64}) {
        const float tc = run<1>(out, work, census, E), t2 = run<2>(out, work, census, E), t3 = run<3>(out, work, census, E),
                    t4 = run<4>(out, work, census, E), t5 = run<5>(out, work, census, E);
        printf("work=%3d  compute %.4f  stores %.4f | both-in-every-wave %.4f  +vmcnt0 %.4f | split by SIMD %.4f (compute on 3 SIMDs alone would be %.4f) | "
               "split within SIMD %.4f\n", work, tc, ts, t2, t5, t3, tc * 4 / 3, t4);
    }
    return 0;
}
