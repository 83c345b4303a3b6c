// antsrl_update.hip — Environment.update minus the pheromone sweep (environment.py:42-47): k_update
// (per-phase loops, any N), k_update_one (one ant per thread, N <= 1024), k_collect_full, launchers.
#include "antsrl_util.h"
#include "antsrl_update_env.h"
#include "antsrl_update_one.h"

template <int C>
__global__ void __launch_bounds__(1024)
k_update(const KP p, const double *__restrict__ wall_jitter, const int out_buf, const int phases)
{
    extern __shared__ __align__(16) unsigned char smem[];
    update_env<C>(p, blockIdx.x, wall_jitter, out_buf, smem, phases);
}

template <int C>
__global__ void __launch_bounds__(1024)
k_update_one(const KP p, const double *__restrict__ wall_jitter, const int out_buf)
{
    extern __shared__ __align__(16) unsigned char smem[];
    update_one_body<C>(p, blockIdx.x, wall_jitter, out_buf, smem, p.g_dep, p.inv_g_dep);
}

// Anthill.update over the WHOLE grid (anthill.py:41-46): needed on the first update after a
// reset (food may lie on the anthill area) and after two steps without an update.
__global__ void __launch_bounds__(256) k_collect_full(const KP p)
{
    __shared__ double red[4];
    const int e = blockIdx.x, tid = threadIdx.x;
    const size_t G = (size_t)p.W * p.H;
    const FoodView food{p.s.food + (size_t)e * G * p.fs, p.fs};
    const uint32_t *area = p.s.area_bits + (size_t)e * p.words;
    double gain = 0.0;
    for (size_t g = tid; g < G; g += blockDim.x)
        if (test_bit(area, (uint32_t)g)) {
            const uint32_t r = frec_cell(p, (uint32_t)g);
            gain += (double)food[r];
            food[r] = 0.0f;
        }
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if ((tid & 63) == 0) red[tid >> 6] = gain;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        p.s.anthill_food[e] += s;
    }
    for (int i = tid; i < p.N; i += blockDim.x) p.s.dirty_cell[(size_t)e * p.N + i] = -1;
}

// host-side launchers (called from antsrl_capi.hip)
static inline int pick_update_threads(int N) { (void)N; return 512; }
static size_t update_lds_bytes(const KP &p, int threads) { return update_scratch_bytes(p.HT, p.R, threads / 64); }

// phases: UPD_ALL (one Environment.update) or a subset of its steps (antsrl_update_phase: the per-phase loop kernel runs them)
hipError_t antsrl_launch_update(const KP &p, const double *jitter, int out_buf, hipStream_t st, int phases)
{
    static const bool force_loops = PROF_ENV("ANTSRL_UPDATE_LOOPS") != nullptr; // A/B: the per-phase loop kernel
    if (p.N <= 1024 && !force_loops && phases == UPD_ALL) { // one ant per thread
        const int t1 = (p.N + 63) / 64 * 64;
        const size_t l1 = update_one_lds_bytes(p.HT, p.R, t1 / 64, p.N);
        switch (p.C) {
        case 1: hipLaunchKernelGGL((k_update_one<1>), dim3(p.E), dim3(t1), l1, st, p, jitter, out_buf); break;
        case 2: hipLaunchKernelGGL((k_update_one<2>), dim3(p.E), dim3(t1), l1, st, p, jitter, out_buf); break;
        case 3: hipLaunchKernelGGL((k_update_one<3>), dim3(p.E), dim3(t1), l1, st, p, jitter, out_buf); break;
        case 4: hipLaunchKernelGGL((k_update_one<4>), dim3(p.E), dim3(t1), l1, st, p, jitter, out_buf); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const int threads = pick_update_threads(p.N);
    const size_t lds = update_lds_bytes(p, threads);
    // more than 64 KiB of dynamic LDS (the hash of > 2048 ants) is an opt-in per kernel function and device
#define UPDATE_GO(CC)                                                                                              \
    {                                                                                                              \
        static size_t seen[ANTSRL_MAX_DEVICES] = {};                                                               \
        int dev = 0;                                                                                               \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice; \
        if (lds > 64 * 1024 && lds > seen[dev]) {                                                                  \
            hipError_t err = hipFuncSetAttribute((const void *)k_update<CC>,                                       \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (err != hipSuccess) return err;                                                                     \
            seen[dev] = lds;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL((k_update<CC>), dim3(p.E), dim3(threads), lds, st, p, jitter, out_buf, phases);         \
    }
    switch (p.C) {
    case 1: UPDATE_GO(1) break;
    case 2: UPDATE_GO(2) break;
    case 3: UPDATE_GO(3) break;
    case 4: UPDATE_GO(4) break;
    default: return hipErrorInvalidValue;
    }
#undef UPDATE_GO
    return hipGetLastError();
}

hipError_t antsrl_launch_collect_full(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_collect_full, dim3(p.E), dim3(256), 0, st, p);
    return hipGetLastError();
}
