// store cache-policy probe: raw buffer stores with each aux (sc0 / nt / sc1) combination, wave-per-row runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(512) void k_run_rows_buf(float *out, int N, int row)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, per = N / 8;
    float *env = out + (size_t)blockIdx.x * N * row;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(env, 0, N * row * 4, 0x00020000);
    for (int i = wave * per; i < (wave + 1) * per; ++i) {
        const uint32_t off = (uint32_t)i * row;          // floats
        const uint32_t mis = (uint32_t)((((uintptr_t)env >> 2) + off) & 3);
        const uint32_t al = off - mis;
        const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
        const vf4 v = {(float)i, (float)lane, 1.0f, 2.0f};
        for (uint32_t j = j_lo + lane; j < j_hi; j += 64)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rs, (al + 4 * j) * 4, 0, AUX);
        const uint32_t hd = 4 * j_lo - mis, tl = mis + row - 4 * j_hi;
        if ((uint32_t)lane < hd) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, 3.0f), rs, (al + mis + lane) * 4, 0, AUX);
        else if ((uint32_t)lane - hd < tl) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, 3.0f), rs, (al + 4 * j_hi + (lane - hd)) * 4, 0, AUX);
    }
}


// carry scheme: each wave streams its contiguous run as aligned blocks of GRAN floats (whole 64- or
// 128-byte units); the remainder of a row is carried into the next row's flush.  Scalar edges only at
// the two ends of the run.
template <int AUX, int GRAN>
__global__ __launch_bounds__(512) void k_run_carry_buf(float *out, int N, int row)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, per = N / 8;
    float *env = out + (size_t)blockIdx.x * N * row;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(env, 0, N * row * 4, 0x00020000);
    const uint32_t start = (uint32_t)wave * per * row;                       // floats, relative to env
    const uint32_t abs0 = (uint32_t)(((uintptr_t)env >> 2) + start);         // absolute float address (low bits)
    const uint32_t mis = abs0 & (GRAN - 1);
    const uint32_t al = start - mis;                                         // aligned window start (may be < start)
    const uint32_t total = mis + (uint32_t)per * row;
    uint32_t done = mis ? GRAN : 0;                                          // floats flushed (first unit by scalars)
    if (mis && (uint32_t)lane >= mis && lane < GRAN) __builtin_amdgcn_raw_buffer_store_b32(0x40400000u, rs, (al + lane) * 4, 0, AUX);
    const vf4 v = {1.0f, (float)lane, 1.0f, 2.0f};
    for (int k = 0; k < per; ++k) {
        const uint32_t end = (mis + (uint32_t)(k + 1) * row) / GRAN * GRAN;
        for (uint32_t f = done + 4 * lane; f < end; f += 256)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rs, (al + f) * 4, 0, AUX);
        done = end;
    }
    if (done + lane < total) __builtin_amdgcn_raw_buffer_store_b32(0x40400000u, rs, (al + done + lane) * 4, 0, AUX);
}

template <class F> static double time_ms(F launch, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}
#define RUN(AUX) { const double t = time_ms([&] { k_run_rows_buf<AUX><<<E, 512>>>(buf, N, r); }, 30); \
    printf("aux=%2d (%s%s%s) %.4f ms %.2f TB/s\n", AUX, (AUX & 1) ? "sc0 " : "", (AUX & 2) ? "nt " : "", (AUX & 16) ? "sc1" : "", t, gb / t); }
int main()
{
    const int E = 1024, N = 512, r = 343;
    float *buf; CK(hipMalloc(&buf, (size_t)E * N * 352 * 4 + 256));
    const double gb = (double)E * N * r * 4 / 1e9;
    for (int rep = 0; rep < 2; ++rep) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(16) RUN(17) RUN(18) RUN(19) }
#define RUNC(AUX, GRAN) { const double t = time_ms([&] { k_run_carry_buf<AUX, GRAN><<<E, 512>>>(buf, N, r); }, 30); \
    printf("carry aux=%2d granularity %3d B: %.4f ms %.2f TB/s\n", AUX, GRAN * 4, t, gb / t); }
    for (int rep = 0; rep < 2; ++rep) { RUNC(0, 16) RUNC(0, 32) RUNC(2, 4) RUNC(2, 16) RUNC(2, 32) RUNC(2, 64) }
    return 0;
}
