// valu_rate_probe.hip — issue cost (cycles per wave64 instruction) of the VALU/LDS operations k_act's
// perception loop is made of, measured on the device it runs on.  One wave per SIMD, 8 independent
// chains per instruction so latency is hidden; cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o probe profiles/valu_rate_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITERS 2048

#define PROBE64(NAME, ASM)                                                                       \
    __global__ void NAME(unsigned long long *out, double seed)                                   \
    {                                                                                            \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
               a6 = a0 + 6, a7 = a0 + 7, b = seed * 0.5 + 1.0;                                   \
        const unsigned long long t0 = __builtin_readcyclecounter();                              \
        for (int i = 0; i < ITERS; ++i) {                                                        \
            asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));        \
            asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));        \
            asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b));        \
            asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));        \
        }                                                                                        \
        const unsigned long long t1 = __builtin_readcyclecounter();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1] = 1;                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                               \
    }

#define PROBE32(NAME, ASM)                                                                       \
    __global__ void NAME(unsigned long long *out, double seed)                                   \
    {                                                                                            \
        float a0 = (float)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
              a6 = a0 + 6, a7 = a0 + 7, b = (float)seed * 0.5f + 1.0f;                           \
        const unsigned long long t0 = __builtin_readcyclecounter();                              \
        for (int i = 0; i < ITERS; ++i) {                                                        \
            asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));        \
            asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));        \
            asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b));        \
            asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));        \
        }                                                                                        \
        const unsigned long long t1 = __builtin_readcyclecounter();                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[1] = 1;                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                               \
    }

PROBE64(p_mul_f64, "v_mul_f64 %0, %0, %1")
PROBE64(p_add_f64, "v_add_f64 %0, %0, %1")
PROBE64(p_fma_f64, "v_fma_f64 %0, %0, %1, %1")
PROBE64(p_rndne_f64, "v_rndne_f64 %0, %0")
PROBE64(p_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %1")
PROBE32(p_mul_f32, "v_mul_f32 %0, %0, %1")
PROBE32(p_fma_f32, "v_fma_f32 %0, %0, %1, %1")
PROBE32(p_rndne_f32, "v_rndne_f32 %0, %0")
PROBE32(p_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
PROBE32(p_add_u32, "v_add_u32 %0, %0, %1")
PROBE32(p_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
PROBE32(p_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
PROBE32(p_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
PROBE32(p_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
PROBE32(p_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE32(p_and_b32, "v_and_b32 %0, %0, %1")
PROBE32(p_ds_read_b32, "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)")
PROBE32(p_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
PROBE32(p_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
PROBE32(p_cmp, "v_cmp_lt_f32 vcc, %0, %1")
PROBE32(p_max_i32, "v_max_i32 %0, %0, %1")
PROBE32(p_mad_u64_u32, "v_mad_u64_u32 v[100:101], s[20:21], %0, %1, v[102:103]")
PROBE32(p_ds_read8, "ds_read_b32 v100, %1\n ds_read_b32 v101, %1 offset:256\n ds_read_b32 v102, %1 offset:512\n ds_read_b32 v103, %1 offset:768\n s_waitcnt lgkmcnt(0)")
PROBE32(p_ds_write, "ds_write_b32 %1, %0")
PROBE32(p_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

// f64 -> i32 and i32 -> f64 conversions change register width: separate bodies
__global__ void p_cvt_i32_f64(unsigned long long *out, double seed)
{
    double a = seed + threadIdx.x;
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r1) : "v"(a));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r2) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r3) : "v"(a));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r4) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r5) : "v"(a));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r6) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r7) : "v"(a));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 123456789) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

__global__ void p_cvt_f64_i32(unsigned long long *out, double seed)
{
    int a = (int)seed + threadIdx.x;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r1) : "v"(a));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r2) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r3) : "v"(a));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r4) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r5) : "v"(a));
        asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r6) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r7) : "v"(a));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long *, double);

int main()
{
    unsigned long long *d, h[2];
    CK(hipMalloc(&d, 16));
    struct { const char *name; kern_t k; } tab[] = {
        {"v_mul_f64", p_mul_f64}, {"v_add_f64", p_add_f64}, {"v_fma_f64", p_fma_f64}, {"v_rndne_f64", p_rndne_f64},
        {"v_cvt_i32_f64", p_cvt_i32_f64}, {"v_cvt_f64_i32", p_cvt_f64_i32}, {"v_lshl_add_u64", p_lshl_add_u64},
        {"v_mul_f32", p_mul_f32}, {"v_fma_f32", p_fma_f32}, {"v_rndne_f32", p_rndne_f32}, {"v_cvt_i32_f32", p_cvt_i32_f32},
        {"v_add_u32", p_add_u32}, {"v_mul_lo_u32", p_mul_lo_u32}, {"v_mul_hi_u32", p_mul_hi_u32},
        {"v_mad_u32_u24", p_mad_u32_u24}, {"v_lshl_add_u32", p_lshl_add_u32}, {"v_cndmask_b32", p_cndmask},
        {"v_and_b32", p_and_b32}, {"v_cndmask sgpr", p_cndmask_sgpr}, {"v_cmp+v_cndmask", p_cmp_cnd}, {"v_cmp_lt_f32", p_cmp},
        {"v_max_i32", p_max_i32}, {"v_mad_u64_u32", p_mad_u64_u32}, {"4x ds_read_b32+wait", p_ds_read8}, {"ds_write_b32", p_ds_write}, {"ds_read_b32+wait", p_ds_read_b32}, {"v_mov_b32_dpp", p_mov_dpp},
    };
    const int wl[] = {1, 2, 4, 8};
    for (int waves : wl) {
        const int blocks = waves == 8 ? 512 : 256, threads = waves == 8 ? 1024 : 256 * waves;
        printf("-- %d wave(s) per SIMD, cycles per wave-instruction as seen by one wave\n", waves);
        for (auto &t : tab) {
            CK(hipMemset(d, 0, 16));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, d, 1.5); // warm
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, d, 1.5);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            // per-SIMD issue interval in ns: kernel time / instructions one SIMD executed
            const double per_simd = (double)waves * ITERS * 8.0;
            printf("%-20s counter %6.2f   wall %6.3f ns per SIMD instruction (= %.2f cycles at 2.4 GHz)\n", t.name,
                   (double)h[0] / (ITERS * 8.0), ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
        }
    }
    CK(hipFree(d));
    return 0;
}
