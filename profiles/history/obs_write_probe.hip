// obs_write_probe.hip — what does the observation write stream of k_act cost on its own?
// Standalone probe (not part of the library): E workgroups x 512 threads write an [E][N][row] f32
// tensor with the access patterns k_act could use.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/obs_write_probe profiles/obs_write_probe.hip && gpurun_out/obs_write_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// wave per row, rows of `row` floats at arbitrary 4-byte alignment: 16-byte stores over the aligned
// interior, scalar stores at the edges (the pattern of k_act's copy-out)
template <bool NT>
__global__ __launch_bounds__(512) void k_rows(float *out, int N, int row)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *env = out + (size_t)blockIdx.x * N * row;
    for (int i = wave; i < N; i += 8) {
        float *dst = env + (size_t)i * row;
        const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
        float *dst_al = dst - mis;
        const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
        const float4 v = make_float4((float)i, (float)lane, 1.0f, 2.0f);
        for (uint32_t j = j_lo + lane; j < j_hi; j += 64) {
            if (NT) __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4 *>(dst_al) + j);
            else reinterpret_cast<float4 *>(dst_al)[j] = v;
        }
        const uint32_t hd = 4 * j_lo - mis, tl = mis + row - 4 * j_hi;
        if ((uint32_t)lane < hd) dst_al[mis + lane] = 3.0f;
        else if ((uint32_t)lane - hd < tl) dst_al[4 * j_hi + (lane - hd)] = 3.0f;
    }
}

// whole workgroup streams its env block in contiguous 8 KiB chunks
template <bool NT>
__global__ __launch_bounds__(512) void k_block(float *out, int N, int row)
{
    const size_t n4 = (size_t)N * row / 4;
    float4 *env = reinterpret_cast<float4 *>(out + (size_t)blockIdx.x * N * row);
    const float4 v = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    for (size_t j = threadIdx.x; j < n4; j += 512) {
        if (NT) __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4 *>(env) + j);
        else env[j] = v;
    }
}


// wave w owns the CONTIGUOUS run of rows [w*N/8, (w+1)*N/8): per row the same three-store pattern
__global__ __launch_bounds__(512) void k_run_rows(float *out, int N, int row)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, per = N / 8;
    float *env = out + (size_t)blockIdx.x * N * row;
    for (int i = wave * per; i < (wave + 1) * per; ++i) {
        float *dst = env + (size_t)i * row;
        const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
        float *dst_al = dst - mis;
        const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
        const float4 v = make_float4((float)i, (float)lane, 1.0f, 2.0f);
        for (uint32_t j = j_lo + lane; j < j_hi; j += 64) reinterpret_cast<float4 *>(dst_al)[j] = v;
        const uint32_t hd = 4 * j_lo - mis, tl = mis + row - 4 * j_hi;
        if ((uint32_t)lane < hd) dst_al[mis + lane] = 3.0f;
        else if ((uint32_t)lane - hd < tl) dst_al[4 * j_hi + (lane - hd)] = 3.0f;
    }
}

// the same run, but the < 16-byte remainder of a row is carried into the next row's first store:
// aligned 16-byte stores only, scalar edges only at the two ends of the run
__global__ __launch_bounds__(512) void k_run_carry(float *out, int N, int row)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, per = N / 8;
    float *run = out + ((size_t)blockIdx.x * N + (size_t)wave * per) * row;
    const uint32_t mis = (uint32_t)(((uintptr_t)run >> 2) & 3);
    float *al = run - mis;                       // 16-byte aligned window start
    const uint32_t total = mis + (uint32_t)per * row;
    uint32_t done4 = (mis + 3) >> 2;             // float4s flushed so far (head handled below)
    if ((uint32_t)lane < 4 * done4 - mis && mis) al[mis + lane] = 3.0f;
    const float4 v = make_float4(1.0f, (float)lane, 1.0f, 2.0f);
    for (int k = 0; k < per; ++k) {
        const uint32_t end4 = (mis + (uint32_t)(k + 1) * row) >> 2;
        for (uint32_t j = done4 + lane; j < end4; j += 64) reinterpret_cast<float4 *>(al)[j] = v;
        done4 = end4;
    }
    if (4 * done4 + lane < total) al[4 * done4 + lane] = 3.0f;
}

// wave-run rows with `work` dependent FMAs per lane before each row's stores: does the write stream
// overlap with VALU work of the same waves, or do the two add up?  (store=false: the compute alone)
template <bool STORE>
__global__ __launch_bounds__(512) void k_run_compute(float *out, int N, int row, int work, float seed)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, per = N / 8;
    float *env = out + (size_t)blockIdx.x * N * row;
    float a = seed + lane, b = seed * 0.5f, c = seed + 2, d = seed + 3;
    for (int i = wave * per; i < (wave + 1) * per; ++i) {
        for (int k = 0; k < work; ++k) {
            a = fmaf(a, b, 1.0f); c = fmaf(c, b, 1.0f); d = fmaf(d, b, 1.0f); a = fmaf(a, c, d);
        }
        float *dst = env + (size_t)i * row;
        const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
        float *dst_al = dst - mis;
        const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
        const float4 v = make_float4(a, c, d, 2.0f);
        if (STORE) {
            for (uint32_t j = j_lo + lane; j < j_hi; j += 64) reinterpret_cast<float4 *>(dst_al)[j] = v;
            const uint32_t hd = 4 * j_lo - mis, tl = mis + row - 4 * j_hi;
            if ((uint32_t)lane < hd) dst_al[mis + lane] = a;
            else if ((uint32_t)lane - hd < tl) dst_al[4 * j_hi + (lane - hd)] = a;
        }
    }
    if (a + c + d == 12345.678f) out[0] = a;
}

// MODE 0: every wave computes then stores its rows.  MODE 1: same + s_waitcnt vmcnt(0) after each row.
// MODE 2: wave specialisation - even waves only compute (two rows' worth), odd waves only store (two rows).
template <int MODE>
__global__ __launch_bounds__(512) void k_overlap(float *out, int N, int row, int work, float seed)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, per = N / 8;
    float *env = out + (size_t)blockIdx.x * N * row;
    float a = seed + lane, b = seed * 0.5f, c = seed + 2, d = seed + 3;
    const bool do_compute = MODE != 2 || (wave & 1) == 0, do_store = MODE != 2 || (wave & 1) == 1;
    const int reps = MODE == 2 ? 2 : 1;
    const int first = MODE == 2 ? (wave >> 1) * 2 * per : wave * per;
    for (int i = first; i < first + reps * per; ++i) {
        if (do_compute)
            for (int k = 0; k < work; ++k) {
                a = fmaf(a, b, 1.0f); c = fmaf(c, b, 1.0f); d = fmaf(d, b, 1.0f); a = fmaf(a, c, d);
            }
        if (do_store) {
            float *dst = env + (size_t)i * row;
            const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
            float *dst_al = dst - mis;
            const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
            const float4 v = MODE == 3 ? make_float4(b, seed, 1.0f, 2.0f) : make_float4(a, c, d, 2.0f);
            const uint32_t ja = min(j_lo + (uint32_t)lane, j_hi - 1), jb = min(j_lo + 64u + (uint32_t)lane, j_hi - 1);
            reinterpret_cast<float4 *>(dst_al)[ja] = v;
            reinterpret_cast<float4 *>(dst_al)[jb] = v;
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (a + c + d == 12345.678f) out[0] = a;
}

// Roles by SIMD: waves that landed on SIMD `store_simd` only store (all N rows), the others only compute
// (all N rows' worth of FMAs); work is handed out through LDS counters so any wave placement is balanced.
// store_simd < 0: every wave takes both kinds of work (control).
__global__ __launch_bounds__(512) void k_simd_split(float *out, int N, int row, int work, float seed, int store_simd)
{
    __shared__ int next_store, next_comp;
    const int lane = threadIdx.x & 63;
    const int simd = (__builtin_amdgcn_s_getreg(4 | (31 << 11)) >> 4) & 3;
    if (threadIdx.x == 0) { next_store = 0; next_comp = 0; }
    __syncthreads();
    float *env = out + (size_t)blockIdx.x * N * row;
    float a = seed + lane, b = seed * 0.5f, c = seed + 2, d = seed + 3;
    const bool storer = store_simd < 0 || simd == store_simd, computer = store_simd < 0 || simd != store_simd;
    if (computer)
        for (;;) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&next_comp, 1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= N) break;
            for (int k = 0; k < work; ++k) {
                a = fmaf(a, b, 1.0f); c = fmaf(c, b, 1.0f); d = fmaf(d, b, 1.0f); a = fmaf(a, c, d);
            }
        }
    if (storer)
        for (;;) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&next_store, 1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= N) break;
            float *dst = env + (size_t)i * row;
            const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
            float *dst_al = dst - mis;
            const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2;
            const float4 v = make_float4(b, seed, 1.0f, 2.0f);
            const uint32_t ja = min(j_lo + (uint32_t)lane, j_hi - 1), jb = min(j_lo + 64u + (uint32_t)lane, j_hi - 1);
            reinterpret_cast<float4 *>(dst_al)[ja] = v;
            reinterpret_cast<float4 *>(dst_al)[jb] = v;
        }
    if (a + c + d == 12345.678f) out[0] = a;
}

// Gather cost on the CU's vector-memory path: each wave issues `n` pairs of (8-byte, 4-byte) gathers
// shaped like one ant's perception (a rotated 7x7 patch of a [256][256] grid; lane = cell), with no
// other work.  hot: all waves of the chip read the same 768 KiB (L2 hits).
__global__ __launch_bounds__(512) void k_gather(const float2 *ph, const float *food, float *sink, int n, int hot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = hot ? 0 : (size_t)blockIdx.x * 65536;
    const int a = lane / 7 - 3, b = lane % 7 - 3;
    float acc = 0.0f;
    uint32_t rng = blockIdx.x * 977u + wave * 131u + 7u;
    for (int i = 0; i < n; i += 2) {
        float2 p[2]; float f[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            rng = rng * 1664525u + 1013904223u;
            const int cx = (rng >> 8) & 255, cy = (rng >> 16) & 255;
            const float th = (float)(rng & 255) * 0.0245f, ct = __cosf(th), st = __sinf(th);
            const int ix = (cx + (int)rintf(1.1f * (ct * b - st * a))) & 255, iy = (cy + (int)rintf(1.1f * (st * b + ct * a))) & 255;
            const uint32_t cell = (uint32_t)(ix * 256 + iy);
            p[u] = ph[base + cell];
            f[u] = food[base + cell];
        }
        acc += p[0].x + p[0].y + f[0] + p[1].x + p[1].y + f[1];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// the same perception-shaped gathers from ONE interleaved array {p0, p1, food, pad} (16 bytes per cell): a
// single 16-byte gather per ant instead of an 8-byte and a 4-byte one
__global__ __launch_bounds__(512) void k_gather16(const float4 *grid, float *sink, int n, int hot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = hot ? 0 : (size_t)blockIdx.x * 65536;
    const int a = lane / 7 - 3, b = lane % 7 - 3;
    float acc = 0.0f;
    uint32_t rng = blockIdx.x * 977u + wave * 131u + 7u;
    for (int i = 0; i < n; i += 2) {
        float4 p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            rng = rng * 1664525u + 1013904223u;
            const int cx = (rng >> 8) & 255, cy = (rng >> 16) & 255;
            const float th = (float)(rng & 255) * 0.0245f, ct = __cosf(th), st = __sinf(th);
            const int ix = (cx + (int)rintf(1.1f * (ct * b - st * a))) & 255, iy = (cy + (int)rintf(1.1f * (st * b + ct * a))) & 255;
            p[u] = grid[base + (uint32_t)(ix * 256 + iy)];
        }
        acc += p[0].x + p[0].y + p[0].z + p[1].x + p[1].y + p[1].z;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// grid-stride streaming fill over the whole tensor (what a memset-like kernel does)
__global__ __launch_bounds__(256) void k_fill(float4 *out, size_t n4)
{
    const float4 v = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (size_t)gridDim.x * 256) out[j] = v;
}

template <class F>
static double time_ms(F launch, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char **argv)
{
    const int E = argc > 1 ? atoi(argv[1]) : 1024, N = 512;
    const int rows[] = {343, 344, 352, 294};
    float *buf;
    const size_t cap = (size_t)E * N * 352 * 4 + 256;
    CK(hipMalloc(&buf, cap));
    for (int r : rows) {
        const double gb = (double)E * N * r * 4 / 1e9;
        double t;
        t = time_ms([&] { k_rows<false><<<E, 512>>>(buf, N, r); }, 50);
        printf("row=%3d floats  wave-per-row          %.4f ms  %.2f TB/s\n", r, t, gb / t);
        t = time_ms([&] { k_rows<true><<<E, 512>>>(buf, N, r); }, 50);
        printf("row=%3d floats  wave-per-row nt       %.4f ms  %.2f TB/s\n", r, t, gb / t);
        t = time_ms([&] { k_rows<false><<<E, 512>>>(buf + 1, N, r); }, 50);
        printf("row=%3d floats  wave-per-row base+4B  %.4f ms  %.2f TB/s\n", r, t, gb / t);
        t = time_ms([&] { k_run_rows<<<E, 512>>>(buf, N, r); }, 50);
        printf("row=%3d floats  wave-run, 3 stores    %.4f ms  %.2f TB/s\n", r, t, gb / t);
        t = time_ms([&] { k_run_carry<<<E, 512>>>(buf, N, r); }, 50);
        printf("row=%3d floats  wave-run, carry       %.4f ms  %.2f TB/s\n", r, t, gb / t);
        t = time_ms([&] { k_run_carry<<<E, 512>>>(buf + 1, N, r); }, 50);
        printf("row=%3d floats  wave-run, carry +4B   %.4f ms  %.2f TB/s\n", r, t, gb / t);
        if (r % 4 == 0) {
            t = time_ms([&] { k_block<false><<<E, 512>>>(buf, N, r); }, 50);
            printf("row=%3d floats  block-contiguous      %.4f ms  %.2f TB/s\n", r, t, gb / t);
            t = time_ms([&] { k_block<true><<<E, 512>>>(buf, N, r); }, 50);
            printf("row=%3d floats  block-contiguous nt   %.4f ms  %.2f TB/s\n", r, t, gb / t);
        }
        const size_t n4 = (size_t)E * N * r / 4;
        t = time_ms([&] { k_fill<<<2048, 256>>>(reinterpret_cast<float4 *>(buf), n4); }, 50);
        printf("row=%3d floats  grid-stride fill      %.4f ms  %.2f TB/s\n", r, t, gb / t);
    }
    for (int work : {0, 16, 32, 48, 64, 96, 128}) {
        const int r = 343;
        const double gb = (double)E * N * r * 4 / 1e9;
        const double tc = time_ms([&] { k_run_compute<false><<<E, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double ts = time_ms([&] { k_run_compute<true><<<E, 512>>>(buf, N, r, work, 1.5f); }, 20);
        printf("work=%3d  compute alone %.4f ms   compute+stores %.4f ms (%.2f TB/s)\n", work, tc, ts, gb / ts);
    }
    for (int e : {256, 512, 1024}) {
        const int r = 343, work = 96;
        const double tc = time_ms([&] { k_run_compute<false><<<e, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double t0 = time_ms([&] { k_overlap<0><<<e, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double t1 = time_ms([&] { k_overlap<1><<<e, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double t2 = time_ms([&] { k_overlap<2><<<e, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double t3 = time_ms([&] { k_overlap<3><<<e, 512>>>(buf, N, r, work, 1.5f); }, 20);
        const double tw = time_ms([&] { k_overlap<0><<<e, 512>>>(buf, N, r, 0, 1.5f); }, 20);
        printf("E=%4d work=%d: compute %.4f  stores %.4f  both %.4f  both+vmcnt0 %.4f  specialised waves %.4f  const-data %.4f ms\n", e, work, tc, tw, t0, t1, t2, t3);
    }
    for (int e : {256, 1024}) {
        const int r = 343, work = 96;
        const double ta = time_ms([&] { k_simd_split<<<e, 512>>>(buf, N, r, work, 1.5f, -1); }, 20);
        const double tb = time_ms([&] { k_simd_split<<<e, 512>>>(buf, N, r, work, 1.5f, 3); }, 20);
        const double tc0 = time_ms([&] { k_simd_split<<<e, 512>>>(buf, N, r, 0, 1.5f, -1); }, 20);
        const double tc3 = time_ms([&] { k_simd_split<<<e, 512>>>(buf, N, r, 0, 1.5f, 3); }, 20);
        printf("E=%4d simd-split: all waves both roles %.4f ms   SIMD3 stores / SIMD0-2 compute %.4f ms   (stores only: %.4f / on SIMD3 only %.4f)\n", e, ta, tb, tc0, tc3);
    }
    {
        float2 *ph; float *fd;
        CK(hipMalloc(&ph, (size_t)1024 * 65536 * 8)); CK(hipMalloc(&fd, (size_t)1024 * 65536 * 4));
        CK(hipMemset(ph, 0, (size_t)1024 * 65536 * 8)); CK(hipMemset(fd, 0, (size_t)1024 * 65536 * 4));
        for (int e : {256, 1024})
            for (int hot : {1, 0}) {
                const double t = time_ms([&] { k_gather<<<e, 512>>>(ph, fd, buf, 64, hot); }, 20);
                // per CU: e/256 workgroups x 8 waves x 64 ant-gathers
                const double per_ant_cycles = t * 1e-3 * 2.4e9 / ((e / 256.0) * 8 * 64);
                printf("gathers E=%4d %s: %.4f ms for 64 ants per wave = %.0f cycles per ant per CU (incl. address math)\n", e,
                       hot ? "L2-hot " : "per-env", t, per_ant_cycles);
            }
        float4 *g16;
        CK(hipMalloc(&g16, (size_t)1024 * 65536 * 16)); CK(hipMemset(g16, 0, (size_t)1024 * 65536 * 16));
        for (int e : {256, 1024})
            for (int hot : {1, 0}) {
                const double t = time_ms([&] { k_gather16<<<e, 512>>>(g16, buf, 64, hot); }, 20);
                printf("gathers (one 16-byte array) E=%4d %s: %.4f ms = %.0f cycles per ant per CU\n", e, hot ? "L2-hot " : "per-env", t,
                       t * 1e-3 * 2.4e9 / ((e / 256.0) * 8 * 64));
            }
        CK(hipFree(g16));
        CK(hipFree(ph)); CK(hipFree(fd));
    }
    CK(hipFree(buf));
    return 0;
}
