// obs_stream_probe.hip — the write stream of k_perceive on its own, with knobs, to find what bounds it:
// E*N rows of ROW floats (c3: 1024 x 512 x 343 = 0.72 GB).  One wave per contiguous run of RUN rows, WPB
// waves per workgroup, GROUP rows per flush (whole 128-byte lines, the partial line carried to the next flush —
// here simply: every flush writes `GROUP*ROW` floats rounded to whole lines, which is the same traffic).
//   --lds KB      dummy dynamic LDS per workgroup: limits resident waves per CU (occupancy knob)
//   --work W      W x 8 dependent-chain FMAs per lane per row (VALU beside the stores)
//   --gather G    G = 1: one 16-byte gather per lane per row from a 1 GiB table (random cells of the row's
//                 "environment" = a 1 MiB window), G = 2: from a 16 KiB window (L1/L2 hits)
//   --nt 0|1      plain or nt stores
// Prints ms and TB/s of written bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st4(float *p, const vf4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(p));
    else *reinterpret_cast<vf4 *>(p) = v;
}

template <bool NT, int GATHER>
__global__ void k_stream(float *__restrict__ out, const float4 *__restrict__ table, const int N, const int row, const int run,
                         const int group, const int work, const float seed, const int nseg, const int wpb, const int dup, const int map, const int blockflush)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int e = map == 0 ? b / nseg : ((b >> 3) / nseg) * 8 + (b & 7), seg = map == 0 ? b % nseg : (b >> 3) % nseg;
    if (blockflush) { // the workgroup's whole segment as ONE contiguous stream: its waves interleave at 1 KiB
        float *envp = out + (size_t)e * N * row;
        const int s_begin = min(seg * wpb * run, N), s_end = min(s_begin + wpb * run, N);
        const size_t lo = ((size_t)s_begin * row) & ~(size_t)31, hi = (((size_t)s_end * row) + 31) & ~(size_t)31;
        const vf4 v = {seed, 1.0f, 2.0f, 3.0f};
        for (size_t j = lo + 4 * (size_t)threadIdx.x; j < hi; j += 4 * (size_t)blockDim.x) st4<NT>(envp + j, v);
        return;
    }
    const int i_begin = min((seg * wpb + wave) * run, N), i_end = min(i_begin + run, N);
    if (smem[0] == 77 && seed == 123.0f) out[0] = 1.0f; // keeps the LDS allocation
    float a = seed + lane, c = seed + 2, d = seed + 3, k = seed * 0.25f;
    uint32_t rng = (uint32_t)(b * 64 + wave) * 2654435761u + lane * 40503u;
    float *env = out + (size_t)e * N * row;
    const float4 *cells = table + (size_t)e * 65536; // 1 MiB of 16-byte cells per environment
    // the run's byte range rounded outwards to whole lines is written line by line, `group` rows' worth per flush
    size_t pos = ((size_t)i_begin * row) & ~(size_t)31;           // float index, 128-byte aligned
    const size_t end = (((size_t)i_end * row) + 31) & ~(size_t)31;
    const size_t per_flush = ((size_t)group * row + 31) & ~(size_t)31;
    float4 g = make_float4(0, 0, 0, 0);
    for (int i = i_begin; i < i_end; i += group) {
        if (GATHER) { // one gather per row of the group, all issued before use
            for (int u = 0; u < group; ++u) {
                rng = rng * 1664525u + 1013904223u;
                const uint32_t cell = GATHER == 2 ? (rng >> 8) & 1023u : (rng >> 8) & 65535u;
                const float4 t = cells[cell];
                g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
            }
        }
        for (int w = 0; w < work * group; ++w) {
            a = fmaf(a, k, 1.0f); c = fmaf(c, k, 1.0f); d = fmaf(d, k, 1.0f); a = fmaf(a, k, 0.5f);
            c = fmaf(c, k, 0.5f); d = fmaf(d, k, 0.5f); a = fmaf(a, k, 0.25f); c = fmaf(c, k, 0.25f);
        }
        const vf4 v = {a + g.x, c + g.y, d + g.z, g.w};
        const size_t stop = min(pos + per_flush, end);
        if (dup == 0) {
            for (size_t j = pos + 4 * (size_t)lane; j < stop; j += 256) st4<NT>(env + j, v);
        } else { // a fixed number of store instructions; a lane with nothing left repeats its previous piece (dup 1)
                 // or the flush's last piece (dup 2): what k_perceive's unconditional stores do
            const int ninst = (int)((per_flush + 255) / 256);
            for (int t = 0; t < ninst; ++t) {
                size_t j = pos + 4 * (size_t)lane + 256 * (size_t)t;
                if (j >= stop) j = dup == 1 ? (j >= pos + 256 ? j - 256 : stop - 4) : stop - 4;
                if (j >= stop) j = stop - 4;
                st4<NT>(env + j, v);
            }
        }
        pos = stop;
    }
    if (a + c + d == 12345.678f) out[1] = a;
}

int main(int argc, char **argv)
{
    int E = 1024, N = 512, row = 343, run = 32, group = 2, work = 0, gather = 0, nt = 1, wpb = 4, lds_kb = 0, reps = 10, dup = 0, map = 1, blockflush = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const char *k = argv[i];
        const int v = atoi(argv[i + 1]);
        if (!strcmp(k, "--envs")) E = v; else if (!strcmp(k, "--run")) run = v; else if (!strcmp(k, "--group")) group = v;
        else if (!strcmp(k, "--work")) work = v; else if (!strcmp(k, "--gather")) gather = v; else if (!strcmp(k, "--nt")) nt = v;
        else if (!strcmp(k, "--wpb")) wpb = v; else if (!strcmp(k, "--lds")) lds_kb = v; else if (!strcmp(k, "--row")) row = v; else if (!strcmp(k, "--dup")) dup = v; else if (!strcmp(k, "--map")) map = v; else if (!strcmp(k, "--blockflush")) blockflush = v;
    }
    const size_t floats = (size_t)E * N * row + 64;
    float *out;
    float4 *table;
    CK(hipMalloc(&out, floats * 4));
    CK(hipMalloc(&table, (size_t)E * 65536 * 16));
    CK(hipMemset(out, 0, floats * 4));
    CK(hipMemset(table, 0, (size_t)E * 65536 * 16));
    const int nseg = (N + run * wpb - 1) / (run * wpb);
    const dim3 grid(E * nseg), block(64 * wpb);
    const size_t lds = (size_t)lds_kb * 1024;
#define GO(NTV, GV) { if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_stream<NTV, GV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                      hipLaunchKernelGGL((k_stream<NTV, GV>), grid, block, lds, 0, out, table, N, row, run, group, work, 1.0f, nseg, wpb, dup, map, blockflush); }
    auto launch = [&]() {
        if (nt) { if (gather == 0) GO(true, 0) else if (gather == 1) GO(true, 1) else GO(true, 2) }
        else { if (gather == 0) GO(false, 0) else if (gather == 1) GO(false, 1) else GO(false, 2) }
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
    ms /= reps;
    printf("envs=%d run=%d group=%d wpb=%d lds=%dK work=%d gather=%d nt=%d dup=%d map=%d blockflush=%d : %.4f ms  %.2f TB/s\n", E, run, group, wpb, lds_kb, work,
           gather, nt, dup, map, blockflush, ms, (double)E * N * row * 4 / ms / 1e9);
    return 0;
}
