// antsrl_sweep.hip — pheromone sweeps: k_sweep0 (radius 0, streaming), k_sweep_march (radius 1..3,
// register-marching stencil, separable form for rank-1 filters), k_sweep_tiled (float64 cross-check),
// the wall clear / re-basing passes of the scaled representation, launchers.
#include "antsrl_util.h"

// ===================================================================================
// Pheromone sweep — Walls zeroing (walls.py:30) + Pheromone.update (pheromone.py:43-45)
// + the whole-grid clip of add_pheromones (pheromone.py:40-41; min is idempotent and the
// deposit is non-negative, so clipping before the deposit and again at the deposit gives the
// same grid).
// ===================================================================================
// Radius 0 (the shipped DIFFUSE_FACTOR = 0 filter): out = thresh(in * f0), pure streaming,
// 16 bytes per lane per access.  The product is formed in float64 so that the coefficient
// (0.999) carries no float32 rounding bias across thousands of steps.
#define SW0_UNROLL 4
template <int C>
__global__ void __launch_bounds__(256)
k_sweep0(const KP p, const float *__restrict__ in, float *__restrict__ out)
{
    // grid = (ceil(per_env / (256*SW0_UNROLL)), E): no per-thread division by the env size
    const uint32_t per_env = (uint32_t)((size_t)p.W * p.H * C / 4); // float4 per env
    const size_t e = blockIdx.y;
    const uint32_t *walls = p.s.walls_bits + e * p.words;
    const float4 *src = reinterpret_cast<const float4 *>(in) + e * per_env;
    float4 *dst = reinterpret_cast<float4 *>(out) + e * per_env;
    const double f0 = p.filter[0], thr = p.threshold;
    const float mx = (float)p.max_val;
    const bool clip = p.has_max_val && p.N > 0;
    const uint32_t v0 = blockIdx.x * (256 * SW0_UNROLL) + threadIdx.x;
    float4 a[SW0_UNROLL];
#pragma unroll
    for (int u = 0; u < SW0_UNROLL; ++u) {
        const uint32_t v = v0 + u * 256;
        if (v < per_env) a[u] = src[v];
    }
#pragma unroll
    for (int u = 0; u < SW0_UNROLL; ++u) {
        const uint32_t v = v0 + u * 256;
        if (v >= per_env) continue;
        float r[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t cell = (v * 4 + j) / C;
            const double o = (double)r[j] * f0;
            float f = (o < thr) ? 0.0f : (float)o;
            if (test_bit(walls, cell)) f = 0.0f;
            if (clip) f = fminf(f, mx);
            r[j] = f;
        }
        store_stream(dst + v, make_float4(r[0], r[1], r[2], r[3]));
    }
}

// scalar fallback for grids whose float count per env is not a multiple of 4
template <int C>
__global__ void __launch_bounds__(256)
k_sweep0_scalar(const KP p, const float *__restrict__ in, float *__restrict__ out, const size_t n)
{
    const size_t per_env = (size_t)p.W * p.H * C;
    const double f0 = p.filter[0], thr = p.threshold;
    const bool clip = p.has_max_val && p.N > 0;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v / per_env;
        const uint32_t cell = (uint32_t)((v - e * per_env) / C);
        const double o = (double)in[v] * f0;
        float f = (o < thr) ? 0.0f : (float)o;
        if (test_bit(p.s.walls_bits + e * p.words, cell)) f = 0.0f;
        if (clip) f = fminf(f, (float)p.max_val);
        out[v] = f;
    }
}

// Radius 1..3: LDS-tiled 2-D convolution with zero-fill boundary,
// scipy.signal.convolve2d(phero, F, 'same', 'fill', 0)  (pheromone.py:44):
//   out[x,y] = sum_{a,b} F[a,b] * in[x-a+r, y-b+r]   (true convolution: kernel flipped)
// The wall mask is applied while the tile is staged (walls.py:30 zeroes the INPUT of the
// convolution, so a wall cell still receives its neighbours' diffusion).
#define SW_TX 16
#define SW_TY 64
template <int C>
__global__ void __launch_bounds__(256)
k_sweep_tiled(const KP p, const float *__restrict__ in, float *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *tile = (float *)smem;
    const int fr = p.filter_radius, fs = 2 * fr + 1;
    const int LX = SW_TX + 2 * fr, LY = SW_TY + 2 * fr;
    const int e = blockIdx.z, x0 = blockIdx.y * SW_TX, y0 = blockIdx.x * SW_TY;
    const int W = p.W, H = p.H;
    const size_t G = (size_t)W * H;
    const float *src = in + (size_t)e * G * C;
    float *dst = out + (size_t)e * G * C;
    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    for (int t = threadIdx.x; t < LX * LY; t += blockDim.x) {
        const int lx = t / LY, ly = t - lx * LY;
        const int gx = x0 + lx - fr, gy = y0 + ly - fr;
        const bool inside = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const uint32_t cell = inside ? (uint32_t)(gx * H + gy) : 0u;
        const bool live = inside && !test_bit(walls, cell);
#pragma unroll
        for (int c = 0; c < C; ++c) tile[(size_t)t * C + c] = live ? src[(size_t)cell * C + c] : 0.0f;
    }
    __syncthreads();
    const bool clip = p.has_max_val && p.N > 0;
    for (int t = threadIdx.x; t < SW_TX * SW_TY; t += blockDim.x) {
        const int lx = t / SW_TY, ly = t - lx * SW_TY;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= W || gy >= H) continue;
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.0;
        for (int a = 0; a < fs; ++a)
            for (int b = 0; b < fs; ++b) {
                const double f = p.filter[a * fs + b];
                const float *tp = tile + ((size_t)(lx - a + 2 * fr) * LY + (ly - b + 2 * fr)) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += f * (double)tp[c];
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float f = (acc[c] < p.threshold) ? 0.0f : (float)acc[c];
            if (clip) f = fminf(f, (float)p.max_val);
            dst[((size_t)gx * H + gy) * C + c] = f;
        }
    }
}

// Scaled mode helpers (rare, full-grid): zero the wall cells of the grid (first update after a
// reset that supplied an initial pheromone grid) and re-base the units when f0^S gets tiny.
// (`buf`: the pheromone buffer — 0 with scaled units; the current one when antsrl_update_phase runs Walls.update's
//  `phero[map] = 0`, walls.py:30, as a step of its own under an explicit sweep)
__global__ void __launch_bounds__(256) k_phero_wall_clear(const KP p, const int buf)
{
    const size_t G = (size_t)p.W * p.H, n = (size_t)p.E * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / G, g = i - e * G;
        if (test_bit(p.s.walls_bits + e * p.words, (uint32_t)g))
            for (int c = 0; c < p.C; ++c) p.s.phero[buf][(e * G + prec_cell(p, (uint32_t)g)) * p.ps + c] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) k_phero_renorm(const KP p)
{
    // u := materialised value (units of f0^0); the host then restarts S at 0.
    const size_t n = (size_t)p.E * p.W * p.H * p.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = (i / p.C) * p.ps + (i % p.C); // (cell, channel) -> position under the cell stride
        const double v = (double)p.s.phero[0][j] * p.g_now;
        p.s.phero[0][j] = v < p.threshold ? 0.0f : (float)v;
    }
}

// Radius 1..3, main kernel: register-marching stencil.  One wave owns a strip of 64-2R output
// columns (lane <-> column y, R halo lanes each side) of one environment and marches down the
// x rows in blocks of S = 2R+1 rows: every input row is read ONCE, coalesced, straight into
// registers (all S rows of a block are in flight together, no branch between them); its 2R
// y-neighbours come from the other lanes of the wave (__shfl); the 2S-1 output rows a block
// touches are running accumulators in registers.  No LDS, no re-reads except the 2R halo columns.
//   out[x,y] = sum_{a,b} F[a,b] * in[x-a+R, y-b+R]      (convolve2d 'same', zero fill)
// Arithmetic: fp32 FMAs with the taps split hi+lo (see KP::ftap) — unbiased to ~1e-15 per step.
// Loop order b -> a -> row keeps only one tap column (2S scalars) live at a time.
// lane i <- lane i - 1 / lane i + 1 across the whole wave (zero shifted in at the ends)
__device__ __forceinline__ float wave_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_shl1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// SEP: the filter is rank-1, F[a][b] = u[a] v[b] (KP::fsep_u / fsep_v): each input row is first
// convolved across the lanes with v (S shuffles), the result feeds the S running output rows with u —
// 4 S FMAs per input value instead of 2 S^2.
template <int C, int R, bool SEP>
__global__ void __launch_bounds__(256)
k_sweep_march(const KP p, const float *__restrict__ in, float *__restrict__ out, const int seg_rows, const int nstrips, const int nsegs)
{
    constexpr int S = 2 * R + 1, OUTW = 64 - 2 * R, NA = 2 * S - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y, W = p.W, H = p.H;
    // one wave per (column strip, row segment) pair, an environment's pairs packed densely into its workgroups
    const int unit = blockIdx.x * 4 + wave, strip = unit % nstrips, segi = unit / nstrips;
    // The S*S taps {hi, lo} sit in LDS and are read back (uniform address = broadcast) right where
    // they are used: ~100 wave-uniform scalars do not fit the SGPR file — as kernel arguments the
    // compiler hoists them out of the march and spills them through v_writelane/v_readlane.
    __shared__ float taps[2 * ANTSRL_MAX_FILTER_TAPS];
    if (!SEP && threadIdx.x < 2 * S * S) taps[threadIdx.x] = p.ftap[threadIdx.x];
    if (SEP && threadIdx.x < 2 * S) {
        taps[threadIdx.x] = p.fsep_u[threadIdx.x];
        taps[2 * S + threadIdx.x] = p.fsep_v[threadIdx.x];
    }
    __syncthreads();
    if (segi >= nsegs) return; // whole wave (no further barriers)
    const int y = strip * OUTW - R + lane;
    const bool col_in = y >= 0 && y < H;
    const bool col_out = lane >= R && lane < 64 - R && y < H;
    const int yc = col_in ? y : 0;
    const float colmask = col_in ? 1.0f : 0.0f;
    const size_t G = (size_t)W * H;
    const float *src = in + (size_t)e * G * C;
    float *dst = out + (size_t)e * G * C;
    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    const bool clip = p.has_max_val && p.N > 0;
    const float thr = (float)p.threshold, mx = (float)p.max_val;
    // acc[j] accumulates output row x = xi0 - R + j of the current block
    float acc[NA][C];
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[j][c] = 0.0f;

    // This wave produces output rows [x_lo, x_hi): it marches input rows x_lo - R .. x_hi - 1 + R
    // (rows outside the grid count as zero), i.e. 2R rows of overlap with its x-neighbour segment.
    const int x_lo = segi * seg_rows, x_hi = min(x_lo + seg_rows, W);
    // one block of S rows is always in flight ahead of the block being accumulated
    float nv[S][C];
    uint32_t nword[S], ncell[S];
#define MARCH_LOAD(XI0)                                                                              \
    {                                                                                                \
        _Pragma("unroll") for (int s = 0; s < S; ++s)                                                \
        {                                                                                            \
            /* clamped to the grid (masked to zero below) and to the segment's last halo row: the prefetch of  \
               the last turn must not pull in rows nobody needs (round 3: it did, +25 % of the reads) */    \
            const int xc = min(max(min((XI0) + s, x_hi + R - 1), 0), W - 1);                        \
            ncell[s] = (uint32_t)(xc * H + yc);                                                      \
            nword[s] = walls[ncell[s] >> 5];                                                         \
        }                                                                                            \
        _Pragma("unroll") for (int s = 0; s < S; ++s)                                                \
        {                                                                                            \
            if (C == 2) {                                                                            \
                const float2 t = *reinterpret_cast<const float2 *>(src + (size_t)ncell[s] * 2);      \
                nv[s][0] = t.x; nv[s][C - 1] = t.y;                                                  \
            } else {                                                                                 \
                _Pragma("unroll") for (int c = 0; c < C; ++c) nv[s][c] = src[(size_t)ncell[s] * C + c]; \
            }                                                                                        \
        }                                                                                            \
    }
    MARCH_LOAD(x_lo - R)
    for (int xi0 = x_lo - R; xi0 < x_hi + R; xi0 += S) {
        float v[S][C];
        uint32_t wword[S];
        uint32_t cellv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            cellv[s] = ncell[s]; wword[s] = nword[s];
#pragma unroll
            for (int c = 0; c < C; ++c) v[s][c] = nv[s][c];
        }
        MARCH_LOAD(xi0 + S) // prefetch (clamped: past the segment's end a re-read of its last row)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // zero fill outside the grid, and walls.py:30 zeroes the INPUT of the convolution.
            // Arithmetic masks (0/1 factors) rather than selects: lane-mask booleans would each
            // occupy an SGPR pair and spill.
            const float keep = colmask * (float)(1u - ((wword[s] >> (cellv[s] & 31)) & 1u)) *
                               ((xi0 + s >= 0 && xi0 + s < W) ? 1.0f : 0.0f);
#pragma unroll
            for (int c = 0; c < C; ++c) v[s][c] *= keep;
        }
        // ---- accumulate: input row s feeds output row j = s + a (x = xi0 + s + a - R).
        //      The tap-column loop is a REAL loop (not unrolled): only one column's 2S taps are live,
        //      so nothing tempts the compiler to hoist ~100 scalars out of the march and spill them.
        if (SEP) {
            float hrow[S][C]; // input row s convolved across the lanes with v
            // neighbours by whole-wave DPP shifts of one lane (v_mov_b32_dpp wave_shr:1 / wave_shl:1, a VALU
            // move) instead of ds_bpermute: the LDS pipe stays out of the inner loop
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float h = fmaf(taps[2 * S + 2 * R], v[s][c], 0.0f);
                    h = fmaf(taps[2 * S + 2 * R + 1], v[s][c], h);
                    float up = v[s][c], dn = v[s][c]; // up: value of lane - d, dn: value of lane + d
#pragma unroll
                    for (int d = 1; d <= R; ++d) {
                        up = wave_shr1(up);
                        dn = wave_shl1(dn);
                        h = fmaf(taps[2 * S + 2 * (R + d)], up, h);
                        h = fmaf(taps[2 * S + 2 * (R + d) + 1], up, h);
                        h = fmaf(taps[2 * S + 2 * (R - d)], dn, h);
                        h = fmaf(taps[2 * S + 2 * (R - d) + 1], dn, h);
                    }
                    hrow[s][c] = h;
                }
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float uh = taps[2 * a], ul = taps[2 * a + 1];
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        acc[s + a][c] = fmaf(uh, hrow[s][c], acc[s + a][c]);
                        acc[s + a][c] = fmaf(ul, hrow[s][c], acc[s + a][c]);
                    }
            }
        } else
#pragma unroll 1
        for (int b = 0; b < S; ++b) {
            float sh[S][C]; // sh[s] = in[xi0+s][y - b + R]
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int c = 0; c < C; ++c) sh[s][c] = __shfl(v[s][c], lane - (b - R));
            const float *tp = taps + 2 * b * S;
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float fh = tp[2 * a], fl = tp[2 * a + 1]; // LDS, uniform address: broadcast
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        acc[s + a][c] = fmaf(fh, sh[s][c], acc[s + a][c]);
                        acc[s + a][c] = fmaf(fl, sh[s][c], acc[s + a][c]);
                    }
            }
        }
        // ---- rows j = 0..S-1 are complete (last contributor: input row xi0 + j, tap row 0)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int x = xi0 - R + j;
            if (x >= x_lo && x < x_hi && col_out) {
                float r[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float f = acc[j][c] < thr ? 0.0f : acc[j][c]; // pheromone.py:45
                    if (clip) f = fminf(f, mx);
                    r[c] = f;
                }
                const size_t o = ((size_t)x * H + y) * C;
                if (C == 2) {
                    // streaming store (see store_stream): the next reader is a later kernel
                    typedef float vf2_t __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(vf2_t{r[0], r[C - 1]}, reinterpret_cast<vf2_t *>(dst + o));
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) dst[o + c] = r[c];
                }
            }
        }
        // carry the S-1 partial rows over to the next block
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[j][c] = (j + S < NA) ? acc[j + S][c] : 0.0f;
    }
#undef MARCH_LOAD
}


// Radius 1, two channels, even H — the reference's own diffusion mode (DIFFUSE_FACTOR > 0, pheromone.py:5-10:
// a 3x3 filter).  Same register march as k_sweep_march, with TWO columns per lane: every access is 16 bytes
// (cells y, y+1 x channels 0, 1), a wave covers exactly 128 output columns (H = 256: two strips, every lane
// busy; the one-column form needs 5 strips of 62 + 2 halo lanes), the y-neighbours of a lane's inner pair are
// its own registers, the outer ones come from the adjacent lane by a whole-wave DPP shift, and the two halo
// columns of the strip (127 cells apart) are one extra 8-byte load per row.
//   out[x,y] = sum_{a,b} F[a,b] * in[x-a+1, y-b+1]      (convolve2d 'same', zero fill; taps split hi + lo)
__global__ void __launch_bounds__(256)
k_sweep_r1x2(const KP p, const float *__restrict__ in, float *__restrict__ out, const int seg_rows, const int nstrips, const int nsegs)
{
    constexpr int S = 3, NA = 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y, W = p.W, H = p.H;
    // one wave per (column strip, row segment) pair, an environment's pairs packed densely into its workgroups
    const int unit = blockIdx.x * 4 + wave, strip = unit % nstrips, segi = unit / nstrips;
    __shared__ float taps[2 * 9];
    if (threadIdx.x < 18) taps[threadIdx.x] = p.ftap[threadIdx.x]; // [b][a]{hi, lo}
    __syncthreads();
    if (segi >= nsegs) return; // whole wave (no further barriers)
    const int y = strip * 128 + 2 * lane;          // this lane's columns y, y + 1 (H even: both in or both out)
    const bool col_in = y < H;
    const int yc = col_in ? y : 0;
    const float colmask = col_in ? 1.0f : 0.0f;
    // halo columns of the strip: lanes 0..31 fetch column strip*128 - 1, lanes 32..63 column strip*128 + 128
    const int yh = lane < 32 ? strip * 128 - 1 : strip * 128 + 128;
    const bool halo_in = yh >= 0 && yh < H;
    const int yhc = halo_in ? yh : 0;
    const float halomask = halo_in ? 1.0f : 0.0f;
    const size_t G = (size_t)W * H;
    const float *src = in + (size_t)e * G * 2;
    float *dst = out + (size_t)e * G * 2;
    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    const bool clip = p.has_max_val && p.N > 0;
    const float thr = (float)p.threshold, mx = (float)p.max_val;
    float acc[NA][4]; // output row x = xi0 - 1 + j: {col0 ch0, col0 ch1, col1 ch0, col1 ch1}
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[j][c] = 0.0f;
    const int x_lo = segi * seg_rows, x_hi = min(x_lo + seg_rows, W);
    float4 nv[S];
    float2 nh[S];
    uint32_t nword[S], nhword[S], ncell[S], nhcell[S];
#define R1_LOAD(XI0)                                                                                 \
    {                                                                                                \
        _Pragma("unroll") for (int s = 0; s < S; ++s)                                                \
        {                                                                                            \
            /* clamped to the grid (masked to zero below) and to the segment's last halo row */     \
            const int xc = min(max(min((XI0) + s, x_hi), 0), W - 1);                                \
            ncell[s] = (uint32_t)(xc * H + yc);                                                      \
            nhcell[s] = (uint32_t)(xc * H + yhc);                                                    \
            nword[s] = walls[ncell[s] >> 5];                                                         \
            nhword[s] = walls[nhcell[s] >> 5];                                                       \
        }                                                                                            \
        _Pragma("unroll") for (int s = 0; s < S; ++s)                                                \
        {                                                                                            \
            nv[s] = *reinterpret_cast<const float4 *>(src + (size_t)ncell[s] * 2);                   \
            nh[s] = *reinterpret_cast<const float2 *>(src + (size_t)nhcell[s] * 2);                  \
        }                                                                                            \
    }
    R1_LOAD(x_lo - 1)
    for (int xi0 = x_lo - 1; xi0 < x_hi + 1; xi0 += S) {
        float v[S][4], hv[S][2];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // zero fill outside the grid; walls.py:30 zeroes the INPUT of the convolution (arithmetic masks)
            const float rowm = (xi0 + s >= 0 && xi0 + s < W) ? 1.0f : 0.0f;
            const uint32_t sh0 = ncell[s] & 31u; // (even cell index: the pair's two bits sit in one word)
            const float k0 = colmask * rowm * (float)(1u - ((nword[s] >> sh0) & 1u));
            const float k1 = colmask * rowm * (float)(1u - ((nword[s] >> (sh0 + 1u)) & 1u));
            const float kh = halomask * rowm * (float)(1u - ((nhword[s] >> (nhcell[s] & 31u)) & 1u));
            v[s][0] = nv[s].x * k0; v[s][1] = nv[s].y * k0; v[s][2] = nv[s].z * k1; v[s][3] = nv[s].w * k1;
            hv[s][0] = nh[s].x * kh; hv[s][1] = nh[s].y * kh;
        }
        R1_LOAD(xi0 + S) // prefetch (clamped: past the segment's end a re-read of its last row)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // y-neighbours: in[x][y-1] of col0 = lane-1's col1 (lane 0: the left halo), in[x][y+2] of col1 = lane+1's
            // col0 (lane 63: the right halo, fetched by lanes 32..63)
            float lft[2], rgt[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float l = wave_shr1(v[s][2 + c]), r = wave_shl1(v[s][c]);
                const float hl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv[s][c]), 0));
                const float hr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv[s][c]), 63));
                lft[c] = lane == 0 ? hl : l;
                rgt[c] = lane == 63 ? hr : r;
            }
            // tap column b multiplies in[x][y - b + 1]: b = 0 -> right neighbour, 1 -> centre, 2 -> left neighbour
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float f0h = taps[2 * (0 * S + a)], f0l = taps[2 * (0 * S + a) + 1];
                const float f1h = taps[2 * (1 * S + a)], f1l = taps[2 * (1 * S + a) + 1];
                const float f2h = taps[2 * (2 * S + a)], f2l = taps[2 * (2 * S + a) + 1];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float t0 = acc[s + a][c], t1 = acc[s + a][2 + c];
                    // column y:     right = own col1, centre = own col0, left = lft
                    t0 = fmaf(f0h, v[s][2 + c], t0); t0 = fmaf(f0l, v[s][2 + c], t0);
                    t0 = fmaf(f1h, v[s][c], t0);     t0 = fmaf(f1l, v[s][c], t0);
                    t0 = fmaf(f2h, lft[c], t0);      t0 = fmaf(f2l, lft[c], t0);
                    // column y + 1: right = rgt, centre = own col1, left = own col0
                    t1 = fmaf(f0h, rgt[c], t1);      t1 = fmaf(f0l, rgt[c], t1);
                    t1 = fmaf(f1h, v[s][2 + c], t1); t1 = fmaf(f1l, v[s][2 + c], t1);
                    t1 = fmaf(f2h, v[s][c], t1);     t1 = fmaf(f2l, v[s][c], t1);
                    acc[s + a][c] = t0; acc[s + a][2 + c] = t1;
                }
            }
        }
        // rows j = 0..S-1 are complete
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int x = xi0 - 1 + j;
            if (x >= x_lo && x < x_hi && col_in) {
                float r[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float f = acc[j][c] < thr ? 0.0f : acc[j][c]; // pheromone.py:45
                    if (clip) f = fminf(f, mx);
                    r[c] = f;
                }
                store_stream(reinterpret_cast<float4 *>(dst + ((size_t)x * H + y) * 2), make_float4(r[0], r[1], r[2], r[3]));
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = (j + S < NA) ? acc[j + S][c] : 0.0f;
    }
#undef R1_LOAD
}

// Radius 2..3, two channels, even H, RANK-1 filter (c4's Gaussian): the separable march with TWO columns per lane
// (16-byte accesses).  Strips overlap by HL = ceil(R / 2) lanes on either side (the halo), so a wave covers 128 columns
// and produces 128 - 4 HL of them (R = 3: 120; the one-column form: 58 of 64).  Each input row is convolved across the
// lanes with v (neighbour columns: the lane's other column, and the adjacent lanes' columns by whole-wave DPP shifts of
// one and two lanes), the result feeds the S running output rows with u; nothing is kept per block but the accumulators
// and the prefetched rows.  out[x,y] = sum_a u[a] sum_b v[b] in[x-a+R, y-b+R]   (taps split hi + lo)
// STACK: a workgroup takes FOUR VERTICALLY STACKED segments of one column strip and its waves march AWAY FROM / TOWARDS
// the boundaries they share — waves 0 and 1 start at the boundary between segments 0 and 1 and march up (descending x) and
// down, waves 2 and 3 likewise around the boundary between segments 2 and 3; waves 1 and 2 then END at their common
// boundary at the same time.  The 2R halo rows two neighbouring segments both need are therefore requested by two waves
// of ONE workgroup within the same few hundred nanoseconds: one fetch from memory, the second an L1 / L2 hit.  With the flat
// packing (STACK = false: one wave per (strip, segment) pair in unit order) the second request came from another
// workgroup — usually on another XCD, i.e. another L2 — much later, and the 32-row segments re-read 2R / 32 = 19 % of the
// grid from memory (PMC, c4: 3.57 GB fetched for a 2.15 GB grid; profiles/r02/pmc_summary.md).
// A descending march is the ascending one in mirrored coordinates: the same ring of S running output rows, the row taps u
// taken in reverse order.  (The order in which an output row's contributions are added differs between the two directions:
// last-bit differences in float32, inside the 1e-5 bar the grid is held to.)
// HL: halo lanes per side (>= ceil(R / 2)).  With HL = 4 a wave writes 56 lanes x 16 bytes = 7 WHOLE 128-byte lines per row
// and every strip starts on a line; with the minimal halo (HL = 2 at R = 3: 120 columns = 960 bytes per row) every
// strip boundary cuts a line in two, written by two waves — and a partial-line write makes the L2 fetch the line first.
template <int R, bool STACK, int HL>
__global__ void __launch_bounds__(256)
k_sweep_sep2(const KP p, const float *__restrict__ in, float *__restrict__ out, const int seg_rows, const int nstrips, const int nsegs)
{
    static_assert(2 * HL >= R, "halo lanes hold two columns each");
    constexpr int S = 2 * R + 1, OUTW = 128 - 4 * HL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y, W = p.W, H = p.H;
    int strip, segi;
    bool desc = false; // march towards smaller x
    if constexpr (STACK) {
        strip = blockIdx.x % nstrips;
        segi = (blockIdx.x / nstrips) * 4 + wave;
        desc = (wave & 1) == 0;
    } else {
        // one wave per (column strip, row segment) pair, the pairs of an environment packed densely into its workgroups:
        // no wave is launched only to exit (strips per row rarely divide by 4), so the CUs keep their full wave count
        const int unit = blockIdx.x * 4 + wave;
        strip = unit % nstrips;
        segi = unit / nstrips;
    }
    __shared__ float taps[4 * S]; // u {hi, lo} [S], then v {hi, lo} [S]
    if (threadIdx.x < 2 * S) {
        taps[threadIdx.x] = p.fsep_u[threadIdx.x];
        taps[2 * S + threadIdx.x] = p.fsep_v[threadIdx.x];
    }
    __syncthreads();
    if (segi >= nsegs) return; // whole wave (no further barriers)
    const int y = strip * OUTW - 2 * HL + 2 * lane; // this lane's columns y, y + 1 (even: both inside the grid or both outside)
    const bool col_in = y >= 0 && y < H;
    const bool col_out = lane >= HL && lane < 64 - HL && y < H;
    const int yc = col_in ? y : 0;
    const float colmask = col_in ? 1.0f : 0.0f;
    const size_t G = (size_t)W * H;
    const float *src = in + (size_t)e * G * 2;
    float *dst = out + (size_t)e * G * 2;
    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    const bool clip = p.has_max_val && p.N > 0;
    const float thr = (float)p.threshold, mx = (float)p.max_val;
    // the filter taps as wave-uniform scalars (SGPRs), hi and lo parts; u in march order
    float uh[S], ul[S], vh[S], vl[S];
#pragma unroll
    for (int a = 0; a < S; ++a) {
        const int au = desc ? S - 1 - a : a;
        uh[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[2 * au])));
        ul[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[2 * au + 1])));
        vh[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[2 * S + 2 * a])));
        vl[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[2 * S + 2 * a + 1])));
    }
    // a ring of S running output rows: the input row at march position i feeds positions i - R .. i + R, position i - R is
    // complete after it.  The loop advances S positions per turn, so the ring slot of every (position, tap) pair is static.
    // All arithmetic on {channel 0, channel 1} PAIRS: v_pk_fma_f32 does both channels' FMA in one instruction (each
    // half is the IEEE fma of the scalar form), which halves the VALU work that otherwise co-limits this kernel.
    typedef float v2f __attribute__((ext_vector_type(2)));
#define FMA2(t, q, r) __builtin_elementwise_fma((v2f)(t), (q), (r))
#define SHR2(q) (v2f){wave_shr1((q).x), wave_shr1((q).y)}
#define SHL2(q) (v2f){wave_shl1((q).x), wave_shl1((q).y)}
    v2f acc[S][2]; // [ring slot][column y, column y + 1]
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j][0] = acc[j][1] = (v2f)(0.0f);
    const int x_lo = segi * seg_rows, x_hi = min(x_lo + seg_rows, W), nrows = x_hi - x_lo;
    // march position i <-> grid row x0 + dx * i: positions 0 .. nrows + 2R - 1 cover the segment and its halos
    const int x0 = desc ? x_hi - 1 + R : x_lo - R, dx = desc ? -1 : 1;
    float4 nv[S];
    uint32_t nword[S], ncell[S];
#define SEP2_LOAD1(I0, s)                                                                            \
    {                                                                                                \
        /* clamped to the grid (masked to zero when used) and to the march's last position: the prefetch of the \
           last turn must not pull in rows nobody needs (it did: 49 rows read per 32-row segment instead of 38) */ \
        const int xc = min(max(x0 + dx * min((I0) + (s), nrows + 2 * R - 1), 0), W - 1);           \
        ncell[s] = (uint32_t)(xc * H + yc);                                                          \
        nword[s] = walls[ncell[s] >> 5];                                                             \
        nv[s] = *reinterpret_cast<const float4 *>(src + (size_t)ncell[s] * 2);                       \
    }
#pragma unroll
    for (int s = 0; s < S; ++s) SEP2_LOAD1(0, s)
    for (int i0 = 0; i0 < nrows + 2 * R; i0 += S) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // zero fill outside the grid; walls.py:30 zeroes the INPUT of the convolution (arithmetic masks)
            const int xin = x0 + dx * (i0 + s);
            const float rowm = (xin >= 0 && xin < W) ? 1.0f : 0.0f;
            const uint32_t sh0 = ncell[s] & 31u; // (even cell index: the pair's two bits sit in one word)
            const float k0 = colmask * rowm * (float)(1u - ((nword[s] >> sh0) & 1u));
            const float k1 = colmask * rowm * (float)(1u - ((nword[s] >> (sh0 + 1u)) & 1u));
            const v2f a0 = (v2f){nv[s].x, nv[s].y} * (v2f)(k0), a1 = (v2f){nv[s].z, nv[s].w} * (v2f)(k1); // columns y, y + 1
            SEP2_LOAD1(i0 + S, s) // the same slot of the next turn (clamped: past the end a re-read of the last row)
            const v2f m1_0 = SHR2(a0), m1_1 = SHR2(a1); // lane - 1: columns y - 2, y - 1
            const v2f p1_0 = SHL2(a0), p1_1 = SHL2(a1); // lane + 1: columns y + 2, y + 3
            // nb0[d + R] = in[y + d], nb1[d + R] = in[y + 1 + d] for d = -R..R
            v2f nb0[S], nb1[S];
            nb0[R] = a0; nb1[R] = a1;
            nb0[R + 1] = a1; nb1[R - 1] = a0;
            nb0[R - 1] = m1_1; nb1[R + 1] = p1_0;
            nb0[R - 2] = m1_0; nb0[R + 2] = p1_0;
            nb1[R - 2] = m1_1; nb1[R + 2] = p1_1;
            if constexpr (R == 3) {
                const v2f m2_1 = SHR2(m1_1); // lane - 2: column y - 3
                const v2f p2_0 = SHL2(p1_0); // lane + 2: column y + 4
                nb0[R - 3] = m2_1; nb0[R + 3] = p1_1;
                nb1[R - 3] = m1_0; nb1[R + 3] = p2_0;
            }
            // this input row convolved across the columns with v: tap column b multiplies in[. - b + R]; summed
            // centre first, then outwards (the one-column march's order)
            v2f h0 = FMA2(vh[R], nb0[R], (v2f)(0.0f)), h1 = FMA2(vh[R], nb1[R], (v2f)(0.0f));
            h0 = FMA2(vl[R], nb0[R], h0); h1 = FMA2(vl[R], nb1[R], h1);
#pragma unroll
            for (int d = 1; d <= R; ++d) {
                h0 = FMA2(vh[R + d], nb0[R - d], h0); h1 = FMA2(vh[R + d], nb1[R - d], h1);
                h0 = FMA2(vl[R + d], nb0[R - d], h0); h1 = FMA2(vl[R + d], nb1[R - d], h1);
                h0 = FMA2(vh[R - d], nb0[R + d], h0); h1 = FMA2(vh[R - d], nb1[R + d], h1);
                h0 = FMA2(vl[R - d], nb0[R + d], h0); h1 = FMA2(vl[R - d], nb1[R + d], h1);
            }
            // position i + a - R += u[a] h (u in march order): ring slot (s + a - R) mod S
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const int slot = (s + a - R + 2 * S) % S;
                acc[slot][0] = FMA2(uh[a], h0, acc[slot][0]); acc[slot][1] = FMA2(uh[a], h1, acc[slot][1]);
                acc[slot][0] = FMA2(ul[a], h0, acc[slot][0]); acc[slot][1] = FMA2(ul[a], h1, acc[slot][1]);
            }
            { // position i - R just received its last contribution: the segment's row number i - 2R in march order
                const int slot = (s - R + 2 * S) % S;
                const int io = i0 + s - 2 * R;
                if (io >= 0 && io < nrows && col_out) {
                    const int x = desc ? x_hi - 1 - io : x_lo + io;
                    float r[4] = {acc[slot][0].x, acc[slot][0].y, acc[slot][1].x, acc[slot][1].y};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        r[c] = r[c] < thr ? 0.0f : r[c]; // pheromone.py:45
                        if (clip) r[c] = fminf(r[c], mx);
                    }
                    store_stream(reinterpret_cast<float4 *>(dst + ((size_t)x * H + y) * 2), make_float4(r[0], r[1], r[2], r[3]));
                }
                acc[slot][0] = acc[slot][1] = (v2f)(0.0f);
            }
        }
    }
#undef FMA2
#undef SHR2
#undef SHL2
#undef SEP2_LOAD1
}

// host-side launchers (called from antsrl_capi.hip)
template <int C>
static hipError_t launch_sweep_c(const KP &p, int cur, hipStream_t st)
{
    const float *in = p.s.phero[cur];
    float *out = p.s.phero[cur ^ 1];
    const size_t n = (size_t)p.E * p.W * p.H * C;
    if (p.filter_radius == 0) {
        if (((size_t)p.W * p.H * C) % 4 == 0) {
            const size_t per_env4 = (size_t)p.W * p.H * C / 4;
            const unsigned bx = (unsigned)((per_env4 + 256 * SW0_UNROLL - 1) / (256 * SW0_UNROLL));
            hipLaunchKernelGGL((k_sweep0<C>), dim3(bx, (unsigned)p.E), dim3(256), 0, st, p, in, out);
        } else {
            size_t blocks = (n + 255) / 256;
            if (blocks > 256 * 64) blocks = 256 * 64;
            hipLaunchKernelGGL((k_sweep0_scalar<C>), dim3((unsigned)blocks), dim3(256), 0, st, p, in, out, n);
        }
    } else if (!PROF_ENV("ANTSRL_SWEEP_TILED")) {
        const int fr = p.filter_radius;
        // The march is split along x into segments (each re-reads 2R rows, mostly from L2 / the Infinity Cache) and one
        // wave takes one (column strip, segment) pair; the pairs of an environment fill its workgroups densely.
        // Rows per segment, measured (profiles/r02/sweep_seg_explore.txt): radius 1 — 16 (c3 + 3x3 diffusion, two columns
        // per lane: 0.191 ms against 0.205 at 32, 0.219 at 64); radius 3, two columns — 32 (c4: 0.827 ms against 0.854 at
        // 64, 0.900 at 16, 0.907 at 128); radius 3, one column — 64 (0.936 against 1.001 at 32, 1.035 at 128).  Enough
        // waves to keep every SIMD full to the end, short enough streams to keep the chip's write window compact.
        const bool two_col = C == 2 && (p.H & 1) == 0 && !PROF_ENV("ANTSRL_SWEEP_ONE_COLUMN");
        int seg_rows = std::min(p.W, fr == 1 ? 16 : (two_col && p.filter_sep) || fr == 2 ? 32 : 64);
        if (const char *v = PROF_ENV("ANTSRL_SWEEP_SEG")) seg_rows = std::max(1, std::min(p.W, atoi(v)));
        const int nsegs = (p.W + seg_rows - 1) / seg_rows;
        if constexpr (C == 2) {
            // radius 1 (the reference's 3x3 diffusion), even H: two columns per lane, 16-byte accesses
            if (fr == 1 && two_col) {
                const int strips2 = (p.H + 127) / 128;
                hipLaunchKernelGGL(k_sweep_r1x2, dim3((strips2 * nsegs + 3) / 4, p.E), dim3(256), 0, st, p, in, out, seg_rows, strips2, nsegs);
                return hipGetLastError();
            }
            // radius 2..3 with a rank-1 filter (c4's Gaussian), even H: the separable march, two columns per lane
            if (fr >= 2 && p.filter_sep && two_col) {
                // 4 halo lanes per side: whole-line stores (see k_sweep_sep2); ANTSRL_SWEEP_MINHALO (profiling) = ceil(R / 2)
                const bool minhalo = PROF_ENV("ANTSRL_SWEEP_MINHALO") != nullptr;
                const int hl = minhalo ? (fr + 1) / 2 : 4;
                const int outw = 128 - 4 * hl;
                const int strips2 = (p.H + outw - 1) / outw;
                const bool stack = PROF_ENV("ANTSRL_SWEEP_STACK") != nullptr; // (four stacked segments per workgroup: measured slower)
                const dim3 grid2(stack ? strips2 * ((nsegs + 3) / 4) : (strips2 * nsegs + 3) / 4, p.E);
#define SEP2_GO(RR, ST, HLV) hipLaunchKernelGGL((k_sweep_sep2<RR, ST, HLV>), grid2, dim3(256), 0, st, p, in, out, seg_rows, strips2, nsegs)
                if (fr == 2) {
                    if (stack) { if (minhalo) SEP2_GO(2, true, 1); else SEP2_GO(2, true, 4); }
                    else { if (minhalo) SEP2_GO(2, false, 1); else SEP2_GO(2, false, 4); }
                } else {
                    if (stack) { if (minhalo) SEP2_GO(3, true, 2); else SEP2_GO(3, true, 4); }
                    else { if (minhalo) SEP2_GO(3, false, 2); else SEP2_GO(3, false, 4); }
                }
#undef SEP2_GO
                return hipGetLastError();
            }
        }
        const int strips = (p.H + (64 - 2 * fr) - 1) / (64 - 2 * fr);
        const dim3 grid((strips * nsegs + 3) / 4, p.E);
#define MARCH_GO(RR, SEPV) hipLaunchKernelGGL((k_sweep_march<C, RR, SEPV>), grid, dim3(256), 0, st, p, in, out, seg_rows, strips, nsegs)
        if (p.filter_sep) {
            if (fr == 1) MARCH_GO(1, true);
            else if (fr == 2) MARCH_GO(2, true);
            else MARCH_GO(3, true);
        } else if (fr == 1) MARCH_GO(1, false);
        else if (fr == 2) MARCH_GO(2, false);
        else MARCH_GO(3, false);
#undef MARCH_GO
    } else { // LDS-tiled float64 reference variant (A/B and cross-check: ANTSRL_SWEEP_TILED=1)
        const int fr = p.filter_radius;
        const size_t lds = (size_t)(SW_TX + 2 * fr) * (SW_TY + 2 * fr) * C * sizeof(float);
        dim3 grid((p.H + SW_TY - 1) / SW_TY, (p.W + SW_TX - 1) / SW_TX, p.E);
        hipLaunchKernelGGL((k_sweep_tiled<C>), grid, dim3(256), lds, st, p, in, out);
    }
    return hipGetLastError();
}

hipError_t antsrl_launch_sweep(const KP &p, int cur, hipStream_t st)
{
    switch (p.C) {
    case 1: return launch_sweep_c<1>(p, cur, st);
    case 2: return launch_sweep_c<2>(p, cur, st);
    case 3: return launch_sweep_c<3>(p, cur, st);
    case 4: return launch_sweep_c<4>(p, cur, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t antsrl_launch_phero_wall_clear(const KP &p, hipStream_t st, int buf)
{
    hipLaunchKernelGGL(k_phero_wall_clear, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p, buf);
    return hipGetLastError();
}

hipError_t antsrl_launch_phero_renorm(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_phero_renorm, dim3(grid_for((size_t)p.E * p.W * p.H * p.C)), dim3(256), 0, st, p);
    return hipGetLastError();
}
