// antsrl_util.h — device helpers shared by the kernel translation units (gfx950, wave64).
// Compile with -ffp-contract=off (numpy rounds every product before adding).
// Reference citations are relative to the reference checkout (environment/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "antsrl_device.h"
#include "antsrl_layout.h"

#define WAVE 64
#define PI_D 3.141592653589793
#define HASH_EMPTY 0xFFFFFFFFu

// food of cell g of one env, whatever the cell stride (DState::food): `food[g]` reads / writes it
struct FoodView {
    float *b;
    int st;
    __device__ __forceinline__ float &operator[](size_t g) const { return b[g * (size_t)st]; }
};

// ------------------------------------------------------------------ cell records
// Index of cell (x, y) / of row-major cell id `cell` = x * H + y in an environment's cell-record arrays under the strides
// KP::ps / KP::fs: prec_* for the PHEROMONE array (DState::phero), frec_* for the FOOD / META array (DState::food).
//   interleaved 16-byte records (KP::tiled): ONE array, blocks of 2 (x) by 4 (y) cells per 128-byte line — both the same;
//   explicit-sweep layout: the pheromone buffers row-major (the stencils march along them), the 8-byte {food, META}
//   records in blocks of 4 x 4 cells per line (KP::ftile);  everything else row-major.
__device__ __forceinline__ uint32_t prec_xy(const KP &p, const int x, const int y)
{
    return p.tiled ? tiled_slot(x, y, p.H) : (uint32_t)(x * p.H + y);
}
__device__ __forceinline__ uint32_t frec_xy(const KP &p, const int x, const int y)
{
    return p.tiled ? tiled_slot(x, y, p.H) : p.ftile ? tiled44_slot(x, y, p.H) : (uint32_t)(x * p.H + y);
}
__device__ __forceinline__ uint32_t prec_cell(const KP &p, const uint32_t cell)
{
    if (!p.tiled) return cell;
    const uint32_t x = cell / (uint32_t)p.H;
    return prec_xy(p, (int)x, (int)(cell - x * (uint32_t)p.H));
}
__device__ __forceinline__ uint32_t frec_cell(const KP &p, const uint32_t cell)
{
    if (!p.tiled && !p.ftile) return cell;
    const uint32_t x = cell / (uint32_t)p.H;
    return frec_xy(p, (int)x, (int)(cell - x * (uint32_t)p.H));
}

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ double np_mod_d(double a, double b)
{
    // np.mod on float64 (python sign convention), ants.py:63,70-71
    double r = fmod(a, b);
    if (r != 0.0) {
        if ((b < 0) != (r < 0)) r += b;
    } else {
        r = copysign(0.0, b);
    }
    return r;
}

__device__ __forceinline__ double warp_coord(double v, double size)
{
    // Ants.warp_xy, ants.py:69-71; the single value `size` (np.mod(-1e-17, W) == W, where
    // the reference raises IndexError) maps to 0 — same convention as the oracle.
    double r = np_mod_d(v, size);
    if (r >= size) r = 0.0;
    return r;
}

__device__ __forceinline__ int wrap_index(int v, int n)
{
    if (v < 0 || v >= n) {
        v %= n;
        if (v < 0) v += n;
    }
    return v;
}

__device__ __forceinline__ bool test_bit(const uint32_t *bits, uint32_t cell)
{
    return (bits[cell >> 5] >> (cell & 31)) & 1u;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

// Counter-based uniform [0,1): same specification as oracle_jitter_u01 (integer-exact).
__device__ __forceinline__ double jitter_u01(uint64_t seed, uint32_t env, uint32_t timestep, uint32_t ant)
{
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)env + 1));
    k = mix64(k ^ (0xD1B54A32D192ED03ULL * ((uint64_t)timestep + 1)));
    k = mix64(k + 0x9E3779B97F4A7C15ULL * ((uint64_t)ant + 1));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

// ---- last-writer-wins resolution --------------------------------------------------
// numpy's `a[idx] += v` with repeated indices keeps only the LAST ant's update
// (pheromone.py:39, ants.py:116).  Deterministic regardless of wave scheduling: an LDS
// open-addressing table maps cell -> highest ant index standing on it.
__device__ __forceinline__ uint32_t lww_hash(uint32_t cell, uint32_t mask)
{
    return (cell * 2654435761u >> 7) & mask;
}

__device__ __forceinline__ void lww_insert(uint32_t *keys, uint32_t *vals, uint32_t mask, uint32_t cell,
                                           uint32_t ant)
{
    uint32_t h = lww_hash(cell, mask);
    for (;;) {
        uint32_t k = atomicCAS(&keys[h], HASH_EMPTY, cell);
        if (k == HASH_EMPTY || k == cell) {
            atomicMax(&vals[h], ant);
            return;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ uint32_t lww_winner(const uint32_t *keys, const uint32_t *vals, uint32_t mask,
                                               uint32_t cell)
{
    uint32_t h = lww_hash(cell, mask);
    while (keys[h] != cell) h = (h + 1) & mask;
    return vals[h];
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS hand-off between lanes of ONE wave.  A wave's LDS instructions execute in issue order,
    // so no s_waitcnt is needed — only a fence that keeps the COMPILER from reordering the staging
    // writes and the copy-out reads.  Wavefront scope on purpose: a workgroup-scope release would
    // also drain the wave's outstanding global stores (vmcnt(0)) and stall it behind the
    // observation writes of the previous round.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Streaming stores for the step's OUTPUT tensors (observation, agent_state, reward): nobody on the
// device re-reads them within the step, and at 0.7 GB per launch a cached write stream evicts the
// pheromone/food lines the perception gathers reuse and the ant state k_update reads next
// (measured on c3, same box: k_act 0.340 -> 0.287 ms, k_update 0.051 -> 0.043 ms).
#define ANTSRL_MAX_DEVICES 64 // per-device launch bookkeeping (dynamic-LDS opt-in)
typedef float stream_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t stream_u4 __attribute__((ext_vector_type(4)));
#define ANTSRL_NT_STORE(v, p) __builtin_nontemporal_store(v, p) // (plain stores: k_perceive 0.278 against 0.233 ms)
__device__ __forceinline__ void store_stream(float *dst, float v) { ANTSRL_NT_STORE(v, dst); }
__device__ __forceinline__ void store_stream(uint16_t *dst, uint16_t v) { ANTSRL_NT_STORE(v, dst); }
__device__ __forceinline__ void store_stream(uint4 *dst, const uint4 &v)
{
    ANTSRL_NT_STORE((stream_u4{v.x, v.y, v.z, v.w}), reinterpret_cast<stream_u4 *>(dst));
}
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); } // RNE
__device__ __forceinline__ void store_stream(float4 *dst, const float4 &v)
{
    ANTSRL_NT_STORE((stream_f4{v.x, v.y, v.z, v.w}), reinterpret_cast<stream_f4 *>(dst));
}

// Per-ant struct-of-arrays state.  The arrays ONLY the per-environment kernels touch (previous position, mandibles, tint,
// activation, the sparse-update lists, the actions) are streamed (STP_LD / STP_ST): they are read and written once per step
// by k_update_move and by nobody else, and as `nt` accesses they take no Infinity Cache space from the ~205 MB of
// cell-record lines the perception gathers re-use step after step (c3, same box: 0.2536 -> 0.2500 ms/step;
// profiles/r03/state_nt2_ab.txt).  k_perceive's inputs (x, y, theta, holding, seed) stay cached: streamed, too, they cost
// k_perceive more than the cache space is worth (state_nt_ab.txt: +2 %).
#define STP_LD(lv) __builtin_nontemporal_load(&(lv))
#define STP_ST(lv, v) __builtin_nontemporal_store((v), &(lv))

// The smallest double T with sqrt(T) >= r, so that  sqrt(d2) < r  <=>  d2 < T  exactly (sqrt is correctly
// rounded and monotone): the per-cell rock test (circle_obstacles.py via RL_api.py:132-135,
// `dist < radius` on a float64 norm) then needs no square root.  r <= 0 never matches (T = 0).
__device__ __forceinline__ double sqrt_lt_threshold(double r)
{
    if (!(r > 0.0)) return 0.0;
    double t = r * r;
    for (int it = 0; it < 8 && sqrt(t) >= r; ++it) t = __longlong_as_double(__double_as_longlong(t) - 1); // step down
    for (int it = 0; it < 16 && sqrt(t) < r; ++it) t = __longlong_as_double(__double_as_longlong(t) + 1); // first t with sqrt(t) >= r
    return t;
}

__host__ __device__ __forceinline__ size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__host__ __device__ inline size_t update_scratch_bytes(int HT, int R, int nwaves)
{
    return align_up(8 * (size_t)HT, 16) + 16 * (size_t)(R > 0 ? R : 1) + 8 * (size_t)nwaves +
           4 * (size_t)nwaves + 16;
}

static inline unsigned grid_for(size_t n)
{
    size_t b = (n + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// The environment a workgroup (or a k_perceive workgroup's environment slot) `i` of `E` takes in observation `seq`: odd
// observations walk the environments from the other end.  The batch's per-step working set (c3: ~240 MB of cell-record
// lines + ~60 MB of ant state, nearly the same lines every step) is a little larger than the 256 MiB Infinity Cache; an
// LRU cycled in ONE direction over slightly more than its capacity hits almost never, walked back and forth it keeps the
// most recently used end (measured: k_perceive 0.2325 -> 0.2238 ms on c3, -3.8 % per step at 768 envs;
// profiles/r02/altorder_ab.txt).  Speed only: any order gives the same results.
__device__ __forceinline__ int env_of_block(const int i, const int E, const uint32_t seq)
{
    return (seq & 1u) ? E - 1 - i : i;
}
