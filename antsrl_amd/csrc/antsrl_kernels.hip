// antsrl_kernels.hip — hand-written HIP kernels for gfx950 (MI355X), wave64.
//
// One simulation step (RLApi.step + Environment.update of the reference) is three launches:
//   k_sweep0 / k_sweep_tiled   pheromone decay / diffuse / threshold / wall-zero / clip,
//                              phero[cur] -> phero[cur^1]          (HBM streaming, the bulk)
//   k_act                      one workgroup per environment: mandibles + food exchange,
//                              activation, rotate, move, 7x7 perception gather, reward
//   k_update                   one workgroup per environment: wall revert + jitter, rock
//                              push, prev:=cur, pheromone deposit, anthill collect
// No MFMA anywhere: nothing on this path is a dense contraction.  Compile with
// -ffp-contract=off (numpy rounds every product before adding).
//
// Reference citations are relative to the reference checkout (environment/...).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "antsrl_device.h"

#define WAVE 64
#define PI_D 3.141592653589793
#define HASH_EMPTY 0xFFFFFFFFu

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
typedef float stream_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t stream_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float *dst, float v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void store_stream(uint16_t *dst, uint16_t v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void store_stream(uint4 *dst, const uint4 &v)
{
    __builtin_nontemporal_store(stream_u4{v.x, v.y, v.z, v.w}, reinterpret_cast<stream_u4 *>(dst));
}
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); } // RNE
__device__ __forceinline__ void store_stream(float4 *dst, const float4 &v)
{
    __builtin_nontemporal_store(stream_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<stream_f4 *>(dst));
}

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

// ===================================================================================
// k_act — RLApi.step (RL_api.py:168-204) / RLApi.observation (RL_api.py:96-165)
// one workgroup per environment
// ===================================================================================
__host__ __device__ inline size_t update_scratch_bytes(int HT, int R, int nwaves)
{
    return align_up(8 * (size_t)HT, 16) + 16 * (size_t)(R > 0 ? R : 1) + 8 * (size_t)nwaves +
           4 * (size_t)nwaves + 16;
}

template <int C>
__device__ __forceinline__ void update_env(const KP &p, const int e, const double *__restrict__ wall_jitter,
                                           const int out_buf, unsigned char *smem);

struct __align__(16) AntFrame { double cx, cy, ct, st; }; // perception centre, cos/sin(theta + pi/2)
struct __align__(16) CellOff { double px, py; };            // rotated-grid offsets, RL_api.py:92-93

struct ActLds {
    AntFrame *frame;                   // [N]
    CellOff *off;                      // [PP]
    uint32_t *cnt;                     // [N] unexplored-cell count / temp cell index
    uint32_t *rockmask;                // [N] rocks that can touch the ant's patch
    uint32_t *b_pres, *b_old;          // [words] presence / explored map as it was before this step
    uint32_t *b_walls, *b_area;        // [words] (only when STATIC_LDS)
    uint8_t *t_mask;                   // [PP]
    double *rock;                      // [3R] cx, cy, radius of this env's rocks
    uint32_t *hkeys, *hvals;           // [HT] — aliases `stage`
    float *stage;                      // [nwaves][stage_stride]: one ant's K*PP outputs (+ alignment pad)
    uint32_t stage_stride;             // floats per wave, multiple of 4
};

#define ACT_UNROLL 2                 // ants in flight per wave (all their gathers are issued before the first is consumed)
#define ACT_ITEMS (64 * ACT_UNROLL)  // work items per wave per iteration

// Byte offsets of the k_act LDS carve.  Plain integers on purpose: the kernel forms its LDS pointers
// locally from `smem + offset`, so they keep the LDS address space no matter what the optimiser does
// (a struct of pointers filled through an out-parameter ends up in scratch once the kernel grows, and
// every LDS access then degrades to flat_* with vmcnt(0) waits).
struct ActOff {
    uint32_t frame, off, cnt, rm, pres, old, walls, area, mask, rock, uni, stride;
    size_t total;
};

__host__ __device__ __forceinline__ ActOff act_offsets(int N, int PP, int words, int HT, int K, int nwaves,
                                                       bool static_lds, int R)
{
    ActOff o;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off = (off + bytes + 15) / 16 * 16;
        return (uint32_t)at;
    };
    o.frame = take(sizeof(AntFrame) * (size_t)N);
    o.off = take(sizeof(CellOff) * (size_t)PP);
    o.cnt = take(4 * (size_t)N);
    o.rm = take(4 * (size_t)N);
    o.pres = take(4 * (size_t)words);
    o.old = take(4 * (size_t)words);
    o.walls = o.area = 0;
    if (static_lds) {
        o.walls = take(4 * (size_t)words);
        o.area = take(4 * (size_t)words);
    }
    o.mask = take((size_t)PP);
    o.rock = take(32 * (size_t)(R > 0 ? R : 1));
    const size_t stride = ((size_t)PP * K + 3 + 3) / 4 * 4; // row + up to 3 floats of misalignment
    const size_t hash_b = update_scratch_bytes(HT, R, nwaves), stage_b = 4 * (size_t)nwaves * stride;
    o.uni = take(hash_b > stage_b ? hash_b : stage_b);
    o.stride = (uint32_t)stride;
    o.total = off;
    return o;
}

__host__ __device__ __forceinline__ size_t act_lds_bytes(int N, int PP, int words, int HT, int K, int nwaves,
                                                         bool static_lds, void *, unsigned char *, int R = 0)
{
    return act_offsets(N, PP, words, HT, K, nwaves, static_lds, R).total;
}

// Perception-channel layouts known at compile time (straight-line output code); anything else
// takes the generic per-channel selection.
#define LAYOUT_GENERIC 0
#define LAYOUT_DEFAULT 1       // [Ants, Phero0, Phero1, Anthill, Walls, Food]   (generator order)
#define LAYOUT_DEFAULT_ROCKS 2 // ... + [CircleObstacles]

// Profiling only (ANTSRL_ABLATE bit ACT_ABL_TRACE): per-workgroup phase timeline of k_act.  Slot k of
// workgroup e = s_memrealtime (100 MHz) at: 0 entry, 1 after phase 2, 2 after phase 3, 3 exit;
// slot 4 = HW_ID, slot 5 = XCC_ID, slot 6 after phase 0, slot 7 after phase 1.  Read back with antsrl_debug_read_act_trace.
#define ACT_TRACE_SLOTS 8
#define ACT_TRACE_MAX_WG 8192
__device__ unsigned long long g_act_trace[ACT_TRACE_SLOTS * ACT_TRACE_MAX_WG];

__device__ __forceinline__ void act_trace(int flags, int e, int tid, int slot)
{
    if ((flags & ACT_ABL_TRACE) && tid == 0 && e < ACT_TRACE_MAX_WG) {
        g_act_trace[e * ACT_TRACE_SLOTS + slot] = wall_clock64();
        if (slot == 0) {
            g_act_trace[e * ACT_TRACE_SLOTS + 4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_HW_ID
            g_act_trace[e * ACT_TRACE_SLOTS + 5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); // HW_REG_XCC_ID
        }
    }
}

extern "C" int antsrl_debug_read_act_trace(unsigned long long *dst, int n_wg)
{
    if (!dst || n_wg < 0 || n_wg > ACT_TRACE_MAX_WG) return ANTSRL_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ANTSRL_E_DEVICE;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_act_trace), sizeof(unsigned long long) * ACT_TRACE_SLOTS * (size_t)n_wg)
                   == hipSuccess ? ANTSRL_OK : ANTSRL_E_DEVICE;
}

// FAST selects the software-pipelined perception loop (see phase 3); the two loops live in
// separate instantiations on purpose: with both in one kernel the optimiser stops scalarising the
// LDS carve (`ActLds`), its pointers go through scratch and every LDS access degrades to flat_*.
// TPB = threads per workgroup (512 or 1024); 4 waves per SIMD (<= 128 VGPRs) is all the LDS plans
// can use (capping at 80 VGPRs for a third workgroup per CU measured slower, see plan_act).
template <int C, bool STATIC_LDS, int LAYOUT, bool FAST, int TPB, bool OBS16 = false>
__global__ void __launch_bounds__(TPB, 4)
k_act(const KP p, const int8_t *__restrict__ rotation, const int8_t *__restrict__ phero_act, const int cur,
      float *__restrict__ obs, float *__restrict__ agent_state, float *__restrict__ reward,
      uint8_t *__restrict__ done, const int flags, const double *__restrict__ wall_jitter, const int out_buf)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int e = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    // the wave index is wave-uniform: say so, and every per-ant address below is computed on the SALU
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = T >> 6;
    const int N = p.N, W = p.W, H = p.H, K = p.K, P = p.P, PP = p.PP, R = p.R;
    const size_t G = (size_t)W * H;
    const ActOff lo = act_offsets(N, PP, p.words, p.HT, K, nwaves, STATIC_LDS, R);
    ActLds L; // filled field by field right here: never has its address taken, stays in registers
    L.frame = (AntFrame *)(smem + lo.frame);
    L.off = (CellOff *)(smem + lo.off);
    L.cnt = (uint32_t *)(smem + lo.cnt);
    L.rockmask = (uint32_t *)(smem + lo.rm);
    L.b_pres = (uint32_t *)(smem + lo.pres);
    L.b_old = (uint32_t *)(smem + lo.old);
    L.b_walls = (uint32_t *)(smem + lo.walls);
    L.b_area = (uint32_t *)(smem + lo.area);
    L.t_mask = smem + lo.mask;
    L.rock = (double *)(smem + lo.rock);
    L.hkeys = (uint32_t *)(smem + lo.uni);
    L.hvals = L.hkeys + p.HT;
    L.stage = (float *)(smem + lo.uni);
    L.stage_stride = lo.stride;

    const size_t eN = (size_t)e * N;
    const uint32_t *g_walls = p.s.walls_bits + (size_t)e * p.words;
    const uint32_t *g_area = p.s.area_bits + (size_t)e * p.words;
    uint32_t *g_expl = p.s.explored_bits + (size_t)e * p.words;
    const uint32_t *walls = STATIC_LDS ? L.b_walls : g_walls;
    const uint32_t *area = STATIC_LDS ? L.b_area : g_area;
    float *food = p.s.food + (size_t)e * G;
    const float *ph = p.s.phero[cur] + (size_t)e * G * C;
    const bool do_step = flags & ACT_STEP;
    const bool explore = p.explore_on != 0;
    const uint8_t primed0 = p.s.reward_primed[e];

    act_trace(flags, e, tid, 0);
    // With one ant per thread (N <= T: the reference's sizes) every independent global load of the ant is
    // issued HERE, ahead of the bitmap staging and the first barrier, and the phases below use the
    // registers: otherwise each phase starts with a dependent round trip to HBM behind a barrier
    // (14 us of a 100 us workgroup at c3, with the memory system idle).
    const bool one = N <= T; // wave-uniform
    const size_t a1 = eN + (tid < N ? tid : 0);
    double h_x = 0.0, h_y = 0.0, h_th = 0.0;
    float h_hold = 0.0f, h_q = 0.0f;
    int h_m = 0, h_rot = 0, h_pa = 0;
    uint32_t h_cprev = 0u;
    if (one) {
        h_x = p.s.x[a1]; h_y = p.s.y[a1]; h_th = p.s.theta[a1];
        h_hold = p.s.holding[a1];
        if (do_step) {
            const double ppx = p.s.prev_x[a1], ppy = p.s.prev_y[a1];
            h_m = p.s.mandibles[a1];
            if (rotation) h_rot = rotation[a1];
            if (phero_act) h_pa = phero_act[a1];
            h_cprev = (uint32_t)((int)ppx * H + (int)ppy);
            h_q = food[h_cprev]; // food is first written in phase 1b
        }
    }
    // ---- phase 0: stage bitmaps and tables in LDS (16 bytes per lane, all loads of a pass in flight)
    {
        const int w4n = (p.words & 3) ? 0 : p.words >> 2; // env bases stay 16-byte aligned only then
        const uint4 *o4 = reinterpret_cast<const uint4 *>(g_expl), *wl4 = reinterpret_cast<const uint4 *>(g_walls),
                    *ar4 = reinterpret_cast<const uint4 *>(g_area);
        for (int w = tid; w < w4n; w += T) { // (the bitmap arrays are 256-byte aligned in the workspace)
            const uint4 vo = o4[w];
            uint4 vw = make_uint4(0, 0, 0, 0), va = vw;
            if (STATIC_LDS) { vw = wl4[w]; va = ar4[w]; }
            reinterpret_cast<uint4 *>(L.b_pres)[w] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4 *>(L.b_old)[w] = vo;
            if (STATIC_LDS) {
                reinterpret_cast<uint4 *>(L.b_walls)[w] = vw;
                reinterpret_cast<uint4 *>(L.b_area)[w] = va;
            }
        }
        for (int w = 4 * w4n + tid; w < p.words; w += T) {
            L.b_pres[w] = 0u;
            L.b_old[w] = g_expl[w];
            if (STATIC_LDS) {
                L.b_walls[w] = g_walls[w];
                L.b_area[w] = g_area[w];
            }
        }
    }
    for (int q = tid; q < PP; q += T) {
        int a = q / P, b = q % P;
        L.off[q].px = (double)(b - p.r) * p.delta; // coords[a][b] = (arange[b], arange[a]) * DELTA
        L.off[q].py = (double)(a - p.r) * p.delta;
        L.t_mask[q] = p.has_mask ? p.mask[q] : (uint8_t)1;
    }
    for (int q = tid; q < R; q += T) {
        const double rad = p.s.rock_r[(size_t)e * R + q];
        L.rock[4 * q + 0] = p.s.rock_cx[(size_t)e * R + q];
        L.rock[4 * q + 1] = p.s.rock_cy[(size_t)e * R + q];
        L.rock[4 * q + 2] = rad;
        L.rock[4 * q + 3] = sqrt_lt_threshold(rad);
    }
    if (do_step)
        for (int h = tid; h < p.HT; h += T) {
            L.hkeys[h] = HASH_EMPTY;
            L.hvals[h] = 0u;
        }
    __syncthreads();
    act_trace(flags, e, tid, 6);

    if (do_step) {
        // ---- phase 1a: mandible target (RL_api.py:178-185) + Ants.update_mandibles reads
        //      (ants.py:102-114).  All food reads happen before any food write.
        float *tmp_q = (float *)L.frame, *tmp_d = tmp_q + N; // frame memory is free until phase 2
        for (int i = tid; i < N; i += T) {
            double x, y;
            uint32_t cprev;
            float q, hold;
            int old_m;
            if (one) {
                x = h_x; y = h_y; cprev = h_cprev; q = h_q; old_m = h_m; hold = h_hold;
            } else {
                x = p.s.x[eN + i]; y = p.s.y[eN + i];
                cprev = (uint32_t)((int)p.s.prev_x[eN + i] * H + (int)p.s.prev_y[eN + i]);
                q = food[cprev];
                old_m = p.s.mandibles[eN + i];
                hold = p.s.holding[eN + i];
            }
            const uint32_t ccur = (uint32_t)((int)x * H + (int)y);
            int m = old_m;
            for (int k = 0; k < K; ++k) { // perceived_objects order matters
                if (p.ch_kind[k] == ANTSRL_CH_FOOD) m = (q > 0.0f) | m;                      // :182
                else if (p.ch_kind[k] == ANTSRL_CH_ANTHILL) m = (1 - (int)test_bit(area, ccur)) & m; // :184
            }
            const int closing = m & (1 - old_m), opening = (1 - m) & old_m; // ants.py:103-104
            const float taken = fminf((float)p.max_hold, fmaxf(0.0f, q)) * (float)closing; // :111
            const float dropped = hold * (float)opening;                                    // :114
            h_hold = hold + (taken - dropped);                                              // :117
            p.s.holding[eN + i] = h_hold;
            p.s.mandibles[eN + i] = (uint8_t)m;                                             // :107
            L.cnt[i] = cprev;
            tmp_q[i] = q;
            tmp_d[i] = dropped - taken;
            lww_insert(L.hkeys, L.hvals, (uint32_t)p.HT - 1, cprev, (uint32_t)i);
        }
        __syncthreads();
        // ---- phase 1b: ants.py:116 `qte[cell] += dropped - taken`, last ant on a cell wins
        for (int i = tid; i < N; i += T) {
            const uint32_t cprev = L.cnt[i];
            const float delta = tmp_d[i];
            int32_t dirty = -1;
            if (delta != 0.0f && lww_winner(L.hkeys, L.hvals, (uint32_t)p.HT - 1, cprev) == (uint32_t)i) {
                food[cprev] = tmp_q[i] + delta;
                if (test_bit(area, cprev)) dirty = (int32_t)cprev;
            }
            p.s.dirty_cell[eN + i] = dirty;
        }
        __syncthreads();
    }
    act_trace(flags, e, tid, 7);

    // ---- phase 2: activation, rotate, move (RL_api.py:187-196) and the perception frame
    const double margin = (double)p.r * p.delta * 1.4142135623730951 + 1.5;
    for (int i = tid; i < N; i += T) {
        double x, y, th;
        if (one) {
            x = h_x; y = h_y; th = h_th;
        } else {
            x = p.s.x[eN + i]; y = p.s.y[eN + i]; th = p.s.theta[eN + i];
        }
        if (do_step) {
            if (phero_act) { // Ants.activate_pheromone, ants.py:89-96
                const int a = one ? h_pa : (int)phero_act[eN + i];
                float a0 = 0.0f, a1 = 0.0f;
                if (a == 1) a0 = (float)p.deposit_strength;
                else if (a != 0) a1 = (float)p.deposit_strength;
                p.s.activation[(eN + i) * C + 0] = a0;
                if (C > 1) p.s.activation[(eN + i) * C + 1] = a1;
            }
            if (rotation) // Ants.rotate_ants + warp_theta, ants.py:62-67
                th = np_mod_d(th + (double)(one ? h_rot : (int)rotation[eN + i]) * p.max_rot_speed, 2 * PI_D);
        }
        double sn, cs;
        sincos(th, &sn, &cs);
        if (do_step) {
            // RL_api.py:194-196, Ants.forward_ants ants.py:77-80
            double fwd = 1.0 * p.max_speed * (1 - (double)(one ? h_hold : p.s.holding[eN + i]) * p.carry);
            if (fwd < 0) fwd *= p.backward;
            x = warp_coord(x + cs * fwd, (double)W);
            y = warp_coord(y + sn * fwd, (double)H);
            p.s.x[eN + i] = x;
            p.s.y[eN + i] = y;
            p.s.theta[eN + i] = th;
        }
        // RL_api.py:100-108
        double xf = x, yf = y;
        if (p.fwd_delta != 0.0) {
            xf += cs * p.fwd_delta;
            yf += sn * p.fwd_delta;
        }
        double st, ct;
        sincos(th + PI_D * 0.5, &st, &ct);
        AntFrame fr;
        fr.cx = xf; fr.cy = yf; fr.ct = ct; fr.st = st;
        L.frame[i] = fr;
        L.cnt[i] = 0u;
        // presence map, RL_api.py:137-141 (0/1, not a count)
        const uint32_t cell = (uint32_t)(wrap_index((int)x, W) * H + wrap_index((int)y, H));
        atomicOr(&L.b_pres[cell >> 5], 1u << (cell & 31));
        // rocks whose disc can reach this ant's patch (conservative; exact test per cell)
        uint32_t rm = 0u;
        if (R > 0) {
            const bool border = xf - margin < 0 || yf - margin < 0 || xf + margin >= W || yf + margin >= H;
            for (int q = 0; q < R; ++q) {
                const double dx = L.rock[4 * q + 0] - xf, dy = L.rock[4 * q + 1] - yf;
                const double rr = L.rock[4 * q + 2] + margin;
                if (border || dx * dx + dy * dy < rr * rr) rm |= 1u << q;
            }
        }
        L.rockmask[i] = rm;
    }
    __syncthreads();
    act_trace(flags, e, tid, 1);

    // ---- phase 3: perception gather, RL_api.py:109-148.  One WAVE per ant, one LANE per perceived
    //      cell (49 of 64 lanes at the reference's 7x7): the cell's offsets, mask bit and output slot
    //      are per-lane constants held in registers, the ant's frame is wave-uniform (LDS
    //      broadcast), ACT_UNROLL ants are in flight per wave so every gather is issued before the
    //      first is consumed.  Each ant's K*PP outputs are staged in LDS and leave as 16-byte stores.
    float *obs_env = (flags & ACT_HAS_OBS) ? obs + (size_t)e * (size_t)N * PP * K : nullptr;
    const float inv_max = 1.0f / (float)p.max_val;
    const float g_now = (float)p.g_now;                       // scaled mode: v = u * f0^S ...
    const float cut = p.scaled ? (float)p.threshold : 0.0f;   // ... and 0 below the 0.01 cut
    const bool abl_gather = flags & ACT_ABL_NO_GATHER, abl_store = flags & ACT_ABL_NO_STORE;
    const bool abl_explore = flags & ACT_ABL_NO_EXPLORE;
    const int npass = (PP + 63) >> 6;
    const uint32_t row = (uint32_t)PP * (uint32_t)K;            // floats per ant
    float *stage = L.stage + (size_t)wave * L.stage_stride;
    const bool wrap_fast = W > 4 * (p.r + 4) && H > 4 * (p.r + 4) && p.fwd_delta < W / 4 && p.fwd_delta < H / 4 &&
                           p.fwd_delta > -W / 4 && p.fwd_delta > -H / 4 && p.delta < 2.0; // one conditional add wraps
    const bool wrap_pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0;
    // Fast path (single pass: PP <= 64, row <= 508 floats — the reference's 7x7 with up to 10
    // channels): software-pipelined by one group of ACT_UNROLL ants.  Everything that touches
    // global memory is STRAIGHT-LINE and unconditional (out-of-range ants/lanes are clamped onto
    // valid ones and redo identical work: benign duplicate stores), so the compiler can count
    // outstanding operations: the wait for group g's gathers is a `vmcnt(n)` that leaves group
    // g+1's gathers AND group g-1's observation stores in flight.  (vmcnt retires in order and
    // counts stores: an uncounted wait would make every gather wait for the previous stores.)
    if (FAST) { // host guarantees: npass == 1, 8 <= row <= 508, obs != nullptr
        const int q = lane < PP ? lane : PP - 1;            // lanes beyond the perception clamp onto its last cell
        const CellOff of = L.off[q];
        const bool mask_q = L.t_mask[q] != 0;
        const uint32_t qK = (uint32_t)q * K;
        // Two register sets (current / prefetched group); plain arrays with compile-time indices only,
        // so they stay in VGPRs (a struct passed by reference ends up in scratch).
#define ACT_FETCH(G0, CELL, IXV, IYV, PVV, FDV)                                                          \
    {                                                                                                    \
        _Pragma("unroll") for (int u = 0; u < ACT_UNROLL; ++u)                                           \
        {                                                                                                \
            const int i_ = min((G0) + u, N - 1);                                                         \
            const AntFrame fr = L.frame[i_]; /* wave-uniform address: LDS broadcast */                   \
            const double rx = fr.ct * of.px - fr.st * of.py; /* RL_api.py:110-111 */                     \
            const double ry = fr.st * of.px + fr.ct * of.py;                                             \
            int ix = (int)rint(rx + fr.cx), iy = (int)rint(ry + fr.cy); /* :114-117 half to even */      \
            if (wrap_pow2) { /* :118-119; two's complement AND is the floor-mod for a power of two */    \
                ix &= W - 1; iy &= H - 1;                                                                \
            } else if (wrap_fast) { /* |ix| < 2W: unsigned min picks the in-range candidate */           \
                ix = (int)min(min((uint32_t)ix, (uint32_t)(ix + W)), (uint32_t)(ix - W));                \
                iy = (int)min(min((uint32_t)iy, (uint32_t)(iy + H)), (uint32_t)(iy - H));                \
            } else {                                                                                     \
                ix = wrap_index(ix, W); iy = wrap_index(iy, H);                                          \
            }                                                                                            \
            IXV[u] = ix; IYV[u] = iy;                                                                    \
            CELL[u] = (uint32_t)(ix * H + iy);                                                           \
        }                                                                                                \
        /* unconditional gathers (masked cells too: in bounds, discarded) */                             \
        _Pragma("unroll") for (int u = 0; u < ACT_UNROLL; ++u)                                           \
        {                                                                                                \
            const uint32_t gc_ = CELL[u];                                                                \
            if (C == 2) {                                                                                \
                const float2 t = *reinterpret_cast<const float2 *>(ph + (size_t)gc_ * 2);                \
                PVV[u][0] = t.x; PVV[u][C - 1] = t.y;                                                    \
            } else {                                                                                     \
                _Pragma("unroll") for (int c = 0; c < C; ++c) PVV[u][c] = ph[(size_t)gc_ * C + c];       \
            }                                                                                            \
            FDV[u] = food[gc_];                                                                          \
        }                                                                                                \
    }
        uint32_t c_cell[ACT_UNROLL], n_cell[ACT_UNROLL];
        int c_ix[ACT_UNROLL], c_iy[ACT_UNROLL], n_ix[ACT_UNROLL], n_iy[ACT_UNROLL];
        float c_pv[ACT_UNROLL][C], n_pv[ACT_UNROLL][C], c_fd[ACT_UNROLL], n_fd[ACT_UNROLL];
        // Each wave owns a CONTIGUOUS run of ants, so its observation rows form one sequential write
        // stream: the partial cache line at the end of a row is completed by the same wave's next row
        // while it is still in L2 (profiles/obs_write_probe.hip: 4.2 -> 4.9 TB/s for this pattern
        // against a run interleaved over the waves).
        const int per = ((N + nwaves - 1) / nwaves + ACT_UNROLL - 1) / ACT_UNROLL * ACT_UNROLL;
        const int i_begin = min(wave * per, N), i_end = min(i_begin + per, N);
        ACT_FETCH(i_begin, c_cell, c_ix, c_iy, c_pv, c_fd)
        for (int i0 = i_begin; i0 < i_end; i0 += ACT_UNROLL) {
            // prefetch the next group (clamped: harmless re-read at the end)
            ACT_FETCH(min(i0 + ACT_UNROLL, N - 1), n_cell, n_ix, n_iy, n_pv, n_fd)
#pragma unroll
            for (int u = 0; u < ACT_UNROLL; ++u) {
                const int i = min(i0 + u, i_end - 1);
                const bool real = (i0 + u < i_end) && lane < PP; // clamped duplicates must not count twice
                const uint32_t cl = c_cell[u];
                const uint32_t wd = cl >> 5, bit = 1u << (cl & 31);
                if (real && explore && !abl_explore && !(L.b_old[wd] & bit)) { // reward_custom.py:19,22 (mask ignored)
                    atomicAdd(&L.cnt[i], 1u);
                    atomicOr(&g_expl[wd], bit); // marks go straight to HBM: every count uses the LDS copy of the pre-step map
                }
                float pvs[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v = c_pv[u][c];
                    if (p.scaled) {
                        v *= g_now;
                        v = v < cut ? 0.0f : v;
                    }
                    pvs[c] = v * inv_max; // :124-125, reciprocal multiply (pheromone channels are held to 1e-5)
                }
                float *dst = obs_env + (size_t)i * row;
                const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
                float *o = stage + mis + qK;
                // bfloat16 observations: the same staging and copy-out on 2-byte elements (8 per 16 bytes)
                uint16_t *dst16 = reinterpret_cast<uint16_t *>(obs) + ((size_t)e * N + (size_t)i) * row;
                const uint32_t mis16 = (uint32_t)(((uintptr_t)dst16 >> 1) & 7);
                uint16_t *o16 = reinterpret_cast<uint16_t *>(stage) + mis16 + qK;
                const float v_ants = (L.b_pres[wd] & bit) ? 1.0f : 0.0f;  // :142
                const float v_area = (area[wd] & bit) ? 1.0f : 0.0f;       // :130-131
                const float v_wall = (walls[wd] & bit) ? 1.0f : 0.0f;      // :128-129
                float v_rock = 0.0f;                                       // :132-135
                if (LAYOUT != LAYOUT_DEFAULT && R > 0) {
                    uint32_t rm = mask_q ? L.rockmask[i] : 0u;
                    bool any = false;
                    while (rm) {
                        const int r = __builtin_ctz(rm);
                        rm &= rm - 1;
                        const double vx = (double)c_ix[u] - L.rock[4 * r + 0];
                        const double vy = (double)c_iy[u] - L.rock[4 * r + 1];
                        any |= vx * vx + vy * vy < L.rock[4 * r + 3]; // == sqrt(d2) < radius, see sqrt_lt_threshold
                    }
                    v_rock = any ? 1.0f : 0.0f;
                }
                const bool m = mask_q; // RL_api.py:147-148: mask*(p+1)-1 == -1 on masked cells
                if (OBS16) { // (only instantiated for the two default layouts)
                    o16[0] = bf16_bits(m ? v_ants : -1.0f); o16[1] = bf16_bits(m ? pvs[0] : -1.0f);
                    o16[2] = bf16_bits(m ? pvs[C - 1] : -1.0f); o16[3] = bf16_bits(m ? v_area : -1.0f);
                    o16[4] = bf16_bits(m ? v_wall : -1.0f); o16[5] = bf16_bits(m ? c_fd[u] : -1.0f);
                    if (LAYOUT == LAYOUT_DEFAULT_ROCKS) o16[6] = bf16_bits(m ? v_rock : -1.0f);
                } else if (LAYOUT == LAYOUT_DEFAULT || LAYOUT == LAYOUT_DEFAULT_ROCKS) {
                    o[0] = m ? v_ants : -1.0f; o[1] = m ? pvs[0] : -1.0f; o[2] = m ? pvs[C - 1] : -1.0f;
                    o[3] = m ? v_area : -1.0f; o[4] = m ? v_wall : -1.0f; o[5] = m ? c_fd[u] : -1.0f;
                    if (LAYOUT == LAYOUT_DEFAULT_ROCKS) o[6] = m ? v_rock : -1.0f;
                } else {
                    for (int k = 0; k < K; ++k) {
                        float v = 0.0f;
                        switch (p.ch_kind[k]) {
                        case ANTSRL_CH_PHERO: {
                            float t = pvs[0];
#pragma unroll
                            for (int c = 1; c < C; ++c) t = (p.ch_arg[k] == c) ? pvs[c] : t;
                            v = t;
                        } break;
                        case ANTSRL_CH_FOOD: v = c_fd[u]; break;      // :126-127
                        case ANTSRL_CH_WALLS: v = v_wall; break;
                        case ANTSRL_CH_ANTHILL: v = v_area; break;
                        case ANTSRL_CH_ANTS: v = v_ants; break;
                        case ANTSRL_CH_ROCKS: v = v_rock; break;
                        default: break;
                        }
                        o[k] = m ? v : -1.0f;
                    }
                }
                wave_lds_sync();
                // copy the row out: two 16-byte stores per lane over the fully-inside float4s of the
                // aligned window [mis, mis+row), one 4-byte store for the <= 6 edge floats; lanes
                // with nothing left repeat a valid store (same address, same data)
                if (OBS16) {
                    // one 16-byte store per lane over the whole 8-element groups of the aligned window
                    // [mis16, mis16 + row) (<= 60 groups), one 2-byte store for the <= 14 edge elements
                    uint16_t *st16 = reinterpret_cast<uint16_t *>(stage), *d_al = dst16 - mis16;
                    const uint32_t g_lo = (mis16 + 7) >> 3, g_hi = (mis16 + row) >> 3;
                    const uint32_t ga = min(g_lo + (uint32_t)lane, g_hi - 1);
                    const uint4 wa = reinterpret_cast<const uint4 *>(st16)[ga];
                    const uint32_t hd16 = 8 * g_lo - mis16, tl16 = mis16 + row - 8 * g_hi;
                    const uint32_t fe16 = (uint32_t)lane < hd16 ? mis16 + lane
                                          : ((uint32_t)lane - hd16 < tl16 ? 8 * g_hi + ((uint32_t)lane - hd16) : mis16);
                    const uint16_t we = st16[fe16];
                    store_stream(reinterpret_cast<uint4 *>(d_al) + ga, wa);
                    store_stream(d_al + fe16, we);
                    wave_lds_sync();
                    continue;
                }
                float *dst_al = dst - mis;
                const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + row) >> 2; // interior float4s [j_lo, j_hi)
                const uint32_t ja = min(j_lo + (uint32_t)lane, j_hi - 1), jb = min(j_lo + 64u + (uint32_t)lane, j_hi - 1);
                const float4 va = reinterpret_cast<const float4 *>(stage)[ja];
                const float4 vb = reinterpret_cast<const float4 *>(stage)[jb];
                const uint32_t hd = 4 * j_lo - mis, tl = mis + row - 4 * j_hi;
                const uint32_t fe = (uint32_t)lane < hd ? mis + lane
                                    : ((uint32_t)lane - hd < tl ? 4 * j_hi + ((uint32_t)lane - hd) : mis);
                const float ve = stage[fe];
                store_stream(reinterpret_cast<float4 *>(dst_al) + ja, va);
                store_stream(reinterpret_cast<float4 *>(dst_al) + jb, vb);
                store_stream(dst_al + fe, ve);
                wave_lds_sync();
            }
#pragma unroll
            for (int u = 0; u < ACT_UNROLL; ++u) {
                c_cell[u] = n_cell[u]; c_ix[u] = n_ix[u]; c_iy[u] = n_iy[u]; c_fd[u] = n_fd[u];
#pragma unroll
                for (int c = 0; c < C; ++c) c_pv[u][c] = n_pv[u][c];
            }
        }
#undef ACT_FETCH
    }
    // A perception of more than 64 cells takes several passes over ONE staging row per wave, so such a
    // wave works on a single ant at a time (the second slot of the group stays empty).
    const int ustep = npass > 1 ? 1 : ACT_UNROLL;
    if (!FAST)
    for (int i0 = wave * ustep; i0 < ((flags & ACT_ABL_NO_ITEMS) ? 0 : N); i0 += nwaves * ustep) {
        for (int pass = 0; pass < npass; ++pass) {
            const int q = pass * 64 + lane;
            const bool lane_on = q < PP;
            const CellOff of = L.off[lane_on ? q : 0];
            const bool mask_q = lane_on && L.t_mask[lane_on ? q : 0];
            uint32_t cell[ACT_UNROLL];
            int ixv[ACT_UNROLL], iyv[ACT_UNROLL];
            bool valid[ACT_UNROLL], vis[ACT_UNROLL];
            float pv[ACT_UNROLL][C];
            float fd[ACT_UNROLL];
#pragma unroll
            for (int u = 0; u < ACT_UNROLL; ++u) {
                const int i = i0 + u;
                valid[u] = lane_on && i < N && u < ustep;
                const AntFrame fr = L.frame[i < N ? i : 0]; // wave-uniform address: LDS broadcast
                const double rx = fr.ct * of.px - fr.st * of.py; // RL_api.py:110-111
                const double ry = fr.st * of.px + fr.ct * of.py;
                int ix = (int)rint(rx + fr.cx), iy = (int)rint(ry + fr.cy); // :114-117 (half to even)
                if (wrap_fast) {                                             // :118-119
                    ix += ix < 0 ? W : 0; ix -= ix >= W ? W : 0;
                    iy += iy < 0 ? H : 0; iy -= iy >= H ? H : 0;
                } else {
                    ix = wrap_index(ix, W); iy = wrap_index(iy, H);
                }
                ixv[u] = ix; iyv[u] = iy;
                cell[u] = (uint32_t)(ix * H + iy);
                vis[u] = valid[u] && mask_q && obs_env;
            }
            // issue every global gather before anything consumes one
#pragma unroll
            for (int u = 0; u < ACT_UNROLL; ++u) {
                fd[u] = 0.0f;
#pragma unroll
                for (int c = 0; c < C; ++c) pv[u][c] = 0.0f;
                if (vis[u] && !abl_gather) {
                    if (C == 2) {
                        const float2 t = *reinterpret_cast<const float2 *>(ph + (size_t)cell[u] * 2);
                        pv[u][0] = t.x; pv[u][C - 1] = t.y;
                    } else {
#pragma unroll
                        for (int c = 0; c < C; ++c) pv[u][c] = ph[(size_t)cell[u] * C + c];
                    }
                    fd[u] = food[cell[u]];
                }
            }
            if (p.scaled) {
#pragma unroll
                for (int u = 0; u < ACT_UNROLL; ++u)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float v = pv[u][c] * g_now;
                        pv[u][c] = v < cut ? 0.0f : v;
                    }
            }
#pragma unroll
            for (int u = 0; u < ACT_UNROLL; ++u) {
                const int i = i0 + u;
                if (i >= N || u >= ustep) break; // wave-uniform
                const uint32_t cl = cell[u];
                const uint32_t wd = cl >> 5, bit = 1u << (cl & 31);
                if (valid[u] && explore && !abl_explore && !(L.b_old[wd] & bit)) { // reward_custom.py:19,22
                    atomicAdd(&L.cnt[i], 1u);                                       // (mask ignored)
                    atomicOr(&g_expl[wd], bit); // marks go straight to HBM: every count uses the LDS copy of the pre-step map
                }
                if (!obs_env) continue;
                // destination row of this ant; the staging image is shifted by the row's misalignment
                // so that 16-byte LDS reads line up with 16-byte global stores
                float *dst = obs_env + (size_t)i * row;
                const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
                if (valid[u]) {
                    float *o = stage + mis + (uint32_t)q * K;
                    const float v_ants = (L.b_pres[wd] & bit) ? 1.0f : 0.0f;  // :142
                    const float v_area = (area[wd] & bit) ? 1.0f : 0.0f;       // :130-131
                    const float v_wall = (walls[wd] & bit) ? 1.0f : 0.0f;      // :128-129
                    float v_rock = 0.0f;                                       // :132-135
                    if (LAYOUT != LAYOUT_DEFAULT && R > 0) {
                        uint32_t rm = vis[u] ? L.rockmask[i] : 0u;
                        bool any = false;
                        while (rm) {
                            const int r = __builtin_ctz(rm);
                            rm &= rm - 1;
                            const double vx = (double)ixv[u] - L.rock[4 * r + 0];
                            const double vy = (double)iyv[u] - L.rock[4 * r + 1];
                            any |= vx * vx + vy * vy < L.rock[4 * r + 3]; // == sqrt(d2) < radius
                        }
                        v_rock = any ? 1.0f : 0.0f;
                    }
                    const bool m = vis[u]; // RL_api.py:147-148: mask*(p+1)-1 == -1 on masked cells
                    if (LAYOUT == LAYOUT_DEFAULT || LAYOUT == LAYOUT_DEFAULT_ROCKS) {
                        // phero/max_val (:124-125) as a multiply by the f32 reciprocal: the
                        // pheromone channels are held to 1e-5, not bit-exactness (fp32 grid)
                        o[0] = m ? v_ants : -1.0f; o[1] = m ? pv[u][0] * inv_max : -1.0f;
                        o[2] = m ? pv[u][C - 1] * inv_max : -1.0f;
                        o[3] = m ? v_area : -1.0f; o[4] = m ? v_wall : -1.0f; o[5] = m ? fd[u] : -1.0f;
                        if (LAYOUT == LAYOUT_DEFAULT_ROCKS) o[6] = m ? v_rock : -1.0f;
                    } else {
                        for (int k = 0; k < K; ++k) {
                            float v = 0.0f;
                            switch (p.ch_kind[k]) {
                            case ANTSRL_CH_PHERO: {
                                float t = pv[u][0];
#pragma unroll
                                for (int c = 1; c < C; ++c) t = (p.ch_arg[k] == c) ? pv[u][c] : t;
                                v = t * inv_max;
                            } break;
                            case ANTSRL_CH_FOOD: v = fd[u]; break;      // :126-127
                            case ANTSRL_CH_WALLS: v = v_wall; break;
                            case ANTSRL_CH_ANTHILL: v = v_area; break;
                            case ANTSRL_CH_ANTS: v = v_ants; break;
                            case ANTSRL_CH_ROCKS: v = v_rock; break;
                            default: break;
                            }
                            o[k] = m ? v : -1.0f;
                        }
                    }
                }
                if (pass == npass - 1) { // the ant's row is complete: copy it out
                    wave_lds_sync();
                    if (!abl_store) {
                        float *dst_al = dst - mis; // 16-byte aligned window [mis, mis + row)
                        const uint32_t n4 = (mis + row + 3) >> 2;
                        for (uint32_t j = lane; j < n4; j += 64) {
                            const float4 v = reinterpret_cast<const float4 *>(stage)[j];
                            const uint32_t lo = 4 * j;
                            if (lo >= mis && lo + 3 < mis + row) {
                                store_stream(reinterpret_cast<float4 *>(dst_al) + j, v);
                            } else {
                                if (lo + 0 >= mis && lo + 0 < mis + row) store_stream(dst_al + lo + 0, v.x);
                                if (lo + 1 >= mis && lo + 1 < mis + row) store_stream(dst_al + lo + 1, v.y);
                                if (lo + 2 >= mis && lo + 2 < mis + row) store_stream(dst_al + lo + 2, v.z);
                                if (lo + 3 >= mis && lo + 3 < mis + row) store_stream(dst_al + lo + 3, v.w);
                            }
                        }
                    }
                    wave_lds_sync();
                }
            }
        }
    }
    __syncthreads();
    act_trace(flags, e, tid, 2);

    // ---- phase 4: agent_state (RL_api.py:160-162), reward.observation hooks, give_reward
    for (int i = tid; i < N; i += T) {
        const float hold = p.s.holding[eN + i];
        if (agent_state) {
            store_stream(agent_state + (eN + i) * 2 + 0, hold);
            store_stream(agent_state + (eN + i) * 2 + 1, p.s.seed[eN + i]);
        }
        double rw = 0.0;
        if (p.reward_kind != ANTSRL_REWARD_NONE) {
            // first observation after Reward.setup sees delta-holding == 0 (alias quirk,
            // reward_custom.py:35,68 — see oracle/antsrl_oracle.c)
            const float prev_h = primed0 ? p.s.prev_holding[eN + i] : hold;
            const double dh = (double)hold - (double)prev_h;
            if (p.reward_kind == ANTSRL_REWARD_EXPLORATION) {
                rw = (double)L.cnt[i] / 10.0; // reward_custom.py:19
            } else if (p.reward_kind == ANTSRL_REWARD_FOOD) {
                rw = dh < 0 ? 10.0 : dh; // reward_custom.py:38-39
                p.s.prev_holding[eN + i] = hold;
            } else { // All_Rewards, reward_custom.py:79-106
                const double r_food = dh < 0 ? 0.0 : dh;
                const double r_anthill = dh < 0 ? 1.0 : 0.0;
                p.s.prev_holding[eN + i] = hold;
                if (explore) {
                    double re = (double)L.cnt[i] / 10.0;
                    re = (hold == 0.0f) ? re * p.fct_explore : re * p.fct_explore_holding;
                    rw += re;
                }
                const double dx = p.s.x[eN + i] - (double)p.s.anthill_xyr[3 * e + 0];
                const double dy = p.s.y[eN + i] - (double)p.s.anthill_xyr[3 * e + 1];
                const double nd = sqrt(dx * dx + dy * dy);
                const double heading = (double)((p.s.prev_dist[eN + i] > nd) && (hold > 0.0f)) * 0.1;
                p.s.prev_dist[eN + i] = nd;
                rw += r_food * p.fct_food + r_anthill * p.fct_anthill + heading * p.fct_heading;
            }
        }
        if (reward) store_stream(reward + eN + i, (float)rw);
        if (do_step && rw - p.reward_threshold > 0) p.s.reward_state[eN + i] = 255; // ants.py:119-121
    }
    if (tid == 0) {
        if (p.reward_kind != ANTSRL_REWARD_NONE) p.s.reward_primed[e] = 1;
        if (do_step && done) done[e] = (uint8_t)(p.max_time == p.s.timestep[e]); // RL_api.py:200
    }
    act_trace(flags, e, tid, 3);
    if (flags & ACT_FUSED_UPDATE) {
        // Environment.update of the same step (main.py:131) in the same launch: the staging
        // region is dead after phase 3 and doubles as the update's scratch.
        __syncthreads();
        update_env<C>(p, e, wall_jitter, out_buf, (unsigned char *)L.hkeys);
    }
}

// ===================================================================================
// k_update — Environment.update minus the pheromone sweep (environment.py:42-47):
// Walls (walls.py:22-28), CircleObstacles (circle_obstacles.py:32-58), Ants.update
// (ants.py:123-130), Anthill.update (anthill.py:41-46, sparse form).
// one workgroup per environment; `out_buf` = pheromone buffer the sweep just wrote.
// ===================================================================================
__device__ __forceinline__ uint32_t block_excl_scan_flag(bool flag, uint32_t *wave_tot, int lane, int wave,
                                                         int nwaves, uint32_t *block_total)
{
    const unsigned long long m = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return off + in_wave;
}

// The update phases of ONE environment, run by the whole workgroup.  `smem` is
// update_scratch_bytes() of LDS.  Called by k_update and, fused, at the tail of k_act.
template <int C>
__device__ __forceinline__ void update_env(const KP &p, const int e, const double *__restrict__ wall_jitter,
                                           const int out_buf, unsigned char *smem)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const int N = p.N, W = p.W, H = p.H, R = p.R;
    const size_t G = (size_t)W * H, eN = (size_t)e * N;
    uint32_t *hkeys = (uint32_t *)smem, *hvals = hkeys + p.HT;
    double *rk = (double *)(smem + align_up(8 * (size_t)p.HT, 16)); // [2R] new rock centres
    double *red = rk + 2 * (R > 0 ? R : 1);                          // [nwaves] reduction scratch
    uint32_t *wave_tot = (uint32_t *)(red + nwaves);                 // [nwaves]

    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    float *food = p.s.food + (size_t)e * G;
    float *out = p.s.phero[out_buf] + (size_t)e * G * C;
    const int ts = p.s.timestep[e] + 1; // environment.py:45

    for (int h = tid; h < p.HT; h += T) {
        hkeys[h] = HASH_EMPTY;
        hvals[h] = 0u;
    }

    // ---- Walls.update, walls.py:25-28
    uint32_t carry = 0;
    for (int base = 0; base < N; base += T) {
        const int i = base + tid;
        bool hit = false;
        if (i < N) hit = test_bit(walls, (uint32_t)((int)p.s.x[eN + i] * H + (int)p.s.y[eN + i]));
        double u = 0.0;
        if (wall_jitter) { // k-th colliding ant (index order) takes the k-th draw
            uint32_t tot;
            const uint32_t rank = carry + block_excl_scan_flag(hit, wave_tot, lane, wave, nwaves, &tot);
            carry += tot;
            if (hit) u = wall_jitter[eN + rank];
        } else if (hit) {
            u = jitter_u01(p.rng_seed, (uint32_t)e, (uint32_t)ts, (uint32_t)i);
        }
        if (hit) {
            p.s.x[eN + i] = p.s.prev_x[eN + i];
            p.s.y[eN + i] = p.s.prev_y[eN + i];
            p.s.theta[eN + i] += u - 0.5; // theta is NOT re-wrapped here
        }
    }
    __syncthreads();

    // ---- CircleObstacles.update, circle_obstacles.py:35-58
    if (R > 0) {
        // pass 1: centres -= sum_over_ants(push)/weight.  numpy sums the ants sequentially in
        // index order; non-colliding ants contribute exact zeros, so adding only the
        // colliding ones in index order reproduces the float64 result bit for bit.
        for (int q = wave; q < R; q += nwaves) {
            const double cx = p.s.rock_cx[(size_t)e * R + q], cy = p.s.rock_cy[(size_t)e * R + q];
            const double rad = p.s.rock_r[(size_t)e * R + q];
            const double rad2_hi = rad * rad * (1.0 + 1e-12) + 1e-300; // d2 above this: sqrt(d2) > rad for sure
            double sx = 0.0, sy = 0.0;
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                double px = 0.0, py = 0.0;
                bool col = false;
                if (i < N) {
                    const double vx = cx - p.s.x[eN + i], vy = cy - p.s.y[eN + i];
                    const double d2 = vx * vx + vy * vy;
                    // only colliding ants contribute: the sqrt and the division are spent on the few
                    // lanes a conservative squared-distance test lets through (the exact test
                    // `!(d > rad)` of the reference then decides)
                    if (!(d2 > rad2_hi)) {
                        const double d = sqrt(d2);
                        const double f = 1 - rad / (d + 0.001);
                        px = vx * f; py = vy * f;
                        col = !(d > rad);
                    }
                }
                unsigned long long m = __ballot(col);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    sx += __shfl(px, l);
                    sy += __shfl(py, l);
                }
            }
            if (lane == 0) {
                const double wgt = p.s.rock_w[(size_t)e * R + q];
                rk[2 * q + 0] = cx - sx / wgt;
                rk[2 * q + 1] = cy - sy / wgt;
            }
        }
        __syncthreads();
        for (int q = tid; q < R; q += T) {
            p.s.rock_cx[(size_t)e * R + q] = rk[2 * q + 0];
            p.s.rock_cy[(size_t)e * R + q] = rk[2 * q + 1];
        }
        // pass 2 (:53-58): ants pushed out of the UPDATED rocks, then warp_xy
        for (int i = tid; i < N; i += T) {
            const double x = p.s.x[eN + i], y = p.s.y[eN + i];
            double sx = 0.0, sy = 0.0;
            for (int q = 0; q < R; ++q) {
                const double vx = rk[2 * q + 0] - x, vy = rk[2 * q + 1] - y;
                const double rad = p.s.rock_r[(size_t)e * R + q];
                const double d2 = vx * vx + vy * vy;
                if (d2 > rad * rad * (1.0 + 1e-12) + 1e-300) continue; // adds an exact +0.0: skip the sqrt and division
                const double d = sqrt(d2);
                const double f = 1 - rad / (d + 0.001);
                double px = vx * f, py = vy * f;
                if (d > rad) { px = 0.0; py = 0.0; }
                sx += px; sy += py;
            }
            p.s.x[eN + i] = warp_coord(x + sx, (double)W);
            p.s.y[eN + i] = warp_coord(y + sy, (double)H);
        }
        __syncthreads();
    }

    // ---- Ants.update, ants.py:123-130: prev := cur; deposit (pheromone.py:36-41)
    for (int i = tid; i < N; i += T) {
        const double x = p.s.x[eN + i], y = p.s.y[eN + i];
        p.s.prev_x[eN + i] = x;
        p.s.prev_y[eN + i] = y;
        lww_insert(hkeys, hvals, (uint32_t)p.HT - 1, (uint32_t)((int)x * H + (int)y), (uint32_t)i);
        p.s.reward_state[eN + i] = (uint8_t)((double)p.s.reward_state[eN + i] * 0.9); // :130
    }
    __syncthreads();
    if (p.scaled) {
        // A deposit that landed on a WALL cell in the previous update is visible to exactly one
        // observation and is zeroed by this update's Walls pass (walls.py:30): clear it now.
        for (int i = tid; i < N; i += T) {
            const int32_t wc = p.s.walldep_cell[eN + i];
            if (wc >= 0) {
                for (int c = 0; c < C; ++c) out[(size_t)wc * C + c] = 0.0f;
                p.s.walldep_cell[eN + i] = -1;
            }
        }
        __syncthreads();
    }
    double gain = 0.0;
    for (int i = tid; i < N; i += T) {
        const uint32_t cell = (uint32_t)((int)p.s.x[eN + i] * H + (int)p.s.y[eN + i]);
        if (lww_winner(hkeys, hvals, (uint32_t)p.HT - 1, cell) == (uint32_t)i) {
            if (!p.scaled) {
                for (int c = 0; c < C; ++c) {
                    const float a = p.s.activation[(eN + i) * C + c];
                    if (a != 0.0f) {
                        float v = out[(size_t)cell * C + c] + a;
                        if (p.has_max_val) v = fminf(v, (float)p.max_val);
                        out[(size_t)cell * C + c] = v;
                    }
                }
            } else {
                // grid holds u = v / f0^S_write.  Value after this update's (conceptual) sweep:
                // v = u * f0^(S+1), zero below the cut (the per-step cut is monotone, so testing the
                // current value equals testing every intermediate one); wall cells hold no
                // pheromone at deposit time (walls.py:30).
                const bool on_wall = test_bit(walls, cell);
                bool wrote = false;
                for (int c = 0; c < C; ++c) {
                    const float a = p.s.activation[(eN + i) * C + c];
                    if (a != 0.0f) {
                        double v = (double)out[(size_t)cell * C + c] * p.g_dep;
                        if (v < p.threshold || on_wall) v = 0.0;
                        v += (double)a;
                        if (p.has_max_val) v = fmin(v, p.max_val);
                        out[(size_t)cell * C + c] = (float)(v * p.inv_g_dep);
                        wrote = true;
                    } else if (on_wall) {
                        out[(size_t)cell * C + c] = 0.0f;
                    }
                }
                if (on_wall && wrote) p.s.walldep_cell[eN + i] = (int32_t)cell;
            }
        }
        // ---- Anthill.update (anthill.py:41-46), sparse: after the first full collect the only
        // non-zero food on the area is what this step's exchange winners wrote there.
        const int32_t dc = p.s.dirty_cell[eN + i];
        if (dc >= 0) {
            gain += (double)food[dc];
            food[dc] = 0.0f;
            p.s.dirty_cell[eN + i] = -1;
        }
    }
    // block sum of gain (integer-valued in every reference workload -> order-independent)
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if (lane == 0) red[wave] = gain;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += red[w];
        if (s != 0.0) p.s.anthill_food[e] += s;
        p.s.timestep[e] = ts;
    }
}

template <int C>
__global__ void __launch_bounds__(1024)
k_update(const KP p, const double *__restrict__ wall_jitter, const int out_buf)
{
    extern __shared__ __align__(16) unsigned char smem[];
    update_env<C>(p, blockIdx.x, wall_jitter, out_buf, smem);
}

// LDS of k_update_one: hash [HT] keys + values, rock table [R][4] + new centres [R][2], reduction and
// scan scratch per wave, ants' x / y [N].
__host__ __device__ inline size_t update_one_lds_bytes(int HT, int R, int nwaves, int N)
{
    return align_up(8 * (size_t)HT, 16) + 48 * (size_t)(R > 0 ? R : 1) + 8 * (size_t)nwaves +
           align_up(4 * (size_t)nwaves, 8) + 16 * (size_t)N + 16;
}

// k_update for N <= 1024: ONE ant per thread.  Same phases and the same arithmetic as update_env,
// restructured around latency: every independent global load of the ant (position, previous position,
// reward tint, activation, the two sparse-update cells) is issued before anything waits; the ant's
// state then stays in registers across the phases, the rock pass reads all ants' positions from LDS,
// and each array is written back once.  update_env re-reads x / y from HBM in every phase (a dependent
// round trip behind each barrier) — it remains the path for N > 1024 and for the fused launch.
template <int C>
__global__ void __launch_bounds__(1024)
k_update_one(const KP p, const double *__restrict__ wall_jitter, const int out_buf)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int e = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = T >> 6;
    const int N = p.N, W = p.W, H = p.H, R = p.R;
    const size_t G = (size_t)W * H, eN = (size_t)e * N;
    uint32_t *hkeys = (uint32_t *)smem, *hvals = hkeys + p.HT;
    double *rock = (double *)(smem + align_up(8 * (size_t)p.HT, 16)); // [R][4] cx, cy, radius, weight
    double *rk = rock + 4 * (R > 0 ? R : 1);                          // [R][2] new centres
    double *red = rk + 2 * (R > 0 ? R : 1);                           // [nwaves]
    uint32_t *wave_tot = (uint32_t *)(red + nwaves);                  // [nwaves]
    double *sx = (double *)((unsigned char *)wave_tot + align_up(4 * (size_t)nwaves, 8)), *sy = sx + N;

    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    float *food = p.s.food + (size_t)e * G;
    float *out = p.s.phero[out_buf] + (size_t)e * G * C;
    const bool on = tid < N;
    const size_t a = eN + (on ? tid : 0);

    // ---- every independent load first
    double x = p.s.x[a], y = p.s.y[a];
    const double px0 = p.s.prev_x[a], py0 = p.s.prev_y[a]; // used on a wall hit only; prefetched all the same
    const uint8_t rstate = p.s.reward_state[a];
    float act[C];
#pragma unroll
    for (int c = 0; c < C; ++c) act[c] = p.s.activation[a * C + c];
    // (unconditional loads on the clamped index + selects: a load inside a per-lane branch is followed by
    // its own s_waitcnt vmcnt(0))
    const int32_t wc_l = p.s.walldep_cell[a], dc_l = p.s.dirty_cell[a];
    const int32_t wc = (p.scaled && on) ? wc_l : -1;
    const int32_t dc = on ? dc_l : -1;
    const int ts = p.s.timestep[e] + 1; // environment.py:45
    if (tid < R) {
        rock[4 * tid + 0] = p.s.rock_cx[(size_t)e * R + tid];
        rock[4 * tid + 1] = p.s.rock_cy[(size_t)e * R + tid];
        rock[4 * tid + 2] = p.s.rock_r[(size_t)e * R + tid];
        rock[4 * tid + 3] = p.s.rock_w[(size_t)e * R + tid];
    }
    for (int h = tid; h < p.HT; h += T) {
        hkeys[h] = HASH_EMPTY;
        hvals[h] = 0u;
    }
    const float food_dc = dc >= 0 ? food[dc] : 0.0f; // nobody else touches a dirty cell during the update

    // ---- Walls.update, walls.py:25-28
    const bool hit = on && test_bit(walls, (uint32_t)((int)x * H + (int)y));
    double u = 0.0;
    if (wall_jitter) { // k-th colliding ant (index order) takes the k-th draw
        uint32_t tot;
        const uint32_t rank = block_excl_scan_flag(hit, wave_tot, lane, wave, nwaves, &tot);
        if (hit) u = wall_jitter[eN + rank];
    } else if (hit) {
        u = jitter_u01(p.rng_seed, (uint32_t)e, (uint32_t)ts, (uint32_t)tid);
    }
    bool moved = hit;
    if (hit) {
        x = px0;
        y = py0;
        p.s.theta[a] += u - 0.5; // theta is NOT re-wrapped here
    }

    // ---- CircleObstacles.update, circle_obstacles.py:35-58
    if (R > 0) {
        if (on) {
            sx[tid] = x;
            sy[tid] = y;
        }
        __syncthreads();
        // pass 1: centres -= sum_over_ants(push)/weight, ants summed in index order (colliding ones only:
        // the others contribute exact zeros)
        for (int q = wave; q < R; q += nwaves) {
            const double cx = rock[4 * q + 0], cy = rock[4 * q + 1], rad = rock[4 * q + 2];
            const double rad2_hi = rad * rad * (1.0 + 1e-12) + 1e-300; // d2 above this: sqrt(d2) > rad for sure
            double sumx = 0.0, sumy = 0.0;
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                double px = 0.0, py = 0.0;
                bool col = false;
                if (i < N) {
                    const double vx = cx - sx[i], vy = cy - sy[i];
                    const double d2 = vx * vx + vy * vy;
                    if (!(d2 > rad2_hi)) {
                        const double d = sqrt(d2);
                        const double f = 1 - rad / (d + 0.001);
                        px = vx * f; py = vy * f;
                        col = !(d > rad);
                    }
                }
                unsigned long long m = __ballot(col);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    sumx += __shfl(px, l);
                    sumy += __shfl(py, l);
                }
            }
            if (lane == 0) {
                rk[2 * q + 0] = cx - sumx / rock[4 * q + 3];
                rk[2 * q + 1] = cy - sumy / rock[4 * q + 3];
            }
        }
        __syncthreads();
        if (tid < R) {
            p.s.rock_cx[(size_t)e * R + tid] = rk[2 * tid + 0];
            p.s.rock_cy[(size_t)e * R + tid] = rk[2 * tid + 1];
        }
        // pass 2 (:53-58): ants pushed out of the UPDATED rocks, then warp_xy
        double ax = 0.0, ay = 0.0;
        for (int q = 0; q < R; ++q) {
            const double vx = rk[2 * q + 0] - x, vy = rk[2 * q + 1] - y;
            const double rad = rock[4 * q + 2];
            const double d2 = vx * vx + vy * vy;
            if (d2 > rad * rad * (1.0 + 1e-12) + 1e-300) continue; // adds an exact +0.0
            const double d = sqrt(d2);
            const double f = 1 - rad / (d + 0.001);
            double px = vx * f, py = vy * f;
            if (d > rad) { px = 0.0; py = 0.0; }
            ax += px; ay += py;
        }
        x = warp_coord(x + ax, (double)W);
        y = warp_coord(y + ay, (double)H);
        moved = true;
    }

    // ---- Ants.update, ants.py:123-130: prev := cur; deposit (pheromone.py:36-41)
    __syncthreads(); // the hash table is initialised (R == 0 and library jitter: no barrier so far)
    const uint32_t cell = (uint32_t)((int)x * H + (int)y);
    if (on) {
        if (moved) {
            p.s.x[a] = x;
            p.s.y[a] = y;
        }
        p.s.prev_x[a] = x;
        p.s.prev_y[a] = y;
        lww_insert(hkeys, hvals, (uint32_t)p.HT - 1, cell, (uint32_t)tid);
        p.s.reward_state[a] = (uint8_t)((double)rstate * 0.9); // :130
    }
    __syncthreads();
    if (p.scaled) {
        // a deposit that landed on a WALL cell in the previous update is zeroed by this update's Walls
        // pass (walls.py:30), before this update's deposits
        if (wc >= 0) {
            for (int c = 0; c < C; ++c) out[(size_t)wc * C + c] = 0.0f;
            p.s.walldep_cell[a] = -1;
        }
        __syncthreads();
    }
    double gain = 0.0;
    if (on) {
        if (lww_winner(hkeys, hvals, (uint32_t)p.HT - 1, cell) == (uint32_t)tid) {
            if (!p.scaled) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (act[c] != 0.0f) {
                        float v = out[(size_t)cell * C + c] + act[c];
                        if (p.has_max_val) v = fminf(v, (float)p.max_val);
                        out[(size_t)cell * C + c] = v;
                    }
                }
            } else { // scaled units, see update_env
                const bool on_wall = test_bit(walls, cell);
                bool wrote = false;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (act[c] != 0.0f) {
                        double v = (double)out[(size_t)cell * C + c] * p.g_dep;
                        if (v < p.threshold || on_wall) v = 0.0;
                        v += (double)act[c];
                        if (p.has_max_val) v = fmin(v, p.max_val);
                        out[(size_t)cell * C + c] = (float)(v * p.inv_g_dep);
                        wrote = true;
                    } else if (on_wall) {
                        out[(size_t)cell * C + c] = 0.0f;
                    }
                }
                if (on_wall && wrote) p.s.walldep_cell[a] = (int32_t)cell;
            }
        }
        // ---- Anthill.update (anthill.py:41-46), sparse form
        if (dc >= 0) {
            gain = (double)food_dc;
            food[dc] = 0.0f;
            p.s.dirty_cell[a] = -1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if (lane == 0) red[wave] = gain;
    __syncthreads();
    if (tid == 0) {
        double s2 = 0.0;
        for (int w = 0; w < nwaves; ++w) s2 += red[w];
        if (s2 != 0.0) p.s.anthill_food[e] += s2;
        p.s.timestep[e] = ts;
    }
}

// Anthill.update over the WHOLE grid (anthill.py:41-46): needed on the first update after a
// reset (food may lie on the anthill area) and after two steps without an update.
__global__ void __launch_bounds__(256) k_collect_full(const KP p)
{
    __shared__ double red[4];
    const int e = blockIdx.x, tid = threadIdx.x;
    const size_t G = (size_t)p.W * p.H;
    float *food = p.s.food + (size_t)e * G;
    const uint32_t *area = p.s.area_bits + (size_t)e * p.words;
    double gain = 0.0;
    for (size_t g = tid; g < G; g += blockDim.x)
        if (test_bit(area, (uint32_t)g)) {
            gain += (double)food[g];
            food[g] = 0.0f;
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
__global__ void __launch_bounds__(256) k_phero_wall_clear(const KP p)
{
    const size_t G = (size_t)p.W * p.H, n = (size_t)p.E * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / G, g = i - e * G;
        if (test_bit(p.s.walls_bits + e * p.words, (uint32_t)g))
            for (int c = 0; c < p.C; ++c) p.s.phero[0][i * p.C + c] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) k_phero_renorm(const KP p)
{
    // u := materialised value (units of f0^0); the host then restarts S at 0.
    const size_t n = (size_t)p.E * p.W * p.H * p.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)p.s.phero[0][i] * p.g_now;
        p.s.phero[0][i] = v < p.threshold ? 0.0f : (float)v;
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
k_sweep_march(const KP p, const float *__restrict__ in, float *__restrict__ out, const int seg_rows)
{
    constexpr int S = 2 * R + 1, OUTW = 64 - 2 * R, NA = 2 * S - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y, W = p.W, H = p.H;
    const int strip = blockIdx.x * 4 + wave;
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
    if (strip * OUTW >= H) return; // whole wave (no further barriers)
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
    const int x_lo = blockIdx.z * seg_rows, x_hi = min(x_lo + seg_rows, W);
    // one block of S rows is always in flight ahead of the block being accumulated
    float nv[S][C];
    uint32_t nword[S], ncell[S];
#define MARCH_LOAD(XI0)                                                                              \
    {                                                                                                \
        _Pragma("unroll") for (int s = 0; s < S; ++s)                                                \
        {                                                                                            \
            const int xc = min(max((XI0) + s, 0), W - 1); /* clamped; masked to zero below */        \
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
        MARCH_LOAD(xi0 + S) // prefetch (clamped addresses: a harmless re-read past the end)
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

// ===================================================================================
// reset / state I/O (not on the hot path)
// ===================================================================================
__global__ void k_reset_ants(const KP p, const double *__restrict__ xyt, const double *__restrict__ seed)
{
    const size_t n = (size_t)p.E * p.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / p.N;
        const double x = warp_coord(xyt[3 * i + 0], (double)p.W); // ants.py:27-30
        const double y = warp_coord(xyt[3 * i + 1], (double)p.H);
        p.s.x[i] = x; p.s.y[i] = y; p.s.theta[i] = xyt[3 * i + 2];
        p.s.prev_x[i] = x; p.s.prev_y[i] = y;
        p.s.holding[i] = 0.0f; p.s.prev_holding[i] = 0.0f;
        p.s.mandibles[i] = 0; p.s.reward_state[i] = 0;
        p.s.seed[i] = (float)seed[i];
        p.s.dirty_cell[i] = -1;
        p.s.walldep_cell[i] = -1;
        for (int c = 0; c < p.C; ++c) p.s.activation[i * p.C + c] = 0.0f;
        const double dx = x - (double)p.s.anthill_xyr[3 * e + 0], dy = y - (double)p.s.anthill_xyr[3 * e + 1];
        p.s.prev_dist[i] = sqrt(dx * dx + dy * dy); // reward_custom.py:77
    }
}

__global__ void k_reset_env(const KP p, const int32_t *__restrict__ xyr, const double *__restrict__ rocks)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    for (int j = 0; j < 3; ++j) p.s.anthill_xyr[3 * e + j] = xyr[3 * e + j];
    p.s.anthill_food[e] = 0.0;
    p.s.timestep[e] = 1; // environment.py:27
    p.s.reward_primed[e] = 0;
    for (int q = 0; q < p.R; ++q) {
        p.s.rock_cx[(size_t)e * p.R + q] = rocks[((size_t)e * p.R + q) * 4 + 0];
        p.s.rock_cy[(size_t)e * p.R + q] = rocks[((size_t)e * p.R + q) * 4 + 1];
        p.s.rock_r[(size_t)e * p.R + q] = rocks[((size_t)e * p.R + q) * 4 + 2];
        p.s.rock_w[(size_t)e * p.R + q] = rocks[((size_t)e * p.R + q) * 4 + 3];
    }
}

// bit-pack walls, rasterise the anthill disc (anthill.py:28-33), clear the explored map
__global__ void k_reset_bits(const KP p, const uint8_t *__restrict__ walls, const int32_t *__restrict__ xyr)
{
    const size_t n = (size_t)p.E * p.words;
    const size_t G = (size_t)p.W * p.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / p.words, w = i - e * p.words;
        const long ax = xyr[3 * e + 0], ay = xyr[3 * e + 1], ar = xyr[3 * e + 2];
        uint32_t wb = 0, ab = 0;
        for (int b = 0; b < 32; ++b) {
            const size_t cell = w * 32 + b;
            if (cell >= G) break;
            if (walls[e * G + cell]) wb |= 1u << b;
            const long x = (long)(cell / p.H), y = (long)(cell % p.H);
            // integer ax, ay, r: sqrt(d2) <= r  <=>  d2 <= r*r  (and r < 0 -> empty)
            if (ar >= 0 && (ax - x) * (ax - x) + (ay - y) * (ay - y) <= ar * ar) ab |= 1u << b;
        }
        p.s.walls_bits[i] = wb;
        p.s.area_bits[i] = ab;
        p.s.explored_bits[i] = 0u;
    }
}

__global__ void k_reset_grids(const KP p, const float *__restrict__ food, const float *__restrict__ phero)
{
    const size_t G = (size_t)p.W * p.H, n = (size_t)p.E * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / G, g = i - e * G;
        p.s.food[i] = food[i];
        for (int c = 0; c < p.C; ++c) {
            const float v = phero ? phero[(e * p.C + c) * G + g] : 0.0f;
            p.s.phero[0][i * p.C + c] = v;
            p.s.phero[1][i * p.C + c] = v;
        }
    }
}

// ---- device-side episode generator (antsrl_generate) ----------------------
// Draw `idx` of stream `tag` of environment `env`; the oracle (oracle_gen_u01) is bit-identical.
#define GEN_SALT 0x6A09E667F3BCC909ULL
enum { GEN_ANTHILL = 0, GEN_WALLS = 1, GEN_FOOD = 2, GEN_ROCKS = 3, GEN_ANT_ANGLE = 4, GEN_ANT_DIST = 5,
       GEN_ANT_THETA = 6, GEN_ANT_SEED = 7 };
__device__ __forceinline__ double gen_u01(uint64_t seed, uint32_t env, uint32_t tag, uint32_t idx)
{
    return jitter_u01(seed ^ GEN_SALT, env, tag, idx);
}

// per env: anthill (environment_generator.py:60-63), rocks (:77-85), food discs (map_generators.py:37-40)
__global__ void k_gen_env(const KP p, const AntsGen g, const uint64_t seed)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const int W = p.W, H = p.H, m = W < H ? W : H;
    p.s.anthill_xyr[3 * e + 0] = (int)(gen_u01(seed, e, GEN_ANTHILL, 0) * W * 0.5 + W * 0.25);
    p.s.anthill_xyr[3 * e + 1] = (int)(gen_u01(seed, e, GEN_ANTHILL, 1) * H * 0.5 + H * 0.25);
    p.s.anthill_xyr[3 * e + 2] = (int)(gen_u01(seed, e, GEN_ANTHILL, 2) * m * 0.05 + m * 0.05);
    p.s.anthill_food[e] = 0.0;
    p.s.timestep[e] = 1; // environment.py:27
    p.s.reward_primed[e] = 0;
    for (int q = 0; q < p.R; ++q) {
        p.s.rock_cx[(size_t)e * p.R + q] = gen_u01(seed, e, GEN_ROCKS, 4 * q + 0) * (W * 0.75) + W * 0.25;
        p.s.rock_cy[(size_t)e * p.R + q] = gen_u01(seed, e, GEN_ROCKS, 4 * q + 1) * (H * 0.25) + H * 0.25;
        p.s.rock_r[(size_t)e * p.R + q] = gen_u01(seed, e, GEN_ROCKS, 4 * q + 2) * 5 + 5;
        p.s.rock_w[(size_t)e * p.R + q] = gen_u01(seed, e, GEN_ROCKS, 4 * q + 3) * 50 + 50;
    }
    for (int d = 0; d < g.n_food_discs; ++d) {
        int rad = (int)(gen_u01(seed, e, GEN_FOOD, 3 * d + 0) * (g.food_rmax - g.food_rmin) + g.food_rmin);
        const int cap = (m - 1) / 2;
        rad = rad > cap ? cap : rad;
        int32_t *dd = p.s.gen_discs + ((size_t)e * ANTSRL_MAX_FOOD_DISCS + d) * 3;
        dd[0] = rad;
        dd[1] = (int)(gen_u01(seed, e, GEN_FOOD, 3 * d + 1) * (W - 2 * rad) + rad);
        dd[2] = (int)(gen_u01(seed, e, GEN_FOOD, 3 * d + 2) * (H - 2 * rad) + rad);
    }
}

// per 32-cell word: anthill area (anthill.py:28-33), walls cleared on it (:66-67), food discs zeroed
// on walls (:71-72), empty pheromone and explored map
__global__ void k_gen_cells(const KP p, const AntsGen g, const uint64_t seed)
{
    const size_t n = (size_t)p.E * p.words;
    const size_t G = (size_t)p.W * p.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / p.words, w = i - e * p.words;
        const long ax = p.s.anthill_xyr[3 * e + 0], ay = p.s.anthill_xyr[3 * e + 1], ar = p.s.anthill_xyr[3 * e + 2];
        const int32_t *discs = p.s.gen_discs + e * ANTSRL_MAX_FOOD_DISCS * 3;
        uint32_t wb = 0, ab = 0;
        for (int b = 0; b < 32; ++b) {
            const size_t cell = w * 32 + b;
            if (cell >= G) break;
            const long x = (long)(cell / p.H), y = (long)(cell % p.H);
            const bool area = ar >= 0 && (ax - x) * (ax - x) + (ay - y) * (ay - y) <= ar * ar;
            const bool wall = !area && gen_u01(seed, (uint32_t)e, GEN_WALLS, (uint32_t)cell) < g.wall_density;
            bool fd = false;
            for (int d = 0; d < g.n_food_discs; ++d) {
                const long rad = discs[3 * d], dx = discs[3 * d + 1] - x, dy = discs[3 * d + 2] - y;
                fd |= dx * dx + dy * dy <= rad * rad;
            }
            if (area) ab |= 1u << b;
            if (wall) wb |= 1u << b;
            p.s.food[e * G + cell] = (fd && !wall) ? 1.0f : 0.0f;
            for (int c = 0; c < p.C; ++c) {
                p.s.phero[0][(e * G + cell) * p.C + c] = 0.0f;
                p.s.phero[1][(e * G + cell) * p.C + c] = 0.0f;
            }
        }
        p.s.walls_bits[i] = wb;
        p.s.area_bits[i] = ab;
        p.s.explored_bits[i] = 0u;
    }
}

// per ant: placement around the anthill (environment_generator.py:87-93), then Ants.__init__
__global__ void k_gen_ants(const KP p, const uint64_t seed)
{
    const size_t n = (size_t)p.E * p.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t e = (uint32_t)(i / p.N), a = (uint32_t)(i - (size_t)e * p.N);
        const double ax = (double)p.s.anthill_xyr[3 * e + 0], ay = (double)p.s.anthill_xyr[3 * e + 1];
        const double ar = (double)p.s.anthill_xyr[3 * e + 2];
        const double ang = gen_u01(seed, e, GEN_ANT_ANGLE, a) * 2 * PI_D;
        const double dist = gen_u01(seed, e, GEN_ANT_DIST, a) * ar * 0.8;
        const double x = warp_coord(cos(ang) * dist + ax, (double)p.W);
        const double y = warp_coord(sin(ang) * dist + ay, (double)p.H);
        p.s.x[i] = x; p.s.y[i] = y;
        p.s.theta[i] = gen_u01(seed, e, GEN_ANT_THETA, a) * 2 * PI_D;
        p.s.prev_x[i] = x; p.s.prev_y[i] = y;
        p.s.holding[i] = 0.0f; p.s.prev_holding[i] = 0.0f;
        p.s.mandibles[i] = 0; p.s.reward_state[i] = 0;
        p.s.seed[i] = (float)gen_u01(seed, e, GEN_ANT_SEED, a);
        p.s.dirty_cell[i] = -1;
        p.s.walldep_cell[i] = -1;
        for (int c = 0; c < p.C; ++c) p.s.activation[i * p.C + c] = 0.0f;
        const double dx = x - ax, dy = y - ay;
        p.s.prev_dist[i] = sqrt(dx * dx + dy * dy); // reward_custom.py:77
    }
}

__global__ void k_set_activation(const KP p, const float *__restrict__ act)
{
    const size_t n = (size_t)p.E * p.N * p.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p.s.activation[i] = act[i];
}

__global__ void k_read_state(const KP p, const int which, const int cur, void *__restrict__ dstv)
{
    const size_t EN = (size_t)p.E * p.N, G = (size_t)p.W * p.H, EG = (size_t)p.E * G;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    switch (which) {
    case ANTSRL_S_ANTS_XYT:
        for (size_t i = t0; i < EN; i += stride) {
            double *d = (double *)dstv + 3 * i;
            d[0] = p.s.x[i]; d[1] = p.s.y[i]; d[2] = p.s.theta[i];
        }
        break;
    case ANTSRL_S_PREV_XY:
        for (size_t i = t0; i < EN; i += stride) {
            ((double *)dstv)[2 * i] = p.s.prev_x[i];
            ((double *)dstv)[2 * i + 1] = p.s.prev_y[i];
        }
        break;
    case ANTSRL_S_HOLDING: for (size_t i = t0; i < EN; i += stride) ((float *)dstv)[i] = p.s.holding[i]; break;
    case ANTSRL_S_SEED: for (size_t i = t0; i < EN; i += stride) ((float *)dstv)[i] = p.s.seed[i]; break;
    case ANTSRL_S_MANDIBLES: for (size_t i = t0; i < EN; i += stride) ((uint8_t *)dstv)[i] = p.s.mandibles[i]; break;
    case ANTSRL_S_REWARD_STATE: for (size_t i = t0; i < EN; i += stride) ((uint8_t *)dstv)[i] = p.s.reward_state[i]; break;
    case ANTSRL_S_ACTIVATION:
        for (size_t i = t0; i < EN * p.C; i += stride) ((float *)dstv)[i] = p.s.activation[i];
        break;
    case ANTSRL_S_PHERO: // interleaved [E][G][C] -> canonical [E][C][G]
        for (size_t i = t0; i < EG * p.C; i += stride) {
            const size_t e = i / (G * p.C), rem = i - e * G * p.C, c = rem / G, g = rem - c * G;
            float v = p.s.phero[cur][(e * G + g) * p.C + c];
            if (p.scaled) {
                v *= (float)p.g_now;
                if (v < (float)p.threshold) v = 0.0f;
            }
            ((float *)dstv)[i] = v;
        }
        break;
    case ANTSRL_S_FOOD: for (size_t i = t0; i < EG; i += stride) ((float *)dstv)[i] = p.s.food[i]; break;
    case ANTSRL_S_EXPLORED:
    case ANTSRL_S_WALLS:
    case ANTSRL_S_ANTHILL_AREA: {
        const uint32_t *bits = which == ANTSRL_S_EXPLORED ? p.s.explored_bits
                               : which == ANTSRL_S_WALLS  ? p.s.walls_bits : p.s.area_bits;
        for (size_t i = t0; i < EG; i += stride) {
            const size_t e = i / G, g = i - e * G;
            ((uint8_t *)dstv)[i] = (uint8_t)test_bit(bits + e * p.words, (uint32_t)g);
        }
    } break;
    case ANTSRL_S_ANTHILL_FOOD: for (size_t i = t0; i < (size_t)p.E; i += stride) ((double *)dstv)[i] = p.s.anthill_food[i]; break;
    case ANTSRL_S_TIMESTEP: for (size_t i = t0; i < (size_t)p.E; i += stride) ((int32_t *)dstv)[i] = p.s.timestep[i]; break;
    case ANTSRL_S_ANTHILL_XYR: for (size_t i = t0; i < (size_t)p.E * 3; i += stride) ((int32_t *)dstv)[i] = p.s.anthill_xyr[i]; break;
    case ANTSRL_S_ROCK_RW:
        for (size_t i = t0; i < (size_t)p.E * p.R; i += stride) {
            ((double *)dstv)[2 * i] = p.s.rock_r[i];
            ((double *)dstv)[2 * i + 1] = p.s.rock_w[i];
        }
        break;
    case ANTSRL_S_ROCK_CENTERS:
        for (size_t i = t0; i < (size_t)p.E * p.R; i += stride) {
            ((double *)dstv)[2 * i] = p.s.rock_cx[i];
            ((double *)dstv)[2 * i + 1] = p.s.rock_cy[i];
        }
        break;
    default: break;
    }
}

// ===================================================================================
// host-side launchers (called from antsrl_capi.hip)
// ===================================================================================
static inline int pick_act_threads(int N) { (void)N; return 512; }

struct ActPlan {
    int threads;
    bool static_lds;
    size_t lds;
};

// LDS budget: 160 KiB per CU.  Preference order is MEASURED (c3, MI355X, profiles/plans.sh): the
// walls/anthill bitmaps in LDS with two workgroups per CU (0.285 ms) beat the plans that fetch those
// bits through L1/L2 (0.303 ms; a third workgroup per CU at 80 VGPRs does not raise the CU's
// throughput either) and one 1024-thread workgroup (0.306-0.315 ms).  So: bitmaps in
// LDS at 3 then 2 workgroups per CU, only then the global-bitmap plans, then one 1024-thread
// workgroup, then 512 threads with the whole CU's LDS.
// ANTSRL_ACT_PLAN=<n> pins candidate n (A/B runs).
static ActPlan plan_act(const KP &p)
{
    const size_t cap = 160 * 1024;
    const struct { int threads; bool st; size_t limit; } cand[] = {
        {512, true, cap / 3},  {512, true, cap / 2},  {512, false, cap / 3}, {512, false, cap / 2},
        {1024, true, cap},     {1024, false, cap},    {512, false, cap},
    };
    static const int pin = getenv("ANTSRL_ACT_PLAN") ? atoi(getenv("ANTSRL_ACT_PLAN")) : -1;
    ActPlan pl{};
    int k = 0;
    for (const auto &c : cand) {
        pl.threads = c.threads;
        pl.static_lds = c.st;
        pl.lds = act_lds_bytes(p.N, p.PP, p.words, p.HT, p.K, c.threads / 64, c.st, nullptr, nullptr, p.R);
        if (pin >= 0 ? k == pin : pl.lds <= c.limit) return pl;
        ++k;
    }
    return pl; // caller checks pl.lds <= cap
}

static size_t update_lds_bytes(const KP &p, int threads) { return update_scratch_bytes(p.HT, p.R, threads / 64); }

static int act_layout(const KP &p)
{
    static const int def[7] = {ANTSRL_CH_ANTS, ANTSRL_CH_PHERO, ANTSRL_CH_PHERO, ANTSRL_CH_ANTHILL,
                               ANTSRL_CH_WALLS, ANTSRL_CH_FOOD, ANTSRL_CH_ROCKS};
    if (p.C != 2 || (p.K != 6 && p.K != 7)) return LAYOUT_GENERIC;
    for (int k = 0; k < p.K; ++k)
        if (p.ch_kind[k] != def[k]) return LAYOUT_GENERIC;
    if (p.ch_arg[1] != 0 || p.ch_arg[2] != 1) return LAYOUT_GENERIC;
    return p.K == 6 ? LAYOUT_DEFAULT : LAYOUT_DEFAULT_ROCKS;
}

template <int C, bool ST, int LAYOUT, bool FAST, int TPB, bool OBS16 = false>
static hipError_t launch_act_t(const KP &p, const ActPlan &pl, const int8_t *rot, const int8_t *ph, int cur,
                               float *obs, float *agent_state, float *reward, uint8_t *done, int flags,
                               const double *jitter, int out_buf, hipStream_t st)
{
    static size_t attr_lds = 0; // dynamic-LDS opt-in is per kernel function, set once per size
    if (pl.lds > attr_lds) {
        hipError_t err = hipFuncSetAttribute((const void *)k_act<C, ST, LAYOUT, FAST, TPB, OBS16>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
        if (err != hipSuccess) return err;
        attr_lds = pl.lds;
    }
    hipLaunchKernelGGL((k_act<C, ST, LAYOUT, FAST, TPB, OBS16>), dim3(p.E), dim3(TPB), pl.lds, st, p, rot, ph, cur, obs,
                       agent_state, reward, done, flags, jitter, out_buf);
    return hipGetLastError();
}

template <int C, bool ST, int LAYOUT, bool FAST, bool OBS16 = false>
static hipError_t launch_act_k(const KP &p, const ActPlan &pl, const int8_t *rot, const int8_t *ph, int cur,
                               float *obs, float *agent_state, float *reward, uint8_t *done, int flags,
                               const double *jitter, int out_buf, hipStream_t st)
{
    if (pl.threads == 1024)
        return launch_act_t<C, ST, LAYOUT, FAST, 1024, OBS16>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags,
                                                       jitter, out_buf, st);
    return launch_act_t<C, ST, LAYOUT, FAST, 512, OBS16>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter,
                                                  out_buf, st);
}

template <int C>
static hipError_t launch_act_c(const KP &p, const int8_t *rot, const int8_t *ph, int cur, float *obs,
                               float *agent_state, float *reward, uint8_t *done, int flags,
                               const double *jitter, int out_buf, hipStream_t st)
{
    const ActPlan pl = plan_act(p);
    if (pl.lds > 160 * 1024) return hipErrorInvalidValue;
    const int layout = (C == 2) ? act_layout(p) : LAYOUT_GENERIC;
    const uint32_t row = (uint32_t)p.PP * p.K;
    // the pipelined loop: one pass (PP <= 64), row of 8..508 floats, observation wanted, no ablation
    const bool fast = C == 2 && layout != LAYOUT_GENERIC && p.PP <= 64 && row >= 8 && row <= 508 && obs &&
                      !(flags & 0x700); // (ACT_ABL_NO_EXPLORE is honoured by the pipelined loop too)
#define ACT_GO(ST, LY, FA) \
    return launch_act_k<C, ST, LY, FA>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st)
#define ACT_GO16(ST, LY) \
    return launch_act_k<C, ST, LY, true, true>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st)
    if (flags & ACT_OBS_BF16) { // bfloat16 observations: the pipelined loop on a default channel layout
        if (!fast || C != 2) return hipErrorNotSupported;
        if constexpr (C == 2) {
            if (layout == LAYOUT_DEFAULT) { if (pl.static_lds) ACT_GO16(true, LAYOUT_DEFAULT); else ACT_GO16(false, LAYOUT_DEFAULT); }
            else { if (pl.static_lds) ACT_GO16(true, LAYOUT_DEFAULT_ROCKS); else ACT_GO16(false, LAYOUT_DEFAULT_ROCKS); }
        }
    }
    if (C == 2 && layout != LAYOUT_GENERIC) {
        constexpr int LD = C == 2 ? LAYOUT_DEFAULT : LAYOUT_GENERIC, LR = C == 2 ? LAYOUT_DEFAULT_ROCKS : LAYOUT_GENERIC;
        constexpr bool F = C == 2;
        if (layout == LAYOUT_DEFAULT) {
            if (fast) { if (pl.static_lds) ACT_GO(true, LD, F); else ACT_GO(false, LD, F); }
            if (pl.static_lds) ACT_GO(true, LD, false); else ACT_GO(false, LD, false);
        } else {
            if (fast) { if (pl.static_lds) ACT_GO(true, LR, F); else ACT_GO(false, LR, F); }
            if (pl.static_lds) ACT_GO(true, LR, false); else ACT_GO(false, LR, false);
        }
    }
    if (pl.static_lds) ACT_GO(true, LAYOUT_GENERIC, false);
    ACT_GO(false, LAYOUT_GENERIC, false);
#undef ACT_GO
#undef ACT_GO16
}

hipError_t antsrl_launch_act(const KP &p, const int8_t *rot, const int8_t *ph, int cur, float *obs,
                             float *agent_state, float *reward, uint8_t *done, int flags,
                             const double *jitter, int out_buf, hipStream_t st)
{
    switch (p.C) {
    case 1: return launch_act_c<1>(p, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st);
    case 2: return launch_act_c<2>(p, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st);
    case 3: return launch_act_c<3>(p, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st);
    case 4: return launch_act_c<4>(p, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st);
    default: return hipErrorInvalidValue;
    }
}

bool antsrl_act_fits(const KP &p) { return plan_act(p).lds <= 160 * 1024; }

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
    } else if (!getenv("ANTSRL_SWEEP_TILED")) {
        const int fr = p.filter_radius;
        const int strips = (p.H + (64 - 2 * fr) - 1) / (64 - 2 * fr);
        // split the march along x into segments of >= 64 rows until the chip has ~16 waves per SIMD
        // to choose from (each extra segment re-reads 2R rows)
        int nseg = 1;
        while ((long long)p.E * strips * nseg < 16 * 1024 && p.W / (nseg * 2) >= 64) nseg *= 2;
        const int seg_rows = (p.W + nseg - 1) / nseg;
        dim3 grid((strips + 3) / 4, p.E, (p.W + seg_rows - 1) / seg_rows);
        if (p.filter_sep) {
            if (fr == 1) hipLaunchKernelGGL((k_sweep_march<C, 1, true>), grid, dim3(256), 0, st, p, in, out, seg_rows);
            else if (fr == 2) hipLaunchKernelGGL((k_sweep_march<C, 2, true>), grid, dim3(256), 0, st, p, in, out, seg_rows);
            else hipLaunchKernelGGL((k_sweep_march<C, 3, true>), grid, dim3(256), 0, st, p, in, out, seg_rows);
        } else if (fr == 1) hipLaunchKernelGGL((k_sweep_march<C, 1, false>), grid, dim3(256), 0, st, p, in, out, seg_rows);
        else if (fr == 2) hipLaunchKernelGGL((k_sweep_march<C, 2, false>), grid, dim3(256), 0, st, p, in, out, seg_rows);
        else hipLaunchKernelGGL((k_sweep_march<C, 3, false>), grid, dim3(256), 0, st, p, in, out, seg_rows);
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

hipError_t antsrl_launch_update(const KP &p, const double *jitter, int out_buf, hipStream_t st)
{
    static const bool force_loops = getenv("ANTSRL_UPDATE_LOOPS") != nullptr; // A/B: the per-phase loop kernel
    if (p.N <= 1024 && !force_loops) { // one ant per thread
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
    const int threads = pick_act_threads(p.N);
    const size_t lds = update_lds_bytes(p, threads);
    switch (p.C) {
    case 1: hipLaunchKernelGGL((k_update<1>), dim3(p.E), dim3(threads), lds, st, p, jitter, out_buf); break;
    case 2: hipLaunchKernelGGL((k_update<2>), dim3(p.E), dim3(threads), lds, st, p, jitter, out_buf); break;
    case 3: hipLaunchKernelGGL((k_update<3>), dim3(p.E), dim3(threads), lds, st, p, jitter, out_buf); break;
    case 4: hipLaunchKernelGGL((k_update<4>), dim3(p.E), dim3(threads), lds, st, p, jitter, out_buf); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t antsrl_launch_collect_full(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_collect_full, dim3(p.E), dim3(256), 0, st, p);
    return hipGetLastError();
}

static inline unsigned grid_for(size_t n);
hipError_t antsrl_launch_phero_wall_clear(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_phero_wall_clear, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t antsrl_launch_phero_renorm(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_phero_renorm, dim3(grid_for((size_t)p.E * p.W * p.H * p.C)), dim3(256), 0, st, p);
    return hipGetLastError();
}

static inline unsigned grid_for(size_t n)
{
    size_t b = (n + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

hipError_t antsrl_launch_reset(const KP &p, const AntsInit *in, hipStream_t st)
{
    hipLaunchKernelGGL(k_reset_env, dim3((p.E + 255) / 256), dim3(256), 0, st, p, in->anthill_xyr, in->rocks);
    hipLaunchKernelGGL(k_reset_ants, dim3(grid_for((size_t)p.E * p.N)), dim3(256), 0, st, p, in->ants_xyt, in->seed);
    hipLaunchKernelGGL(k_reset_bits, dim3(grid_for((size_t)p.E * p.words)), dim3(256), 0, st, p, in->walls,
                       in->anthill_xyr);
    hipLaunchKernelGGL(k_reset_grids, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p, in->food,
                       in->phero);
    return hipGetLastError();
}

hipError_t antsrl_launch_generate(const KP &p, const AntsGen &g, uint64_t seed, hipStream_t st)
{
    hipLaunchKernelGGL(k_gen_env, dim3((p.E + 255) / 256), dim3(256), 0, st, p, g, seed);
    hipLaunchKernelGGL(k_gen_cells, dim3(grid_for((size_t)p.E * p.words)), dim3(256), 0, st, p, g, seed);
    hipLaunchKernelGGL(k_gen_ants, dim3(grid_for((size_t)p.E * p.N)), dim3(256), 0, st, p, seed);
    return hipGetLastError();
}

hipError_t antsrl_launch_set_activation(const KP &p, const float *act, hipStream_t st)
{
    hipLaunchKernelGGL(k_set_activation, dim3(grid_for((size_t)p.E * p.N * p.C)), dim3(256), 0, st, p, act);
    return hipGetLastError();
}

hipError_t antsrl_launch_read_state(const KP &p, int which, int cur, void *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_read_state, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p, which, cur, dst);
    return hipGetLastError();
}
