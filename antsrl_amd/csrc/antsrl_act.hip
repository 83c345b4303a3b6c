// antsrl_act.hip — k_act: RLApi.step (RL_api.py:168-204) / RLApi.observation (RL_api.py:96-165),
// one workgroup per environment: mandibles + food exchange, activation, rotate, move, perception
// gather, reward.  Host launchers at the end.  No MFMA: nothing on this path is a dense contraction.
#include "antsrl_util.h"
#include "antsrl_update_env.h"
#include "antsrl_flush.h"

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
                                                       bool static_lds, int R, bool big = false)
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
    o.pres = take(big ? 0 : 4 * (size_t)words); // big: both maps live in HBM (DState::big_pres / big_old)
    o.old = take(big ? 0 : 4 * (size_t)words);
    o.walls = o.area = 0;
    if (static_lds) {
        o.walls = take(4 * (size_t)words);
        o.area = take(4 * (size_t)words);
    }
    o.mask = take((size_t)PP);
    o.rock = take(32 * (size_t)(R > 0 ? R : 1));
    // staging: one row + up to 3 floats of misalignment; TWO rows for the shapes the pipelined loop takes
    // (it stages a group's rows back to back)
    const size_t rowf = (size_t)PP * K;
    const size_t stride = ((PP <= 64 && rowf <= 368 ? 2 * rowf : rowf) + 3 + 3) / 4 * 4;
    const size_t hash_b = update_scratch_bytes(HT, R, nwaves), stage_b = 4 * (size_t)nwaves * stride;
    o.uni = take(hash_b > stage_b ? hash_b : stage_b);
    o.stride = (uint32_t)stride;
    o.total = off;
    return o;
}

__host__ __device__ __forceinline__ size_t act_lds_bytes(int N, int PP, int words, int HT, int K, int nwaves,
                                                         bool static_lds, void *, unsigned char *, int R = 0,
                                                         bool big = false)
{
    return act_offsets(N, PP, words, HT, K, nwaves, static_lds, R, big).total;
}

// Perception-channel layouts known at compile time (straight-line output code); anything else
// takes the generic per-channel selection.
#define LAYOUT_GENERIC 0
#define LAYOUT_DEFAULT 1       // [Ants, Phero0, Phero1, Anthill, Walls, Food]   (generator order)
#define LAYOUT_DEFAULT_ROCKS 2 // ... + [CircleObstacles]

// Profiling build only (-DANTSRL_PROFILING, ANTSRL_ABLATE bit ACT_ABL_TRACE): per-workgroup phase timeline of
// k_act.  Slot k of workgroup e = s_memrealtime (100 MHz) at: 0 entry, 1 after phase 2, 2 after phase 3,
// 3 exit; slot 4 = HW_ID, slot 5 = XCC_ID, slot 6 after phase 0, slot 7 after phase 1.  Read back with
// antsrl_debug_read_act_trace (exported by libantsrl_hip_prof.so only).
#ifdef ANTSRL_PROFILING
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
#else
__device__ __forceinline__ void act_trace(int, int, int, int) {}
#endif

// FAST selects the software-pipelined perception loop (see phase 3); the two loops live in
// separate instantiations on purpose: with both in one kernel the optimiser stops scalarising the
// LDS carve (`ActLds`), its pointers go through scratch and every LDS access degrades to flat_*.
// TPB = threads per workgroup (512 or 1024); 4 waves per SIMD (<= 128 VGPRs) is all the LDS plans
// can use (capping at 80 VGPRs for a third workgroup per CU measured slower, see plan_act).
// BIG: a grid whose presence / explored bit maps do not fit the workgroup's LDS keeps both in HBM scratch
// (DState::big_pres / big_old, generic loop only): same code, the two pointers change address space.
template <int C, bool STATIC_LDS, int LAYOUT, bool FAST, int TPB, bool OBS16 = false, bool ILV = false, bool BIG = false>
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
    const ActOff lo = act_offsets(N, PP, p.words, p.HT, K, nwaves, STATIC_LDS, R, BIG);
    ActLds L; // filled field by field right here: never has its address taken, stays in registers
    L.frame = (AntFrame *)(smem + lo.frame);
    L.off = (CellOff *)(smem + lo.off);
    L.cnt = (uint32_t *)(smem + lo.cnt);
    L.rockmask = (uint32_t *)(smem + lo.rm);
    if (BIG) {
        L.b_pres = p.s.big_pres + (size_t)blockIdx.x * p.words;
        L.b_old = p.s.big_old + (size_t)blockIdx.x * p.words;
    } else {
        L.b_pres = (uint32_t *)(smem + lo.pres);
        L.b_old = (uint32_t *)(smem + lo.old);
    }
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
    const FoodView food{p.s.food + (size_t)e * G * p.fs, p.fs};
    const size_t PS = (size_t)p.ps; // floats per cell of the pheromone array (4 = interleaved record, see DState)
    const float *ph = p.s.phero[cur] + (size_t)e * G * PS;
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
#ifdef ANTSRL_PROFILING
    const bool abl_gather = flags & ACT_ABL_NO_GATHER, abl_store = flags & ACT_ABL_NO_STORE;
    const bool abl_explore = flags & ACT_ABL_NO_EXPLORE, abl_items = flags & ACT_ABL_NO_ITEMS;
#else
    constexpr bool abl_gather = false, abl_store = false, abl_explore = false, abl_items = false;
#endif
    const int npass = (PP + 63) >> 6;
    const uint32_t row = (uint32_t)PP * (uint32_t)K;            // floats per ant
    float *stage = L.stage + (size_t)wave * L.stage_stride;
    const bool wrap_fast = W > 4 * (p.r + 4) && H > 4 * (p.r + 4) && p.fwd_delta < W / 4 && p.fwd_delta < H / 4 &&
                           p.fwd_delta > -W / 4 && p.fwd_delta > -H / 4 && p.delta < 2.0; // one conditional add wraps
    const bool wrap_pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0;
    // Fast path (single pass: PP <= 64, row <= 368 floats — the reference's 7x7 with its 6 or 7
    // channels): software-pipelined by one group of ACT_UNROLL ants.  Everything that touches
    // global memory is STRAIGHT-LINE and unconditional (out-of-range ants/lanes are clamped onto
    // valid ones and redo identical work: benign duplicate stores), so the compiler can count
    // outstanding operations: the wait for group g's gathers is a `vmcnt(n)` that leaves group
    // g+1's gathers AND group g-1's observation stores in flight.  (vmcnt retires in order and
    // counts stores: an uncounted wait would make every gather wait for the previous stores.)
    if (FAST) { // host guarantees: npass == 1, 8 <= row <= 368, obs != nullptr
        const int q = lane < PP ? lane : PP - 1;            // lanes beyond the perception clamp onto its last cell
        const CellOff of = L.off[q];
        const bool mask_q = L.t_mask[q] != 0;
        const uint32_t qK = (uint32_t)q * K;
        // Lanes whose gathered value is never used (masked cells and the lanes past the perception that
        // duplicate a masked last cell: 27 of 64 with the reference's mask) gather what the first needed lane
        // gathers.  The CU's texture-address path
        // works through a scattered gather at about one distinct cache line per clock — it, not HBM, paces
        // the perception loop (49 + 86 clocks per ant for the gather and the row's stores) — and lanes on
        // a line that is fetched anyway cost nothing (measured: k_act -4 %; within the quad only: -3.4 %).
        // (Lanes past the perception are duplicates of its last cell, q = PP - 1: they stage the same values
        // into the same slots, so they keep that cell's address whenever it is a needed one.)
        const unsigned long long need_mask = __ballot(mask_q);
        const int src_lane = mask_q ? lane : (need_mask ? __builtin_ctzll(need_mask) : 0);
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
            const uint32_t gc_ = (uint32_t)__shfl((int)CELL[u], src_lane); /* (CELL keeps the lane's own cell) */ \
            if (ILV) { /* one {p0, p1, food, pad} record per cell: a single 16-byte gather */            \
                const float4 t = *reinterpret_cast<const float4 *>(ph + (size_t)gc_ * 4);                \
                PVV[u][0] = t.x; PVV[u][C - 1] = t.y; FDV[u] = t.z;                                      \
            } else {                                                                                     \
                if (C == 2) {                                                                            \
                    const float2 t = *reinterpret_cast<const float2 *>(ph + (size_t)gc_ * 2);            \
                    PVV[u][0] = t.x; PVV[u][C - 1] = t.y;                                                \
                } else {                                                                                 \
                    _Pragma("unroll") for (int c = 0; c < C; ++c) PVV[u][c] = ph[(size_t)gc_ * C + c];   \
                }                                                                                        \
                FDV[u] = food[gc_];                                                                      \
            }                                                                                            \
        }                                                                                                \
    }
        uint32_t c_cell[ACT_UNROLL], n_cell[ACT_UNROLL];
        int c_ix[ACT_UNROLL], c_iy[ACT_UNROLL], n_ix[ACT_UNROLL], n_iy[ACT_UNROLL];
        float c_pv[ACT_UNROLL][C], n_pv[ACT_UNROLL][C], c_fd[ACT_UNROLL], n_fd[ACT_UNROLL];
        // Each wave owns a CONTIGUOUS run of ants, so its observation rows form one sequential write
        // stream: the partial cache line at the end of a row is completed by the same wave's next row
        // while it is still in L2 (profiles/history/obs_write_probe.hip: 4.2 -> 4.9 TB/s for this pattern
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
                // float32: the group's rows are staged back to back, as they lie in memory (image shifted by
                // the first row's 16-byte misalignment), and flushed together after the loop
                const uint32_t mis0 = (uint32_t)(((uintptr_t)(obs_env + (size_t)i0 * row) >> 2) & 3);
                float *o = stage + mis0 + (uint32_t)u * row + qK;
                // bfloat16 observations: the same staging and copy-out on 2-byte elements (8 per 16 bytes)
                uint16_t *dst16_0 = reinterpret_cast<uint16_t *>(obs) + ((size_t)e * N + (size_t)i0) * row;
                const uint32_t mis16_0 = (uint32_t)(((uintptr_t)dst16_0 >> 1) & 7);
                uint16_t *o16 = reinterpret_cast<uint16_t *>(stage) + mis16_0 + (uint32_t)u * row + qK;
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
            }
            if (OBS16) {
                // the group's bfloat16 rows as one run: two 16-byte stores per lane over the whole 8-element
                // pieces (the first 64 start on a 128-byte line), one 2-byte store for the <= 14 edge elements
                wave_lds_sync();
                const uint32_t rowp = (i0 + 1 < i_end) ? 2u * row : row;
                uint16_t *dst16 = reinterpret_cast<uint16_t *>(obs) + ((size_t)e * N + (size_t)i0) * row;
                const uint32_t mis16 = (uint32_t)(((uintptr_t)dst16 >> 1) & 7);
                uint16_t *st16 = reinterpret_cast<uint16_t *>(stage), *d_al = dst16 - mis16;
                const FlushPlanB16 f = flush_plan_b16((uint32_t)lane, mis16, rowp, (uint32_t)((uintptr_t)d_al >> 4) & 7u);
                const uint4 w1 = reinterpret_cast<const uint4 *>(st16)[f.g1];
                const uint4 w2 = reinterpret_cast<const uint4 *>(st16)[f.g2];
                const uint16_t we = st16[f.fe];
                store_stream(reinterpret_cast<uint4 *>(d_al) + f.g1, w1);
                store_stream(reinterpret_cast<uint4 *>(d_al) + f.g2, w2);
                store_stream(d_al + f.fe, we);
                wave_lds_sync();
            }
            if (!OBS16) {
                // copy the group out: its rows are contiguous in memory, so the (up to) two rows leave as ONE
                // run: two 16-byte stores per lane over 128 float4s that start on a 128-byte line (16 whole
                // lines: streaming stores of whole lines are cheaper than pieces,
                // profiles/history/store_policy_probe.hip), a third over the pieces in front of and behind them,
                // one 4-byte store for the <= 6 edge floats; lanes with nothing left repeat a valid store
                wave_lds_sync();
                const uint32_t rowp = (i0 + 1 < i_end) ? 2u * row : row; // (odd tail of the run: one row)
                float *dst = obs_env + (size_t)i0 * row;
                const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
                float *dst_al = dst - mis;
                const FlushPlanF32 f = flush_plan_f32((uint32_t)lane, mis, rowp, (uint32_t)((uintptr_t)dst_al >> 4) & 7u);
                const float4 v1 = reinterpret_cast<const float4 *>(stage)[f.j1];
                const float4 v2 = reinterpret_cast<const float4 *>(stage)[f.j2];
                const float4 v3 = reinterpret_cast<const float4 *>(stage)[f.j3];
                const float ve = stage[f.fe];
                store_stream(reinterpret_cast<float4 *>(dst_al) + f.j1, v1);
                store_stream(reinterpret_cast<float4 *>(dst_al) + f.j2, v2);
                store_stream(reinterpret_cast<float4 *>(dst_al) + f.j3, v3);
                store_stream(dst_al + f.fe, ve);
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
    for (int i0 = wave * ustep; i0 < (abl_items ? 0 : N); i0 += nwaves * ustep) {
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
                        const float2 t = *reinterpret_cast<const float2 *>(ph + (size_t)cell[u] * PS);
                        pv[u][0] = t.x; pv[u][C - 1] = t.y;
                    } else {
#pragma unroll
                        for (int c = 0; c < C; ++c) pv[u][c] = ph[(size_t)cell[u] * PS + c];
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
                    // (BIG: the map was built by L2 atomics, read it past the CU's L1)
                    const uint32_t presw = BIG ? __hip_atomic_load(&L.b_pres[wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : L.b_pres[wd];
                    const float v_ants = (presw & bit) ? 1.0f : 0.0f;  // :142
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
// host-side launchers (called from antsrl_capi.hip)
// ===================================================================================
struct ActPlan {
    int threads;
    bool static_lds;
    size_t lds;
    bool big; // presence / explored maps in HBM scratch (grids past ~600k cells)
};

// LDS budget: 160 KiB per CU.  Preference order is MEASURED (c3, MI355X, profiles/history/plans.sh): the
// walls/anthill bitmaps in LDS with two workgroups per CU (0.285 ms) beat the plans that fetch those
// bits through L1/L2 (0.303 ms; a third workgroup per CU at 80 VGPRs does not raise the CU's
// throughput either) and one 1024-thread workgroup (0.306-0.315 ms).  So: bitmaps in
// LDS at 3 then 2 workgroups per CU, only then the global-bitmap plans, then one 1024-thread
// workgroup, then 512 threads with the whole CU's LDS.
// ANTSRL_ACT_PLAN=<n> pins candidate n (A/B runs, profiling build only).
static ActPlan plan_act(const KP &p)
{
    const size_t cap = 160 * 1024;
    const struct { int threads; bool st; size_t limit; } cand[] = {
        {512, true, cap / 3},  {512, true, cap / 2},  {512, false, cap / 3}, {512, false, cap / 2},
        {1024, true, cap},     {1024, false, cap},    {512, false, cap},
    };
    static const int pin = PROF_ENV("ANTSRL_ACT_PLAN") ? atoi(PROF_ENV("ANTSRL_ACT_PLAN")) : -1;
    ActPlan pl{};
    // A batch that leaves half the CUs without a workgroup (E <= CUs / 2) with at least 512 ants per env:
    // one 1024-thread workgroup per env puts twice the waves on the env's perception (c3's envs at
    // E = 128: k_act 0.052 -> 0.041 ms; at E = 256 it is 6 % slower, profiles/history/plan_small_e.sh).
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n;
    }();
    if (pin < 0 && (long)p.E * 2 <= n_cus && p.N >= 512) {
        for (bool st : {true, false}) {
            pl.threads = 1024; pl.static_lds = st;
            pl.lds = act_lds_bytes(p.N, p.PP, p.words, p.HT, p.K, 16, st, nullptr, nullptr, p.R);
            if (pl.lds <= cap) return pl;
        }
    }
    int k = 0;
    for (const auto &c : cand) {
        pl.threads = c.threads;
        pl.static_lds = c.st;
        pl.lds = act_lds_bytes(p.N, p.PP, p.words, p.HT, p.K, c.threads / 64, c.st, nullptr, nullptr, p.R);
        if (pin >= 0 ? k == pin : pl.lds <= c.limit) return pl;
        ++k;
    }
    if (pin < 0) { // nothing fits with the grid's bit maps in LDS: keep them in HBM
        for (size_t limit : {cap / 2, cap}) {
            pl.threads = 512; pl.static_lds = false; pl.big = true;
            pl.lds = act_lds_bytes(p.N, p.PP, p.words, p.HT, p.K, 8, false, nullptr, nullptr, p.R, true);
            if (pl.lds <= limit) return pl;
        }
    }
    return pl; // caller checks pl.lds <= cap
}

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

template <int C, bool ST, int LAYOUT, bool FAST, int TPB, bool OBS16 = false, bool ILV = false, bool BIG = false>
static hipError_t launch_act_t(const KP &p, const ActPlan &pl, const int8_t *rot, const int8_t *ph, int cur,
                               float *obs, float *agent_state, float *reward, uint8_t *done, int flags,
                               const double *jitter, int out_buf, hipStream_t st)
{
    // dynamic-LDS opt-in is per kernel function AND per device: set once per size on each device
    static size_t attr_lds[ANTSRL_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice;
    if (pl.lds > attr_lds[dev]) {
        hipError_t err = hipFuncSetAttribute((const void *)k_act<C, ST, LAYOUT, FAST, TPB, OBS16, ILV, BIG>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
        if (err != hipSuccess) return err;
        attr_lds[dev] = pl.lds;
    }
    hipLaunchKernelGGL((k_act<C, ST, LAYOUT, FAST, TPB, OBS16, ILV, BIG>), dim3(p.E), dim3(TPB), pl.lds, st, p, rot, ph, cur, obs,
                       agent_state, reward, done, flags, jitter, out_buf);
    return hipGetLastError();
}

template <int C, bool ST, int LAYOUT, bool FAST, bool OBS16 = false, bool ILV = false>
static hipError_t launch_act_k(const KP &p, const ActPlan &pl, const int8_t *rot, const int8_t *ph, int cur,
                               float *obs, float *agent_state, float *reward, uint8_t *done, int flags,
                               const double *jitter, int out_buf, hipStream_t st)
{
    if (pl.threads == 1024)
        return launch_act_t<C, ST, LAYOUT, FAST, 1024, OBS16, ILV>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags,
                                                       jitter, out_buf, st);
    return launch_act_t<C, ST, LAYOUT, FAST, 512, OBS16, ILV>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter,
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
    // the pipelined loop: one pass (PP <= 64), row of 8..368 floats (a group of two rows leaves in three
    // 16-byte stores per lane), observation wanted, no ablation
    const bool fast = C == 2 && layout != LAYOUT_GENERIC && p.PP <= 64 && row >= 8 && row <= 368 && obs &&
                      !(flags & 0x700) && !pl.big; // (ACT_ABL_NO_EXPLORE is honoured by the pipelined loop too)
#define ACT_GO(ST, LY, FA) \
    return launch_act_k<C, ST, LY, FA>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st)
#define ACT_GOF(ST, LY, O16, IL) \
    return launch_act_k<C, ST, LY, true, O16, IL>(p, pl, rot, ph, cur, obs, agent_state, reward, done, flags, jitter, out_buf, st)
    // the pipelined loop is specialised on the observation format and on the cell record (DState::phero)
#define ACT_FAST(ST, LY)                                                                          \
    {                                                                                             \
        if (o16) { if (ilv) ACT_GOF(ST, LY, true, true); else ACT_GOF(ST, LY, true, false); }     \
        else { if (ilv) ACT_GOF(ST, LY, false, true); else ACT_GOF(ST, LY, false, false); }       \
    }
    const bool o16 = (flags & ACT_OBS_BF16) != 0, ilv = p.ps == 4 && p.fs == 4;
    if (o16 && (!fast || C != 2 || row < 16)) return hipErrorNotSupported; // bfloat16 observations: pipelined loop only
    if (pl.big) // generic loop, generic channel selection, 512 threads
        return launch_act_t<C, false, LAYOUT_GENERIC, false, 512, false, false, true>(p, pl, rot, ph, cur, obs, agent_state, reward,
                                                                                      done, flags, jitter, out_buf, st);
    if constexpr (C == 2) {
        if (layout != LAYOUT_GENERIC) {
            if (layout == LAYOUT_DEFAULT) {
                if (fast) { if (pl.static_lds) ACT_FAST(true, LAYOUT_DEFAULT) else ACT_FAST(false, LAYOUT_DEFAULT) }
                if (pl.static_lds) ACT_GO(true, LAYOUT_DEFAULT, false); else ACT_GO(false, LAYOUT_DEFAULT, false);
            } else {
                if (fast) { if (pl.static_lds) ACT_FAST(true, LAYOUT_DEFAULT_ROCKS) else ACT_FAST(false, LAYOUT_DEFAULT_ROCKS) }
                if (pl.static_lds) ACT_GO(true, LAYOUT_DEFAULT_ROCKS, false); else ACT_GO(false, LAYOUT_DEFAULT_ROCKS, false);
            }
        }
    }
    if (pl.static_lds) ACT_GO(true, LAYOUT_GENERIC, false);
    ACT_GO(false, LAYOUT_GENERIC, false);
#undef ACT_GO
#undef ACT_GOF
#undef ACT_FAST
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
bool antsrl_act_needs_hbm_maps(const KP &p) { return plan_act(p).big; }
