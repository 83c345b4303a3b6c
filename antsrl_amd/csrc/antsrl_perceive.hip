// antsrl_perceive.hip — RLApi.step (RL_api.py:168-204) / RLApi.observation (RL_api.py:96-165) on the
// cell-meta layout (antsrl_device.h), as two kernels:
//
//   k_move      one workgroup per environment, one ant per thread: mandible decision + food exchange
//               (RL_api.py:178-185, ants.py:102-117, LDS last-writer-wins), activation (ants.py:89-96),
//               rotate, forward move (RL_api.py:190-196) and the presence stamp of the ant's cell
//               (RL_api.py:137-142).  (The steady loop runs it fused behind the previous step's update: k_update_move.)
//   k_perceive  the perception gather (RL_api.py:109-153) + rewards (rewards/reward_custom.py) + agent_state
//               (RL_api.py:160-162).  ONE WAVE PER RUN OF ANTS, no per-environment state in LDS and no
//               barrier after the prologue: each perceived cell is ONE gather that returns pheromone,
//               food, wall / anthill / presence bits and the explored stamp (META word), so the work
//               splits into (environment, segment) workgroups of any size — the chip never waits for an
//               environment's per-ant phases and the tail is one short run per wave, not a 512-ant
//               workgroup (round 1's k_act: 2 rounds of 512 resident workgroups, the first 20 us and
//               the last 60 us of a 206 us launch half idle).
//
// No MFMA: nothing on this path is a dense contraction.  Host launchers at the end.
#include "antsrl_util.h"
#include "antsrl_flush.h"
#include "antsrl_update_one.h"

#define PRC_UNROLL 2 // ants per group (their gathers are issued together, their rows leave together)
// The gathers of TWO groups are in flight ahead of the group being consumed (same-box A/B of one / two / all four groups
// ahead, k_perceive ms: c3 0.2274 / 0.2211 / 0.2284, c4 0.518 / 0.484 / 0.480, c5 0.0760 / 0.0749 / 0.0805 —
// profiles/r03/depth_ab.txt; the other two forms are in profiles/r04/perceive_cleanup.patch).

// Compile-time ablations for profiles/ (results WRONG by design): bit masks, variant builds of the PROFILING library only
// (`python -m antsrl_amd.build --variant NAME -DPRC_ABL=3`); antsrl_device.h refuses them in a product build.
//   PRC_ABL  1: no record gathers (every lane reads one L1-resident line)   2: no observation stores   4: no explored marks
//   UM_ABL   1: no food read   2: no presence stamp   4: no deposit read-modify-write      (k_update_move / k_move)
constexpr bool abl_gather = (PRC_ABL & 1) != 0, abl_store = (PRC_ABL & 2) != 0, abl_mark = (PRC_ABL & 4) != 0;

#define PLAYOUT_DEFAULT 1       // [Ants, Phero0, Phero1, Anthill, Walls, Food]   (generator order)
#define PLAYOUT_DEFAULT_ROCKS 2 // ... + [CircleObstacles]

// ---------------------------------------------------------------------------------------------------
// k_move
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t move_lds_bytes(int HT, int N)
{
    return align_up(8 * (size_t)HT + 12 * (size_t)N, 16);
}

template <int C>
__device__ __forceinline__ void move_body(const KP &p, const int e, const int8_t *__restrict__ rotation,
                                          const int8_t *__restrict__ phero_act, uint8_t *__restrict__ done, const int do_step,
                                          const uint32_t seq, unsigned char *smem, const UmFwd *fw = nullptr)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int N = p.N, W = p.W, H = p.H;
    const size_t G = (size_t)W * H, eN = (size_t)e * N;
    uint32_t *hkeys = (uint32_t *)smem, *hvals = hkeys + p.HT;
    float *tmp_q = (float *)(hvals + p.HT), *tmp_d = tmp_q + N;
    uint32_t *cprevs = (uint32_t *)(tmp_d + N);

    const uint32_t *area = p.s.area_bits + (size_t)e * p.words;
    const FoodView food{p.s.food + (size_t)e * G * p.fs, p.fs};
    // presence stamp of cell g (lower half-word of its META word): pres[g * 2 * fs]
    uint16_t *pres = reinterpret_cast<uint16_t *>(p.s.food + (size_t)e * G * p.fs) + 2;
    const size_t FS2 = 2 * (size_t)p.fs;

    // One ant per thread (N <= T, the reference's sizes): every independent global load of the ant is issued
    // here, ahead of the first barrier; the phases below use the registers.
    const bool one = fw ? true : N <= T; // wave-uniform (k_update_move: one ant per thread by construction, known at compile time)
    const size_t a1 = eN + (tid < N ? tid : 0);
    double h_x = 0.0, h_y = 0.0, h_th = 0.0;
    float h_hold = 0.0f, h_q = 0.0f;
    int h_m = 0, h_rot = 0, h_pa = 0;
    uint32_t h_cprev = 0u;
    if (one) {
        // (k_update_move: the update of the same workgroup has just written x / y / theta and prev := (x, y) of this very ant:
        // handed over in registers — the same values, one memory round trip less in front of the dependent food read)
        if (fw) { // (k_update_move: no global load at all in front of the move's first barrier)
            h_x = fw->x; h_y = fw->y; h_th = fw->th;
            h_hold = fw->hold; h_m = fw->m & 1; h_rot = fw->rot; h_pa = fw->pa;
            h_cprev = frec_xy(p, (int)fw->x, (int)fw->y);
            // The record was loaded by the update before its anthill collect ran.  Food on the anthill area is all zero once
            // the collect is done (anthill.py:41-46; the handle only defers an update that needs no full-grid collect), and
            // nothing else in the update writes food: an area cell reads 0, any other cell what the load returned.
            if (fw->rec) h_q = (UM_ABL & 1) ? 0.0f : ((fw->meta & META_AREA) ? 0.0f : fw->food);
            else h_q = (UM_ABL & 1) ? 0.0f : food[h_cprev];
        } else {
            h_x = p.s.x[a1]; h_y = p.s.y[a1]; h_th = p.s.theta[a1];
            h_hold = p.s.holding[a1];
            if (do_step) {
                const double ppx = STP_LD(p.s.prev_x[a1]), ppy = STP_LD(p.s.prev_y[a1]);
                h_m = STP_LD(p.s.mandibles[a1]);
                if (rotation) h_rot = STP_LD(rotation[a1]);
                if (phero_act) h_pa = STP_LD(phero_act[a1]);
                h_cprev = frec_xy(p, (int)ppx, (int)ppy); // the food / META RECORD of the previous cell (hash key, food, dirty list)
                h_q = (UM_ABL & 1) ? 0.0f : food[h_cprev]; // food is first written in phase 1b
            }
        }
    }
    // (k_update_move: the update of the same launch has resolved last-writer-wins over exactly these cells — prev := cur, so
    //  the food cell of the exchange is the deposit cell — and hands every ant its verdict, UmFwd::win: no table, no
    //  inserts, and none of the move's barriers, which exist for the table alone; profiles/r04/um_trace_final.txt slots 10-12)
    if (do_step && !fw)
        for (int h = tid; h < p.HT; h += T) {
            hkeys[h] = HASH_EMPTY;
            hvals[h] = 0u;
        }
    if (tid == 0) {
        // (k_update_move: both values arrive in registers — no load, hence no memory round trip of thread 0's wave in
        //  front of the barrier the whole workgroup waits at)
        if (p.reward_kind != ANTSRL_REWARD_NONE) {
            // the first observation after Reward.setup sees delta-holding == 0 (alias quirk,
            // reward_custom.py:35,68): k_perceive reads the flag as it stood before this observation
            p.s.primed_cur[e] = fw ? fw->primed : p.s.reward_primed[e];
            p.s.reward_primed[e] = 1;
        }
        if (do_step && done) done[e] = (uint8_t)(p.max_time == (fw ? fw->ts : p.s.timestep[e])); // RL_api.py:200
    }
    if (!fw) __syncthreads();
    else __builtin_amdgcn_sched_barrier(0); // (no barrier, but nothing of the phases below is hoisted above this point either:
                                            //  the explicit-sweep variant of k_update_move spills three more VGPRs otherwise)
    if (fw) UM_STAMP(10);

    if (do_step) {
        // ---- phase 1a: mandible target (RL_api.py:178-185) + Ants.update_mandibles reads
        //      (ants.py:102-114).  All food reads happen before any food write.
        for (int i = tid; i < N; i += T) {
            double x, y;
            uint32_t cprev;
            float q, hold;
            int old_m;
            if (one) {
                x = h_x; y = h_y; cprev = h_cprev; q = h_q; old_m = h_m; hold = h_hold;
            } else {
                x = p.s.x[eN + i]; y = p.s.y[eN + i];
                cprev = frec_xy(p, (int)p.s.prev_x[eN + i], (int)p.s.prev_y[eN + i]);
                q = food[cprev];
                old_m = p.s.mandibles[eN + i];
                hold = p.s.holding[eN + i];
            }
            const uint32_t ccur = (uint32_t)((int)x * H + (int)y);
            int m = old_m;
            { // perceived_objects order matters (:178-185): the walk's last two distinct operations, KP::mand_first / mand_last
                const int set = q > 0.0f; // Food, :182
                // Anthill, :184 (k_update_move: the ant still stands where the update left it — the forwarded record's own area bit)
                const int keep = 1 - (int)((fw && fw->rec) ? (fw->meta & META_AREA) != 0 : test_bit(area, ccur));
                if (p.mand_first == 1) m |= set;
                else if (p.mand_first == 2) m &= keep;
                if (p.mand_last == 1) m |= set;
                else if (p.mand_last == 2) m &= keep;
            }
            const int closing = m & (1 - old_m), opening = (1 - m) & old_m; // ants.py:103-104
            const float taken = fminf((float)p.max_hold, fmaxf(0.0f, q)) * (float)closing; // :111
            const float dropped = hold * (float)opening;                                    // :114
            h_hold = hold + (taken - dropped);                                              // :117
            p.s.holding[eN + i] = h_hold;
            STP_ST(p.s.mandibles[eN + i], (uint8_t)m);                                            // :107
            cprevs[i] = cprev;
            tmp_q[i] = q;
            tmp_d[i] = dropped - taken;
            if (fw) break; // (one ant per thread: a single pass, known at compile time; its slots of the scratch arrays are its own)
            lww_insert(hkeys, hvals, (uint32_t)p.HT - 1, cprev, (uint32_t)i);
        }
        if (!fw) __syncthreads();
        else __builtin_amdgcn_sched_barrier(0);
        if (fw) UM_STAMP(11);
        // ---- phase 1b: ants.py:116 `qte[cell] += dropped - taken`, last ant on a cell wins
        for (int i = tid; i < N; i += T) {
            const uint32_t cprev = cprevs[i];
            const float delta = tmp_d[i];
            int32_t dirty = -1;
            if (delta != 0.0f && (fw ? (fw->m & UMFWD_WIN) != 0 : lww_winner(hkeys, hvals, (uint32_t)p.HT - 1, cprev) == (uint32_t)i)) {
                food[cprev] = tmp_q[i] + delta;
                // (on the anthill area?  the record's own META word says so: cprev is a record index, not a cell id)
                if (__float_as_uint((&food[cprev])[1]) & META_AREA) dirty = (int32_t)cprev;
            }
            STP_ST(p.s.dirty_cell[eN + i], dirty);
            if (fw) break;
        }
        if (fw) UM_STAMP(12);
    }

    // ---- phase 2: activation, rotate, move (RL_api.py:187-196), presence stamp
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
                STP_ST(p.s.activation[(eN + i) * C + 0], a0);
                if (C > 1) STP_ST(p.s.activation[(eN + i) * C + 1], a1);
            }
            double sn, cs;
            if (fw && fw->pre) { // (k_update_move: evaluated by the update under its record load — same operations, same inputs)
                th = fw->th_new; sn = fw->sn; cs = fw->cs;
            } else {
                if (rotation) // Ants.rotate_ants + warp_theta, ants.py:62-67
                    th = np_mod_d(th + (double)(one ? h_rot : (int)rotation[eN + i]) * p.max_rot_speed, 2 * PI_D);
                sincos(th, &sn, &cs);
            }
            // RL_api.py:194-196, Ants.forward_ants ants.py:77-80
            double fwd = 1.0 * p.max_speed * (1 - (double)(one ? h_hold : p.s.holding[eN + i]) * p.carry);
            if (fwd < 0) fwd *= p.backward;
            x = warp_coord(x + cs * fwd, (double)W);
            y = warp_coord(y + sn * fwd, (double)H);
            p.s.x[eN + i] = x;
            p.s.y[eN + i] = y;
            p.s.theta[eN + i] = th;
        }
        // presence map, RL_api.py:137-141 (0/1, not a count): this observation's number into the cell's stamp
        const uint32_t cell = frec_xy(p, wrap_index((int)x, W), wrap_index((int)y, H));
        // (a plain store: as an nt store k_update_move gains 1 us and k_perceive, whose gathers then miss the line, loses 4:
        //  profiles/r03/ntstamp_ab.txt)
        if (!(UM_ABL & 2) || cell == 0xFFFFFFFFu) pres[(size_t)cell * FS2] = (uint16_t)seq;
        if (fw && fw->frm_off) {
            // the ant's perception frame, exactly as k_perceive's prologue builds it from the x / y / theta just stored
            // (centre shifted by fwd_delta along the heading, RL_api.py:100-108): that launch then only loads it (ACT_FRAMES)
            const double *fs = reinterpret_cast<const double *>(smem + fw->frm_off) + 2 * i;
            AntFrame f;
            const bool shifted = p.fwd_delta != 0.0;
            f.cx = shifted ? x + fw->cs * p.fwd_delta : x;
            f.cy = shifted ? y + fw->sn * p.fwd_delta : y;
            f.ct = fs[0];
            f.st = fs[1];
            p.s.frames[eN + i] = f;
        }
        if (fw) break;
    }
    if (fw) UM_STAMP(13);
}

template <int C>
__global__ void __launch_bounds__(1024)
k_move(const KP p, const int8_t *__restrict__ rotation, const int8_t *__restrict__ phero_act,
       uint8_t *__restrict__ done, const int do_step, const uint32_t seq)
{
    extern __shared__ __align__(16) unsigned char smem[];
    move_body<C>(p, env_of_block(blockIdx.x, p.E, seq), rotation, phero_act, done, do_step, seq, smem);
}

// Environment.update of step t (deferred by the host, include/antsrl.h "deferred update") and the move of step t + 1
// in ONE launch: both are one-workgroup-per-environment, one-ant-per-thread kernels, and the second re-reads what the
// first has just written — the ant's state and the record of its cell (the deposit cell IS the food cell of the next
// mandible decision).  Back to back in one workgroup those values travel in registers (UmFwd), and one launch ramp and
// tail go away.  What the kernel's time is made of is a chain of exposed latencies, not bytes: profiles/um_trace.py,
// DESIGN.md section 5.1.  Same device functions as k_update_one / k_move:
// the results are bit-identical to the two launches (tests/test_gpu_parity.py::test_deferred_update_is_bit_identical).
// (1024, 8): at most 64 VGPRs (the kernel needs 66 without the bound and fits without a spill) — 8 waves per SIMD, i.e. FOUR
// 512-thread workgroups per CU: c3's 1024 environments are resident at once instead of 768 + a 256-workgroup second round
// (profiles/r04/um_trace*.txt); c4's 1024-thread workgroups two per CU instead of one.
template <int C, bool ILV>
__global__ void __launch_bounds__(1024, 8)
k_update_move(const KP p, const int out_buf, const double g_dep, const double inv_g_dep, const int8_t *__restrict__ rotation,
              const int8_t *__restrict__ phero_act, uint8_t *__restrict__ done, const uint32_t seq, const int with_frames)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int e = env_of_block(blockIdx.x, p.E, seq);
    UM_STAMP(0);
    UmFwd fw = {};
    { // the move's own per-ant inputs, ahead of the update: their memory round trip rides with the update's loads
        const size_t a = (size_t)e * p.N + (threadIdx.x < (unsigned)p.N ? threadIdx.x : 0);
        fw.hold = p.s.holding[a];
        fw.m = STP_LD(p.s.mandibles[a]);
        if (rotation) fw.rot = STP_LD(rotation[a]);
        if (phero_act) fw.pa = STP_LD(phero_act[a]);
        fw.has_rot = rotation != nullptr;
        fw.primed = p.s.reward_primed[e]; // (every thread, one address: a branch around a load would carry its own wait)
    }
    fw.rec = (ILV && C == 2 && !(UM_ABL & 4)) ? 1 : 0; // (interleaved records: the update forwards the cell's food / META words)
    // (with_frames: the host's choice per launch — antsrl_capi.hip, meta_observe; 0 = no frame is built, nothing is parked)
    fw.frm_off = with_frames ? (uint32_t)align_up(max(update_one_lds_bytes(p.HT, p.R, (int)(blockDim.x >> 6), p.N), move_lds_bytes(p.HT, p.N)), 16) : 0u;
    update_one_body<C, ILV>(p, e, nullptr, out_buf, smem, g_dep, inv_g_dep, &fw);
    __syncthreads(); // the update's global writes are visible to the whole workgroup; its LDS is dead
    UM_STAMP(9);
    move_body<C>(p, e, rotation, phero_act, done, 1, seq, smem, &fw);
#ifdef UM_TRACE
    __builtin_amdgcn_s_waitcnt(0x0F70); // (the trace build waits for the stores' acknowledgement: slot 14 - slot 13)
    UM_STAMP(14);
    if (threadIdx.x == 0 && blockIdx.x < UM_TRACE_MAX_WGS) g_um_trace[blockIdx.x * UM_TRACE_SLOTS + 15] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); // HW_ID
#endif
}

// ---------------------------------------------------------------------------------------------------
// k_perceive
// ---------------------------------------------------------------------------------------------------
// LDS: rock table of the workgroup's environment [R][4] doubles, then per wave: frames [run] (written by the workgroup's
// prologue wave), rockmask [run] (by the wave's own lanes, one ant per lane), staging (two rows back to back + alignment
// slack).
struct PrcOff {
    uint32_t rock, wave0, frame, rm, stage, per_wave, stride;
    size_t total;
};

__host__ __device__ __forceinline__ PrcOff prc_offsets(int run, int PP, int K, int R, int nwaves, bool no_stage = false, uint32_t pitch = 0)
{
    PrcOff o;
    o.rock = 0;
    o.wave0 = (uint32_t)(32 * (size_t)(R > 0 ? R : 1));
    o.frame = 0;
    o.rm = (uint32_t)(sizeof(AntFrame) * (size_t)run);
    // (no rocks: no rock masks — 128 bytes per workgroup that decide the in-loop policy launch's residency: its tile image
    //  brings a workgroup to 23 488 bytes with them, 23 360 without, and the CU's 160 KB admit SEVEN workgroups up to 23 408)
    o.stage = (uint32_t)align_up(o.rm + (R > 0 ? 4 * (size_t)run : 0), 16);
    const size_t rowf = (size_t)PP * K;
    o.stride = (uint32_t)((2 * rowf + 32 + 3) / 4 * 4); // floats: misalignment / carry (< one 128-byte line) + two rows
    if (2 * pitch > o.stride) o.stride = 2 * pitch;     // padded rows (PAD): two rows at the observation's row pitch
    o.per_wave = (uint32_t)align_up(o.stage + (no_stage ? 0 : 4 * (size_t)o.stride), 16); // (POLICY: the tile image is the staging)
    o.total = o.wave0 + (size_t)nwaves * o.per_wave;
    return o;
}

// POLICY: bf16 elements of the workgroup's tile image: up to 7 elements of alignment shift, 32 rows, and what the last
// k-step's 5-dword read can reach past them
__host__ __device__ __forceinline__ uint32_t prc_policy_img_elems(int row) { return (uint32_t)((32 * row + 8 + 32 + 7) / 8 * 8); }

#define PRC_TPB 256 // 4 waves per workgroup; 64 / 128 measured no better (profiles/r02/prc_tpb.txt)

// Which ant of the workgroup's tile the j-th ant of wave `w` is: every wave takes one contiguous run.  (The waves taking
// alternating 2-ant groups — one dense advancing window per workgroup — measured 2 % slower: profiles/r03/ilv_ab.txt.)
__device__ __forceinline__ int prc_tile_ant(const int w, const int j, const int run, const int nwaves)
{
    (void)nwaves;
    return w * run + j;
}

// ---------------------------------------------------------------------------------------------------
// In-loop policy (antsrl_set_inloop_policy, BASELINE config 5): the reference's linear DQN net (antsrl_policy.hip) on the 32
// rows a workgroup has just written, straight from its LDS image of them — the standalone kernel re-reads the whole
// observation tensor from HBM for the same arithmetic.  The consumer of k_policy_flat verbatim (same fragments, same order
// of MFMAs and additions: the actions are bit-identical to antsrl_policy_mlp on the stored rows); W1's A fragments come from
// the pre-packed, L2-resident DState::pol_pack (k_policy_pack) instead of a per-workgroup LDS copy.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 prc_bf16x8;
typedef __attribute__((ext_vector_type(16))) float prc_f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t prc_u4;

__device__ __forceinline__ void policy_tile(const PolArgs &pol, const uint16_t *img, const uint32_t img_shift, const float *as_lds,
                                            const int F, const int rows, const size_t ant0, const int lane)
{
    const int r = lane & 31, h = lane >> 5;
    const prc_u4 *wpack = reinterpret_cast<const prc_u4 *>(pol.pack);
    const prc_u4 *a2pack = reinterpret_cast<const prc_u4 *>(pol.pack + ANTSRL_POL_WPACK_BYTES);
    const float *lanepack = reinterpret_cast<const float *>(pol.pack + ANTSRL_POL_WPACK_BYTES + ANTSRL_POL_A2PACK_BYTES);
    const int rc = min(r, rows - 1);
    const float as0 = (float)(__bf16)as_lds[2 * rc], as1 = (float)(__bf16)as_lds[2 * rc + 1];
    prc_f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
    const uint32_t off0 = img_shift + (uint32_t)r * (uint32_t)F + 8u * h; // (element offset from the 16-byte aligned image base)
    const int ks = pol.ks;
    // W1's fragments stream from L2 POL_AHEAD k-steps ahead of the MFMA that takes them (the chain is the workgroup's
    // tail: its other waves are done, so the L2 round trips are not hidden by anything else)
    constexpr int POL_AHEAD = 6;
    prc_u4 a[POL_AHEAD];
#pragma unroll
    for (int i = 0; i < POL_AHEAD; ++i) a[i] = wpack[min(i, ks - 1) * 64 + lane];
    for (int s0 = 0; s0 < ks; s0 += POL_AHEAD) {
#pragma unroll
        for (int i = 0; i < POL_AHEAD; ++i) {
            const int s = s0 + i;
            const prc_u4 a_s = a[i];
            a[i] = wpack[min(s + POL_AHEAD, ks - 1) * 64 + lane];
            if (s < ks) {
                const uint32_t off = off0 + 16u * s, sh = (off & 1u) * 16u;
                const uint32_t *d = reinterpret_cast<const uint32_t *>(img) + (off >> 1);
                const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
                const prc_u4 packed = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                                       __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(prc_bf16x8, a_s),
                                                              __builtin_bit_cast(prc_bf16x8, packed), acc, 0, 0, 0);
            }
        }
    }
    prc_f32x16 acc2;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc2[g] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        prc_bf16x8 hfrag;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = 8 * s + j;
            const float bias1 = lanepack[g * 64 + lane], was0 = lanepack[(16 + g) * 64 + lane], was1 = lanepack[(32 + g) * 64 + lane];
            hfrag[j] = (__bf16)(acc[g] + (as0 * was0 + as1 * was1) + bias1);
        }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(prc_bf16x8, a2pack[s * 64 + lane]), hfrag, acc2, 0, 0, 0);
    }
    float lg[6];
    lg[0] = acc2[0]; lg[1] = acc2[1]; lg[2] = acc2[2]; lg[3] = acc2[3];
    lg[4] = __shfl(acc2[0], r + 32); lg[5] = __shfl(acc2[1], r + 32);
#pragma unroll
    for (int o = 0; o < 6; ++o) lg[o] += lanepack[48 * 64 + o];
    if (h == 0 && r < rows) {
        int ar = 0, ap = 0; // torch.max(...).indices: first maximum wins
        if (lg[1] > lg[ar]) ar = 1;
        if (lg[2] > lg[ar]) ar = 2;
        if (lg[4] > lg[3 + ap]) ap = 1;
        if (lg[5] > lg[3 + ap]) ap = 2;
        pol.rot[ant0 + r] = (int8_t)(ar - 1);
        if (pol.ph) pol.ph[ant0 + r] = (int8_t)ap;
    }
}

// Variant build only (-DPRC_TRACE; profiles/prc_trace.py): the time line of every wave of k_perceive's last launch, in
// 10 ns ticks (s_memrealtime): 0 entry, 1 past the prologue's barrier, 2 first gathers back (in front of the first group's work),
// 3..6 behind group 1..4 of the first chunk (stores issued), 7 loop done, 8 every store acknowledged; 9 = HW_ID, 10 = XCC_ID;
// POLICY: 11 behind the barrier in front of the in-loop net, 12 (wave 0) the net's actions are out.  The
// stamps are kept in LDS (the launch needs ANTSRL_PRC_LDS_PAD >= 1) and written out by the wave's last instructions.
#ifdef UM_TRACE
extern "C" int antsrl_debug_read_um_trace(uint32_t *dst, int n_wgs)
{
    if (!dst || n_wgs < 0 || n_wgs > UM_TRACE_MAX_WGS) return ANTSRL_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ANTSRL_E_DEVICE;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_um_trace), sizeof(uint32_t) * UM_TRACE_SLOTS * (size_t)n_wgs) == hipSuccess
               ? ANTSRL_OK : ANTSRL_E_DEVICE;
}
#endif
#ifdef PRC_TRACE
#define PRC_TRACE_SLOTS 16
#define PRC_TRACE_MAX_WAVES (1 << 17)
__device__ uint32_t g_prc_trace[PRC_TRACE_SLOTS * PRC_TRACE_MAX_WAVES];
#define PRC_STAMP(slot)                                                                       \
    do {                                                                                      \
        if (lane == 0) prc_tr[(slot)] = (uint32_t)wall_clock64();                             \
    } while (0)
extern "C" int antsrl_debug_read_prc_trace(uint32_t *dst, int n_waves)
{
    if (!dst || n_waves < 0 || n_waves > PRC_TRACE_MAX_WAVES) return ANTSRL_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ANTSRL_E_DEVICE;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_prc_trace), sizeof(uint32_t) * PRC_TRACE_SLOTS * (size_t)n_waves) == hipSuccess
               ? ANTSRL_OK : ANTSRL_E_DEVICE;
}
#else
#define PRC_STAMP(slot) do { } while (0)
#endif

// HAS_OBS is a template parameter on purpose: with the observation stores behind a run-time branch the
// compiler cannot count them, every wait on a gather becomes vmcnt(0), i.e. a wait for the previous
// group's observation stores to be acknowledged by memory — stores and everything else then add up instead
// of overlapping (measured: 0.245 ms against 0.092 ms without the stores).
// POLICY (with OBS16 only): the workgroup also keeps the bf16 rows of its ants as one contiguous LDS image and its first
// wave evaluates the in-loop policy on them at the end (policy_tile).  POLICY without HAS_OBS is the act-only rollout
// (collect_agent_memory.py:189-199 with training=False needs the actions, nothing else): the rows exist in LDS only —
// same image, same MFMAs, so the actions are bit-identical to the launch that also writes the tensor.
// PAD (antsrl_set_obs_row_stride): the observation rows lie `pitch` elements apart, a whole number of 128-byte lines
// (float32 7x7x7: 343 -> 352 elements = 11 lines): every 2-row group is then whole lines of this wave's own, copied out
// with aligned 16-byte stores only — no misalignment, no edge store, no line shared with another wave.  The padding
// elements are written as zeros.  Same values in the same [E][N][P][P][K] positions; the dense layout is the default.
// Residency is set by the SCALAR registers here: 256-thread workgroups are admitted per CU up to
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) (MI355X_MICROARCH.md, "Residency and cooperative launch": <= 80 SGPRs -> 8,
// 82-96 -> 7, 98+ -> 6).  Left alone the compiler takes 102-106 SGPRs — six workgroups per CU whatever the 66-70 VGPRs would
// allow (rounds 2-4 read "7 waves per SIMD" off the VGPR count; the c5 time line's "5.9 workgroups per CU" was this limit).
// Capped at 96 (2-8 of them spilled to VGPR lanes, no vector spill) with the vector budget of seven waves per SIMD (72):
// SEVEN workgroups per CU for real — same-device A/B, ms/step: c3 0.1985 -> 0.1957, c5 0.0847 -> 0.0827, act-only 0.0752 ->
// 0.0745, c2 / c4 +-0 (profiles/r05/sgpr_cap_ab.txt).  Eight (cap 80 + 64 VGPRs) spills vector registers and loses 4-11 %.
#ifndef PRC_SGPR_CAP
#define PRC_SGPR_CAP 96
#endif
#ifndef PRC_MIN_WAVES
#define PRC_MIN_WAVES 7
#endif
#define PRC_SGPR_ATTR __attribute__((amdgpu_num_sgpr(PRC_SGPR_CAP)))
#ifdef ANTSRL_PROFILING
// Placement probe (profiles/r05/env_pitch_probe.py): extra bytes between two environments' blocks of observation rows — the
// caller's buffer is E * (N * row bytes + pad) then.  A pitch knob for measurements; the product's tensor is dense.
__device__ uint32_t g_prc_env_pad;
extern "C" int antsrl_debug_set_obs_env_pad(uint32_t bytes)
{
    if (bytes % 16) return ANTSRL_E_INVALID;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prc_env_pad), &bytes, sizeof(bytes)) == hipSuccess ? ANTSRL_OK : ANTSRL_E_DEVICE;
}
#endif

template <int LAYOUT, bool OBS16, bool ILV, bool HAS_OBS, bool POLICY = false, bool PAD = false>
__global__ void __launch_bounds__(PRC_TPB, PRC_MIN_WAVES) PRC_SGPR_ATTR
k_perceive(const KP p, const float *__restrict__ cells, float *__restrict__ obs_arg, float *__restrict__ agent_state,
           float *__restrict__ reward, const int flags, const uint32_t seq, const int run, const int nseg, const PolArgs pol,
           const uint32_t pitch_arg)
{
    static_assert(!POLICY || OBS16, "the in-loop policy reads bfloat16 rows");
    static_assert(!PAD || (HAS_OBS && !POLICY), "padded rows: an observation tensor, no in-loop policy image");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int C = 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nwaves = PRC_TPB / 64;
    const int N = p.N, W = p.W, H = p.H, K = p.K, P = p.P, PP = p.PP, R = p.R;
    const size_t G = (size_t)W * H;
#ifdef PRC_TRACE
    uint32_t *prc_tr = reinterpret_cast<uint32_t *>(smem + align_up(prc_offsets(run, PP, K, R, nwaves, POLICY, PAD ? pitch_arg : 0u).total, 16) +
                                                    (POLICY ? 2 * (size_t)prc_policy_img_elems(PP * K) + 4 * 64 : 0)) + wave * PRC_TRACE_SLOTS;
    if (lane < PRC_TRACE_SLOTS) prc_tr[lane] = 0u;
    PRC_STAMP(0);
    if (lane == 0) {
        prc_tr[9] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
        prc_tr[10] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); // HW_REG_XCC_ID
    }
#endif
    // (environment, segment) of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (b and
    // b + 8 share one): all segments of an environment go to ONE XCD, back to back, so the cell records its
    // ants share are fetched into one L2 (speed only — nothing depends on the placement).
    int e, seg;
    {
        const int b = blockIdx.x;
        if ((p.E & 7) == 0) {
            const int j = b >> 3;
            e = (j / nseg) * 8 + (b & 7);
            seg = j % nseg;
        } else {
            e = b / nseg;
            seg = b % nseg;
        }
        e = env_of_block(e, p.E, seq); // (odd observations from the other end: antsrl_util.h)
    }
#ifdef ANTSRL_PROFILING
    float *__restrict__ obs = HAS_OBS ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(obs_arg) + (size_t)e * g_prc_env_pad) : obs_arg;
#else
    float *__restrict__ obs = obs_arg;
#endif
    const PrcOff lo = prc_offsets(run, PP, K, R, nwaves, POLICY, PAD ? pitch_arg : 0u);
    double *rock = (double *)(smem + lo.rock);
    unsigned char *wbase = smem + lo.wave0 + (size_t)wave * lo.per_wave;
    AntFrame *frames = (AntFrame *)(wbase + lo.frame);
    uint32_t *rmask = (uint32_t *)(wbase + lo.rm);
    float *stage = (float *)(wbase + lo.stage);
    // POLICY: [tile rows][row] bf16, contiguous like the rows in memory (+ a zeroed pad the last k-step reads into), then
    // {holding, seed} of the tile's ants
    uint16_t *pol_img = reinterpret_cast<uint16_t *>(smem + align_up(lo.total, 16));
    float *pol_as = reinterpret_cast<float *>(pol_img + prc_policy_img_elems(PP * K));
    // The image doubles as the copy-out staging of all four waves: it starts at the tile's own 16-byte misalignment in
    // the observation tensor, so LDS and global addresses of every row agree modulo 16 bytes (no per-wave staging rows:
    // a third less LDS per workgroup, one LDS write per value).
    const size_t tile_elem0 = ((size_t)e * N + (size_t)(seg * nwaves * run)) * (size_t)(PP * K);
    uint16_t *tile0 = pol_img + (uint32_t)((((uintptr_t)obs >> 1) + tile_elem0) & 7);
    (void)tile0;
    if constexpr (POLICY) {
        // What the net reads and no row store covers must be zero: the rows of tile slots past the environment's end, the
        // alignment shift in front of row 0 and the pad the last k-step reads into.  Every wave zeroes the rows of ITS OWN
        // empty slots (element-wide stores: a row's neighbours belong to other waves), wave 0 the shift and the pad — no
        // wave ever stores into bytes another wave stages, so the zeroing needs no barrier in front of the staging (the
        // barrier in front of policy_tile orders all of it).  Until round 4 the whole workgroup cleared the whole image
        // here, thread t the pieces t + 256 k: behind ACT_FRAMES without rocks the prologue has no workgroup barrier, and a
        // lagging wave's zeros could land on rows another wave had already staged.
        const uint32_t rowe = (uint32_t)PP * (uint32_t)K, n_img = prc_policy_img_elems(PP * K);
        const int t0_ = seg * nwaves * run;
        for (int j = 0; j < run; ++j) {
            if (t0_ + wave * run + j < N) continue; // (wave-uniform)
            uint16_t *rz = tile0 + (uint32_t)(wave * run + j) * rowe;
            for (uint32_t i = lane; i < rowe; i += 64) rz[i] = 0;
        }
        if (wave == 0) {
            const uint32_t sh = (uint32_t)(tile0 - pol_img), tail0 = sh + (uint32_t)(nwaves * run) * rowe;
            if ((uint32_t)lane < sh) pol_img[lane] = 0;
            for (uint32_t i = tail0 + lane; i < n_img; i += 64) pol_img[i] = 0;
        }
    }

    const size_t eN = (size_t)e * N;
    const int t_begin = seg * nwaves * run; // first ant of the workgroup's tile
    // ant of this wave's j-th slot: t_begin + prc_tile_ant(wave, j, ...), increasing in j; n_run = slots inside the env
    int n_run = 0; // wave-uniform, 0 for a wave past the end of the environment
    for (int j = 0; j < run; ++j) n_run += (t_begin + prc_tile_ant(wave, j, run, nwaves) < N) ? 1 : 0;
#define PRC_ANT(j) (t_begin + prc_tile_ant(wave, (j), run, nwaves))

    // ---- prologue: rock table of the environment, the perception frames of the tile's ants (RL_api.py:100-108: centre
    // shifted by `fwd_delta` along the heading, cos / sin of theta + pi/2), this wave's rock masks.
    // ONE wave computes the frames of the whole tile, one (ant, which-sincos) task per lane — 32 ants x {rotation, centre}
    // fill its 64 lanes — while another wave fetches the rock table; one barrier behind both.  (Until round 3 every wave
    // computed the frames of its own 8 ants behind the rock table's barrier.)  The workgroup's start-up chain loses a memory
    // round trip (ant state and rock table travel together) and a sincos, and the two float64 sincos — ~400 VALU instructions
    // per wave whether 8 lanes are active or 64, 51 of the kernel's 119 VALU instructions per ant — are issued once per
    // workgroup, not four times.  Same instructions on the same inputs: the frames are bit-identical.
    // Same-box A/B on k_perceive (profiles/r03/prologue_ab.txt): c3 -2.7 %, c4 -1.5 %, c5 -2.5 %, small batches with rocks
    // -3 ... -9 %, without rocks (nothing to overlap with) +-0.
    // The epilogue's inputs are fetched HERE, in front of the prologue's barrier (which waits for them: from there on the
    // compiler knows them complete).  Fetched at the wave's end they are one more memory round trip during which the wave
    // keeps its slot and its workgroup's LDS for nothing — and, vector-memory operations retiring in order, a wait for every
    // observation store of the run; fetched in front of the loop without a wait the compiler can see, the epilogue's first
    // use of them became `vmcnt(0)`, the same wait.  (Clamped lanes: every lane loads a valid ant.)
    const size_t a_pre = eN + (size_t)min(PRC_ANT(min(lane, max(n_run, 1) - 1)), N - 1);
    const float pre_hold = p.s.holding[a_pre];
    const float pre_seed = (agent_state || POLICY) ? p.s.seed[a_pre] : 0.0f;
    {
        // the two waves rotate with the workgroup index (wave 0 has the net at the end of a POLICY launch)
        const uint32_t wsel = (blockIdx.x >> 3) + (blockIdx.x >> 8);
        const int w_pro = (POLICY && nwaves > 1) ? 1 + (int)(wsel % (nwaves > 1 ? nwaves - 1 : 1)) : (int)(wsel % nwaves);
        const int w_rock = (w_pro + 1) % nwaves;
        // ACT_FRAMES (the launch follows a k_update_move): the frames exist already — that kernel had the new position,
        // the heading's sine and cosine and the action in registers and evaluated cos / sin(theta + pi/2) under its record
        // load.  Every wave loads the frames of its own ants (32 bytes each): no sincos, no wave waiting for another one's,
        // and without rocks no workgroup barrier at all in front of the first gathers.  Bit-identical by construction (the
        // same expressions on the same stored x / y / theta): test_deferred_update_is_bit_identical compares the two forms.
        const bool have_frames = (flags & ACT_FRAMES) != 0; // (launch-uniform)
        if (have_frames && lane < n_run) frames[lane] = p.s.frames[eN + (size_t)PRC_ANT(lane)];
        if (wave == w_rock) {
            for (int q = lane; q < R; q += 64) {
                const double rad = p.s.rock_r[(size_t)e * R + q];
                rock[4 * q + 0] = p.s.rock_cx[(size_t)e * R + q];
                rock[4 * q + 1] = p.s.rock_cy[(size_t)e * R + q];
                rock[4 * q + 2] = rad;
                rock[4 * q + 3] = sqrt_lt_threshold(rad);
            }
        }
        if (wave == w_pro && !have_frames) {
            const bool fwd = p.fwd_delta != 0.0;
            const int n_slot = nwaves * run, n_task = fwd ? 2 * n_slot : n_slot;
            for (int t = lane; t < n_task; t += 64) {
                const int which = t >= n_slot ? 1 : 0; // 0: the rotation (and the centre when it is the ant's own cell), 1: the shifted centre
                const int sl = which ? t - n_slot : t, w2 = sl / run, j2 = sl - w2 * run;
                const int ant = t_begin + prc_tile_ant(w2, j2, run, nwaves);
                if (ant < N) {
                    const size_t a = eN + (size_t)ant;
                    AntFrame *fr = reinterpret_cast<AntFrame *>(smem + lo.wave0 + (size_t)w2 * lo.per_wave + lo.frame) + j2;
                    // (all three loads in front of the sincos: one memory round trip, not two)
                    const double th = p.s.theta[a], x = p.s.x[a], y = p.s.y[a];
                    double sn, cs;
                    sincos(which ? th : th + PI_D * 0.5, &sn, &cs);
                    if (which) {
                        fr->cx = x + cs * p.fwd_delta;
                        fr->cy = y + sn * p.fwd_delta;
                    } else {
                        fr->ct = cs;
                        fr->st = sn;
                        if (!fwd) {
                            fr->cx = x;
                            fr->cy = y;
                        }
                    }
                }
            }
        }
        if (!have_frames || R > 0) __syncthreads(); // (the rock table and every wave's frames are complete)
        else wave_lds_sync();                       // (this wave's own frames)
        // the rocks whose disc can reach the patch (conservative; the exact test runs per cell below)
        if (lane < n_run) {
            uint32_t rm = 0u;
            if (R > 0) {
                const double xf = frames[lane].cx, yf = frames[lane].cy;
                const double margin = (double)p.r * p.delta * 1.4142135623730951 + 1.5;
                const bool border = xf - margin < 0 || yf - margin < 0 || xf + margin >= W || yf + margin >= H;
                for (int q = 0; q < R; ++q) {
                    const double dx = rock[4 * q + 0] - xf, dy = rock[4 * q + 1] - yf;
                    const double rr = rock[4 * q + 2] + margin;
                    if (border || dx * dx + dy * dy < rr * rr) rm |= 1u << q;
                }
            }
            if (R > 0) rmask[lane] = rm;
        }
    }
    PRC_STAMP(1);
    wave_lds_sync();
    if (n_run <= 0) { // (no barrier below: waves run independently from here on — but for the policy's hand-over)
        if constexpr (POLICY) __syncthreads();
        return;
    }

    const float *ph = cells + (size_t)e * G * (size_t)p.ps;
    const float *fm = p.s.food + (size_t)e * G * p.fs; // {food, META} records when not interleaved
    uint32_t *metaw = reinterpret_cast<uint32_t *>(p.s.food + (size_t)e * G * p.fs) + 1;
    const size_t FS = (size_t)p.fs;
    const bool explore = p.explore_on != 0;
    constexpr bool has_obs = HAS_OBS;
    constexpr bool has_rows = HAS_OBS || POLICY; // the perceived values are materialised (staging / tile image)
    const float inv_max = 1.0f / (float)p.max_val;
    const float g_now = (float)p.g_now;                       // scaled mode: v = u * f0^S ...
    const float cut = p.scaled ? (float)p.threshold : 0.0f;   // ... and 0 below the 0.01 cut
    const uint32_t row = (uint32_t)PP * (uint32_t)K;            // elements per ant
    const bool wrap_fast = W > 4 * (p.r + 4) && H > 4 * (p.r + 4) && p.fwd_delta < W / 4 && p.fwd_delta < H / 4 &&
                           p.fwd_delta > -W / 4 && p.fwd_delta > -H / 4 && p.delta < 2.0;
    const bool wrap_pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0;

    // ---- perception: one LANE per perceived cell (49 of 64 lanes at the reference's 7x7): the cell's
    // offsets, mask bit and output slot are per-lane constants, the ant's frame is wave-uniform (LDS
    // broadcast).  Software-pipelined by one group of PRC_UNROLL ants; everything that touches global memory
    // is straight-line and unconditional (out-of-range ants / lanes are clamped onto valid ones and redo
    // identical work: stores of the same value to the same address), so the compiler can count outstanding
    // operations: the wait for group g's gathers leaves group g+1's gathers and group g-1's stores in flight.
    const int q = lane < PP ? lane : PP - 1; // lanes beyond the perception clamp onto its last cell
    const double of_px = (double)(q % P - p.r) * p.delta; // coords[a][b] = (arange[b], arange[a]) * DELTA, RL_api.py:92-93
    const double of_py = (double)(q / P - p.r) * p.delta;
    const bool mask_q = p.has_mask ? p.mask[q] != 0 : true;
    const uint32_t qK = (uint32_t)q * K;
    // A lane whose gathered record is never used (a masked cell when no reward counts explored cells) takes
    // the address of the wave's first needed lane: the CU's address path works through a scattered gather at
    // about one distinct address per clock, and a lane on an address that is fetched anyway costs nothing.
    const bool own = mask_q || explore;
    const unsigned long long need_mask = __ballot(own);
    const int src_lane = own ? lane : (need_mask ? __builtin_ctzll(need_mask) : 0);

    // food / META record index of cell (ix, iy) (frec_xy in antsrl_util.h): blocks of 2 x 4 cells per 128-byte line of the
    // interleaved 16-byte records (KP::tiled: the one gather), blocks of 4 x 4 cells of the 8-byte {food, META} records beside
    // row-major pheromone buffers (KP::ftile: the second gather; the first one, `pc_`, is row-major), else row-major
    const bool tiled = p.tiled != 0, ftile = p.ftile != 0;
#define PRC_SLOT(ix, iy) (tiled ? tiled_slot((ix), (iy), H) : ftile ? tiled44_slot((ix), (iy), H) : (uint32_t)((ix) * H + (iy)))
#define PRC_LOAD4(ptr) (*reinterpret_cast<const stream_f4 *>(ptr)) // (cached: as nt loads the gathers lose their L1 hits, +5 %)
#define PRC_FETCH(G0, GRP)                                                                               \
    {                                                                                                    \
        _Pragma("unroll") for (int u = 0; u < PRC_UNROLL; ++u)                                           \
        {                                                                                                \
            const int j_ = min((G0) + u, n_run - 1);                                                     \
            const AntFrame fr = frames[j_]; /* wave-uniform address: LDS broadcast */                    \
            const double rx = fr.ct * of_px - fr.st * of_py; /* RL_api.py:110-111 */                     \
            const double ry = fr.st * of_px + fr.ct * of_py;                                             \
            int ix = (int)rint(rx + fr.cx);                              /* :114-117 half to even */     \
            int iy = (int)rint(ry + fr.cy);                                                              \
            if (wrap_pow2) { /* :118-119; two's complement AND is the floor-mod for a power of two */    \
                ix &= W - 1; iy &= H - 1;                                                                \
            } else if (wrap_fast) { /* |ix| < 2W: unsigned min picks the in-range candidate */           \
                ix = (int)min(min((uint32_t)ix, (uint32_t)(ix + W)), (uint32_t)(ix - W));                \
                iy = (int)min(min((uint32_t)iy, (uint32_t)(iy + H)), (uint32_t)(iy - H));                \
            } else {                                                                                     \
                ix = wrap_index(ix, W); iy = wrap_index(iy, H);                                          \
            }                                                                                            \
            GRP.cell[u] = PRC_SLOT(ix, iy);                                                              \
            if (!ILV) GRP.pc[u] = (uint32_t)(ix * H + iy); /* the row-major pheromone buffers' index */  \
        }                                                                                                \
        _Pragma("unroll") for (int u = 0; u < PRC_UNROLL; ++u)                                           \
        {                                                                                                \
            const uint32_t gc_ = abl_gather ? (uint32_t)lane : (uint32_t)__shfl((int)GRP.cell[u], src_lane); \
            if (ILV) { /* one {p0, p1, food, META} record per cell: a single 16-byte gather */     \
                const stream_f4 t = PRC_LOAD4(ph + (size_t)gc_ * 4);                                     \
                GRP.pv[u][0] = t.x; GRP.pv[u][1] = t.y; GRP.fd[u] = t.z; GRP.mt[u] = __float_as_uint(t.w); \
            } else {                                                                                     \
                const uint32_t pc_ = abl_gather ? (uint32_t)lane : (uint32_t)__shfl((int)GRP.pc[u], src_lane); \
                const float2 t = *reinterpret_cast<const float2 *>(ph + (size_t)pc_ * 2);                \
                const float2 f = *reinterpret_cast<const float2 *>(fm + (size_t)gc_ * 2);                \
                GRP.pv[u][0] = t.x; GRP.pv[u][1] = t.y; GRP.fd[u] = f.x; GRP.mt[u] = __float_as_uint(f.y); \
            }                                                                                            \
        }                                                                                                \
    }
    // one group of PRC_UNROLL ants in flight: cell indices and the gathered record of this lane's cell
    // (the cell's coordinates are not kept: only the rock test needs them, for the few cells whose ant has a rock in reach —
    //  recomputed there from the frame, the same operations on the same values; four registers per set less)
    struct PrcGrp {
        uint32_t cell[PRC_UNROLL], mt[PRC_UNROLL], pc[ILV ? 1 : PRC_UNROLL];
        float pv[PRC_UNROLL][C], fd[PRC_UNROLL];
    };
    PrcGrp gA, gB, gC;
    uint32_t cntv = 0u; // lane j: unexplored cells in the patch of the wave's j-th ant
    // copy-out state: elements per 128-byte line / per 16 bytes, the run's first row, the aligned line the LDS
    // image currently starts at, and how many image elements in front of the next row are already taken
    constexpr uint32_t LINE = OBS16 ? 64u : 32u, VEC = OBS16 ? 8u : 4u, ESZ = OBS16 ? 2u : 4u;
    const uint32_t pitch = PAD ? pitch_arg : row; // elements from one ant's row to the next
    unsigned char *run0 = reinterpret_cast<unsigned char *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(0)) * pitch * ESZ;
    uint32_t carry = (has_obs && !PAD) ? (uint32_t)(((uintptr_t)run0 & 15) / ESZ) : 0u;
    (void)LINE; (void)VEC;
    if constexpr (PAD) { // the padding elements of the two staged rows: zero, written once (the rows never reach them)
        const uint32_t npad = pitch - row;
        if ((uint32_t)lane < npad) {
            if (OBS16) {
                reinterpret_cast<uint16_t *>(stage)[row + lane] = 0;
                reinterpret_cast<uint16_t *>(stage)[pitch + row + lane] = 0;
            } else {
                stage[row + lane] = 0.0f;
                stage[pitch + row + lane] = 0.0f;
            }
        }
    }
    // ---- what happens to one group once its gathers are back: counts / marks, channel values, LDS staging, copy-out
    auto process = [&](const PrcGrp &g, const int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PRC_UNROLL; ++u) {
            const int j = min(j0 + u, n_run - 1);
            const bool real = (j0 + u < n_run) && lane < PP; // clamped duplicates must not count twice
            const uint32_t mt = g.mt[u];
            // reward_custom.py:19,22 (mask ignored): unexplored before THIS observation <=> stamp >= seq
            const bool unexp = real && explore && (mt >> META_STAMP_SHIFT) >= seq;
            const uint32_t n_un = (uint32_t)__popcll(__ballot(unexp));
            cntv = (lane == j0 + u) ? n_un : cntv; // (a clamped duplicate lands on a lane past the run: never read)
            // mark: the stamp half-word of the META word := seq.  A plain 2-byte store, no atomic: every writer of
            // this observation stores the same value, cells explored earlier (stamp < seq) are never written, and
            // a reader that still sees the old stamp counts the cell as unexplored just the same (stamp >= seq).
            if (unexp && !abl_mark)
                reinterpret_cast<uint16_t *>(metaw)[(size_t)g.cell[u] * FS * 2 + 1] = (uint16_t)((seq << 2) | ((mt >> 16) & 3u));
            if (has_rows) {
                float pvs[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v = g.pv[u][c];
                    if (p.scaled) {
                        v *= g_now;
                        v = v < cut ? 0.0f : v;
                    }
                    pvs[c] = v * inv_max; // :124-125, reciprocal multiply (pheromone channels are held to 1e-5)
                }
                const float v_ants = (mt & META_PRES_MASK) == seq ? 1.0f : 0.0f; // :142
                const float v_area = (mt & META_AREA) ? 1.0f : 0.0f;  // :130-131
                const float v_wall = (mt & META_WALL) ? 1.0f : 0.0f;  // :128-129
                float v_rock = 0.0f;                                  // :132-135
                if (LAYOUT == PLAYOUT_DEFAULT_ROCKS) {
                    uint32_t rm = mask_q ? rmask[j] : 0u;
                    bool any = false;
                    if (rm) { // (rare: a rock within reach of this ant's patch)
                        const AntFrame fr = frames[j];
                        int ix = (int)rint((fr.ct * of_px - fr.st * of_py) + fr.cx); // as in PRC_FETCH (RL_api.py:110-119)
                        int iy = (int)rint((fr.st * of_px + fr.ct * of_py) + fr.cy);
                        if (wrap_pow2) {
                            ix &= W - 1; iy &= H - 1;
                        } else if (wrap_fast) {
                            ix = (int)min(min((uint32_t)ix, (uint32_t)(ix + W)), (uint32_t)(ix - W));
                            iy = (int)min(min((uint32_t)iy, (uint32_t)(iy + H)), (uint32_t)(iy - H));
                        } else {
                            ix = wrap_index(ix, W); iy = wrap_index(iy, H);
                        }
                        while (rm) {
                            const int r = __builtin_ctz(rm);
                            rm &= rm - 1;
                            const double vx = (double)ix - rock[4 * r + 0];
                            const double vy = (double)iy - rock[4 * r + 1];
                            any |= vx * vx + vy * vy < rock[4 * r + 3]; // == sqrt(d2) < radius, see sqrt_lt_threshold
                        }
                    }
                    v_rock = any ? 1.0f : 0.0f;
                }
                const bool m = mask_q; // RL_api.py:147-148: mask*(p+1)-1 == -1 on masked cells
                if (OBS16) {
                    // bfloat16 observations: the same staging and copy-out on 2-byte elements (8 per 16 bytes)
                    // (POLICY: straight into the workgroup's tile image, row = the ant's index in the tile; a clamped
                    // duplicate of the run's last ant has no row of its own)
                    uint16_t *o16 = POLICY ? tile0 + (uint32_t)prc_tile_ant(wave, min(j0 + u, n_run - 1), run, nwaves) * row + qK
                                           : reinterpret_cast<uint16_t *>(stage) + carry + (uint32_t)u * pitch + qK;
                    o16[0] = bf16_bits(m ? v_ants : -1.0f); o16[1] = bf16_bits(m ? pvs[0] : -1.0f);
                    o16[2] = bf16_bits(m ? pvs[1] : -1.0f); o16[3] = bf16_bits(m ? v_area : -1.0f);
                    o16[4] = bf16_bits(m ? v_wall : -1.0f); o16[5] = bf16_bits(m ? g.fd[u] : -1.0f);
                    if (LAYOUT == PLAYOUT_DEFAULT_ROCKS) o16[6] = bf16_bits(m ? v_rock : -1.0f);
                } else {
                    // float32: the group's rows are staged back to back behind the carry, as they lie in memory
                    // (the image mirrors the destination modulo one 128-byte line), and flushed together below
                    float *o = stage + carry + (uint32_t)u * pitch + qK;
                    o[0] = m ? v_ants : -1.0f; o[1] = m ? pvs[0] : -1.0f; o[2] = m ? pvs[1] : -1.0f;
                    o[3] = m ? v_area : -1.0f; o[4] = m ? v_wall : -1.0f; o[5] = m ? g.fd[u] : -1.0f;
                    if (LAYOUT == PLAYOUT_DEFAULT_ROCKS) o[6] = m ? v_rock : -1.0f;
                }
            }
        }
        if (has_obs) { // the three-store copy-out (antsrl_flush.h: flush_plan_f32 / flush_plan_b16)
            // The group's rows are contiguous in memory and leave as ONE run: 16-byte stores over the interior
            // pieces (the first 128 / 64 of them start on a 128-byte line), one element-wide store for the edge
            // elements; lanes with nothing left repeat a valid store.  (The whole-line copy-out with a carry — every store
            // instruction covering whole 128-byte lines — measured slower at every stage of the kernel, 0.2444 against
            // 0.2212 ms last: DESIGN.md, profiles/r03/lines_ab.txt; the code is in profiles/r04/perceive_cleanup.patch.)
            wave_lds_sync();
            const uint32_t rowp = (j0 + 1 < n_run) ? 2u * row : row; // (odd tail of the run: one row)
            if constexpr (PAD) {
                // whole lines of this wave's own: the image IS the destination, 16-byte piece for piece
                const uint32_t n16 = ((j0 + 1 < n_run) ? 2u : 1u) * pitch / VEC, last = n16 - 1u;
                uint4 *d = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(j0)) * pitch * ESZ);
                const uint4 *sp = reinterpret_cast<const uint4 *>(stage);
                const uint32_t i1 = min((uint32_t)lane, last), i2 = min((uint32_t)lane + 64u, last), i3 = min((uint32_t)lane + 128u, last);
                const uint4 w1 = sp[i1], w2 = sp[i2];
                uint4 w3 = w2;
                if (!OBS16) w3 = sp[i3]; // (bfloat16: two rows are at most 96 pieces)
                if (!abl_store) {
                    store_stream(d + i1, w1);
                    store_stream(d + i2, w2);
                    if (!OBS16) store_stream(d + i3, w3);
                }
            } else if (OBS16) {
                uint16_t *dst16 = reinterpret_cast<uint16_t *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(j0)) * row;
                const uint32_t mis16 = (uint32_t)(((uintptr_t)dst16 >> 1) & 7);
                uint16_t *d_al = dst16 - mis16;
                // POLICY: the rows sit in the tile image at the same 16-byte phase as in memory
                uint16_t *st16 = POLICY ? tile0 + (uint32_t)prc_tile_ant(wave, j0, run, nwaves) * row - mis16 : reinterpret_cast<uint16_t *>(stage);
                const FlushPlanB16 f = flush_plan_b16((uint32_t)lane, mis16, rowp, (uint32_t)((uintptr_t)d_al >> 4) & 7u);
                const uint4 w1 = reinterpret_cast<const uint4 *>(st16)[f.g1];
                const uint4 w2 = reinterpret_cast<const uint4 *>(st16)[f.g2];
                const uint16_t we = st16[f.fe];
                if (!abl_store) {
                    store_stream(reinterpret_cast<uint4 *>(d_al) + f.g1, w1);
                    store_stream(reinterpret_cast<uint4 *>(d_al) + f.g2, w2);
                    store_stream(d_al + f.fe, we);
                }
                // the next group's 16-byte misalignment
                carry = (uint32_t)(((uintptr_t)(reinterpret_cast<uint16_t *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(j0 + 2)) * row) >> 1) & 7);
            } else {
                float *dst = reinterpret_cast<float *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(j0)) * row;
                const uint32_t mis = (uint32_t)(((uintptr_t)dst >> 2) & 3);
                float *dst_al = dst - mis;
                const FlushPlanF32 f = flush_plan_f32((uint32_t)lane, mis, rowp, (uint32_t)((uintptr_t)dst_al >> 4) & 7u);
                const float4 v1 = reinterpret_cast<const float4 *>(stage)[f.j1];
                const float4 v2 = reinterpret_cast<const float4 *>(stage)[f.j2];
                const float4 v3 = reinterpret_cast<const float4 *>(stage)[f.j3];
                const float ve = stage[f.fe];
                if (!abl_store) {
                    store_stream(reinterpret_cast<float4 *>(dst_al) + f.j1, v1);
                    store_stream(reinterpret_cast<float4 *>(dst_al) + f.j2, v2);
                    store_stream(reinterpret_cast<float4 *>(dst_al) + f.j3, v3);
                    store_stream(dst_al + f.fe, ve);
                }
                // the next group's 16-byte misalignment
                carry = (uint32_t)(((uintptr_t)(reinterpret_cast<float *>(obs) + ((size_t)e * N + (size_t)PRC_ANT(j0 + 2)) * row) >> 2) & 3);
            }
            wave_lds_sync();
        }
    };
    // Two groups ahead.  Straight-line code per chunk of four groups (8 ants: the shipped run length), three register
    // sets in rotation and no copy between them (a copy of a pending load's destination is a wait for it; a loop would
    // make the compiler merge the back edge's pending loads into the loop head and drain them there).  When group g is
    // consumed the gathers of g + 1 and g + 2 are in flight: the wait for g + 1's gathers — issued before the stores of
    // g - 1 — no longer waits for stores one iteration old but for those of g - 2.  (Groups past the run's end are
    // clamped onto its last ant: harmless re-reads; every chunk starts with no load pending.)
    for (int c0 = 0; c0 < n_run; c0 += 4 * PRC_UNROLL) {
        PRC_FETCH(min(c0, n_run - 1), gA)
        PRC_FETCH(min(c0 + PRC_UNROLL, n_run - 1), gB)
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt / lgkmcnt untouched
        if (c0 == 0) PRC_STAMP(2);
        PRC_FETCH(min(c0 + 2 * PRC_UNROLL, n_run - 1), gC)
        process(gA, c0);
        if (c0 == 0) PRC_STAMP(3);
        if (c0 + PRC_UNROLL < n_run) {
            PRC_FETCH(min(c0 + 3 * PRC_UNROLL, n_run - 1), gA)
            process(gB, c0 + PRC_UNROLL);
            if (c0 == 0) PRC_STAMP(4);
            if (c0 + 2 * PRC_UNROLL < n_run) {
                process(gC, c0 + 2 * PRC_UNROLL);
                if (c0 == 0) PRC_STAMP(5);
                if (c0 + 3 * PRC_UNROLL < n_run) {
                    process(gA, c0 + 3 * PRC_UNROLL);
                    if (c0 == 0) PRC_STAMP(6);
                }
            }
        }
    }
    PRC_STAMP(7);
#undef PRC_FETCH

    // ---- agent_state (RL_api.py:160-162), reward.observation hooks, give_reward: lane j <-> the wave's j-th ant
    if (lane < n_run) {
        const size_t a = eN + (size_t)PRC_ANT(lane);
        const float hold = pre_hold, seed = pre_seed;
        if (agent_state) {
            store_stream(agent_state + a * 2 + 0, hold);
            store_stream(agent_state + a * 2 + 1, seed);
        }
        if constexpr (POLICY) { // the net's two agent_state inputs (RL_api.py:160-162)
            pol_as[2 * prc_tile_ant(wave, lane, run, nwaves)] = hold;
            pol_as[2 * prc_tile_ant(wave, lane, run, nwaves) + 1] = seed;
        }
        double rw = 0.0;
        if (p.reward_kind == ANTSRL_REWARD_EXPLORATION) { // (no load on this path: the wave's last instructions are stores)
            rw = (double)cntv / 10.0; // reward_custom.py:19
        } else if (p.reward_kind != ANTSRL_REWARD_NONE) {
            const float prev_h = p.s.primed_cur[e] ? p.s.prev_holding[a] : hold;
            const double dh = (double)hold - (double)prev_h;
            if (p.reward_kind == ANTSRL_REWARD_FOOD) {
                rw = dh < 0 ? 10.0 : dh; // reward_custom.py:38-39
                p.s.prev_holding[a] = hold;
            } else { // All_Rewards, reward_custom.py:79-106
                const double r_food = dh < 0 ? 0.0 : dh;
                const double r_anthill = dh < 0 ? 1.0 : 0.0;
                p.s.prev_holding[a] = hold;
                if (explore) {
                    double re = (double)cntv / 10.0;
                    re = (hold == 0.0f) ? re * p.fct_explore : re * p.fct_explore_holding;
                    rw += re;
                }
                const double dx = p.s.x[a] - (double)p.s.anthill_xyr[3 * e + 0];
                const double dy = p.s.y[a] - (double)p.s.anthill_xyr[3 * e + 1];
                const double nd = sqrt(dx * dx + dy * dy);
                const double heading = (double)((p.s.prev_dist[a] > nd) && (hold > 0.0f)) * 0.1;
                p.s.prev_dist[a] = nd;
                rw += r_food * p.fct_food + r_anthill * p.fct_anthill + heading * p.fct_heading;
            }
        }
        if (reward) store_stream(reward + a, (float)rw);
        if ((flags & ACT_STEP) && rw - p.reward_threshold > 0) p.s.reward_state[a] = 255; // ants.py:119-121
    }
#undef PRC_ANT
#ifdef PRC_TRACE
    __builtin_amdgcn_s_waitcnt(0x0F70); // (the trace measures when the wave's stores are acknowledged, too: slot 8 - slot 7)
    PRC_STAMP(8);
#endif
    if constexpr (POLICY) {
        __syncthreads(); // every wave's rows and agent_state inputs are in the image
        PRC_STAMP(11);
        const int t0 = seg * nwaves * run; // first ant of this workgroup's tile
        if (wave == 0) {
            policy_tile(pol, pol_img, (uint32_t)(tile0 - pol_img), pol_as, (int)row, min(nwaves * run, N - t0), eN + (size_t)t0, lane);
#ifdef PRC_TRACE
            __builtin_amdgcn_s_waitcnt(0x0F70); // (the actions are out)
#endif
            PRC_STAMP(12);
        }
    }
#ifdef PRC_TRACE
    wave_lds_sync();
    {
        const uint32_t wv = (uint32_t)blockIdx.x * nwaves + (uint32_t)wave;
        if (wv < PRC_TRACE_MAX_WAVES && lane < PRC_TRACE_SLOTS) g_prc_trace[wv * PRC_TRACE_SLOTS + lane] = prc_tr[lane];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// cell-meta maintenance (not on the hot path)
// ---------------------------------------------------------------------------------------------------
// Re-base the stamps before the observation counter can reach META_NEVER: every explored cell gets explored
// stamp 0, every presence stamp 0, the counter restarts at 1 (16 381 observations apart).
__global__ void k_meta_rebase(const KP p)
{
    const size_t n = (size_t)p.E * p.W * p.H;
    uint32_t *m = reinterpret_cast<uint32_t *>(p.s.food) + 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = m[i * p.fs];
        m[i * p.fs] = (v & (META_WALL | META_AREA)) | ((v >> META_STAMP_SHIFT) != META_NEVER ? 0u : META_NEVER << META_STAMP_SHIFT);
    }
}

// ===================================================================================
// host-side launchers (called from antsrl_capi.hip)
// ===================================================================================
static int prc_layout(const KP &p)
{
    static const int def[7] = {ANTSRL_CH_ANTS, ANTSRL_CH_PHERO, ANTSRL_CH_PHERO, ANTSRL_CH_ANTHILL,
                               ANTSRL_CH_WALLS, ANTSRL_CH_FOOD, ANTSRL_CH_ROCKS};
    if (p.C != 2 || (p.K != 6 && p.K != 7)) return 0;
    for (int k = 0; k < p.K; ++k)
        if (p.ch_kind[k] != def[k]) return 0;
    if (p.ch_arg[1] != 0 || p.ch_arg[2] != 1) return 0;
    return p.K == 6 ? PLAYOUT_DEFAULT : PLAYOUT_DEFAULT_ROCKS;
}

// Which configurations take the cell-meta path: the reference's perception shapes (the generator's channel
// order with 2 pheromone channels, a perception of at most 64 cells whose two-row group leaves in three
// 16-byte stores per lane) with at most 4096 ants per env (k_move's hash and exchange scratch are LDS-resident).
bool antsrl_meta_supported(const KP &p)
{
    const uint32_t row = (uint32_t)p.PP * p.K;
    // (row >= 128 elements: every copy-out then has at least two whole 128-byte lines in either observation format)
    return prc_layout(p) != 0 && p.PP <= 64 && row >= 128 && row <= 368 && p.N <= 4096 &&
           move_lds_bytes(p.HT, p.N) <= 160 * 1024;
}

static int n_cus()
{
    static int n[ANTSRL_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return 256;
    if (!n[dev] && hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n[dev] = 256;
    return n[dev] > 0 ? n[dev] : 256;
}

// Ants per wave of k_perceive: long runs amortise the prologue and keep a wave's rows one sequential write
// stream; short runs fill the chip (16 waves per CU) several times over so that the tail is short.
static int pick_run(const KP &p)
{
    if (const char *s = PROF_ENV("ANTSRL_PRC_RUN")) return atoi(s);
    const long total = (long)p.E * p.N, slots = (long)n_cus() * 16;
    // c3 (1024 x 512 ants), k_perceive ms early / late in the episode: run 2: 0.313 / 0.308, 4: 0.253 / 0.247,
    // 8: 0.236 / 0.231, 16: 0.237 (late), 32: 0.250 / 0.245, 64: 0.250 (late) — with short runs the segments of one
    // environment run close together in time and share its cell records in L2 (profiles/r02/prc_run_sweep.txt)
    int run = 8;
    while (run > 2 && total / run < slots) run >>= 1; // tiny batches: at least one wave per slot
    return run;
}

int antsrl_perceive_run(const KP &p) { return pick_run(p); }

hipError_t antsrl_launch_move(const KP &p, const int8_t *rot, const int8_t *ph, uint8_t *done, int do_step,
                              uint32_t seq, hipStream_t st)
{
    const int T = p.N <= 1024 ? (p.N + 63) / 64 * 64 : 1024;
    const size_t lds = move_lds_bytes(p.HT, p.N);
    static size_t seen[ANTSRL_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice;
    if (lds > 64 * 1024 && lds > seen[dev]) {
        hipError_t err = hipFuncSetAttribute((const void *)k_move<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        seen[dev] = lds;
    }
    hipLaunchKernelGGL((k_move<2>), dim3(p.E), dim3(T), lds, st, p, rot, ph, done, do_step, seq);
    return hipGetLastError();
}

// The deferred update of the previous step + this step's move in one launch (see k_update_move).
bool antsrl_update_move_supported(const KP &p)
{
    // (scaled units or an explicit sweep alike: with a sweep the host enqueues k_update_move AHEAD of the step's sweep, whose
    //  input buffer the deferred deposit lands in — antsrl_step_update in antsrl_capi.hip)
    return p.meta && p.C == 2 && p.N <= 1024 && !PROF_ENV("ANTSRL_NO_DEFER_UPDATE");
}

hipError_t antsrl_launch_update_move(const KP &p, int out_buf, double g_dep, double inv_g_dep, const int8_t *rot,
                                     const int8_t *ph, uint8_t *done, uint32_t seq, hipStream_t st, bool with_frames)
{
    const int T = (p.N + 63) / 64 * 64;
    size_t lds = align_up(std::max(update_one_lds_bytes(p.HT, p.R, T / 64, p.N), move_lds_bytes(p.HT, p.N)), 16) +
                 16 * (size_t)T; // + cos / sin(theta + pi/2) per ant, parked between the update and the move (UmFwd::frm_off)
    if (const char *s = PROF_ENV("ANTSRL_UM_LDS_PAD")) lds += (size_t)atoi(s) * 1024; // occupancy knob (profiling build)
    static size_t seen[ANTSRL_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice;
    const bool ilv = p.ps == 4; // {p0, p1, food, META} records
    if (lds > 64 * 1024 && lds > seen[dev]) {
        hipError_t err = hipFuncSetAttribute((const void *)k_update_move<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err == hipSuccess)
            err = hipFuncSetAttribute((const void *)k_update_move<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        seen[dev] = lds;
    }
    const int wf = with_frames ? 1 : 0;
    if (ilv) hipLaunchKernelGGL((k_update_move<2, true>), dim3(p.E), dim3(T), lds, st, p, out_buf, g_dep, inv_g_dep, rot, ph, done, seq, wf);
    else hipLaunchKernelGGL((k_update_move<2, false>), dim3(p.E), dim3(T), lds, st, p, out_buf, g_dep, inv_g_dep, rot, ph, done, seq, wf);
    return hipGetLastError();
}

template <int LAYOUT, bool OBS16, bool ILV, bool HAS_OBS, bool POLICY = false, bool PAD = false>
static hipError_t launch_perceive_t(const KP &p, const float *cells, float *obs, float *agent_state, float *reward,
                                    int flags, uint32_t seq, hipStream_t st, const PolArgs &pol = PolArgs{}, uint32_t pitch = 0)
{
    const int run = pick_run(p), nwaves = PRC_TPB / 64;
    if (run < 1 || run > 64) return hipErrorInvalidValue;
    if (POLICY && run * nwaves > 32) return hipErrorInvalidValue; // one MFMA tile of 32 ants per workgroup
    const int nseg = (p.N + run * nwaves - 1) / (run * nwaves);
    const PrcOff lo = prc_offsets(run, p.PP, p.K, p.R, nwaves, POLICY, PAD ? pitch : 0u);
    const size_t pad = PROF_ENV("ANTSRL_PRC_LDS_PAD") ? (size_t)atoi(PROF_ENV("ANTSRL_PRC_LDS_PAD")) * 1024 : 0; // occupancy knob
    size_t lds = lo.total + pad;
    if (POLICY) lds = align_up(lo.total, 16) + 2 * (size_t)prc_policy_img_elems(p.PP * p.K) + 4 * 64 + pad;
    if (lds > 64 * 1024) return hipErrorInvalidValue; // (never with the shapes antsrl_meta_supported admits)
    hipLaunchKernelGGL((k_perceive<LAYOUT, OBS16, ILV, HAS_OBS, POLICY, PAD>), dim3((unsigned)((size_t)p.E * nseg)), dim3(PRC_TPB), lds, st, p,
                       cells, obs, agent_state, reward, flags, seq, run, nseg, pol, pitch);
    return hipGetLastError();
}

// the in-loop policy needs one tile of at most 32 ants per k_perceive workgroup
bool antsrl_inloop_policy_supported(const KP &p) { return p.meta && pick_run(p) * (PRC_TPB / 64) <= 32; }

// obs_pitch: elements between two ants' observation rows (antsrl_set_obs_row_stride), 0 = dense
hipError_t antsrl_launch_perceive(const KP &p, int cur, float *obs, float *agent_state, float *reward, int flags,
                                  uint32_t seq, hipStream_t st, const PolArgs *pol, uint32_t obs_pitch)
{
    const int layout = prc_layout(p);
    const bool o16 = (flags & ACT_OBS_BF16) != 0, ilv = p.ps == 4 && p.fs == 4;
    const float *cells = p.s.phero[cur];
    const bool with_pol = pol && pol->pack && o16; // (the in-loop policy reads bf16 rows: antsrl_set_inloop_policy)
    const bool padded = obs && obs_pitch != 0 && obs_pitch != (uint32_t)(p.PP * p.K);
    if (padded && with_pol) return hipErrorNotSupported; // (the tile image the net reads is dense)
#define PRC_GO(LY)                                                                                             \
    {                                                                                                          \
        if (padded && o16) return launch_perceive_t<LY, true, ILVV, true, false, true>(p, cells, obs, agent_state, reward, flags, seq, st, PolArgs{}, obs_pitch); \
        if (padded) return launch_perceive_t<LY, false, ILVV, true, false, true>(p, cells, obs, agent_state, reward, flags, seq, st, PolArgs{}, obs_pitch); \
        if (with_pol && !obs) return launch_perceive_t<LY, true, ILVV, false, true>(p, cells, obs, agent_state, reward, flags, seq, st, *pol); \
        if (!obs) return launch_perceive_t<LY, false, ILVV, false>(p, cells, obs, agent_state, reward, flags, seq, st); \
        if (with_pol) return launch_perceive_t<LY, true, ILVV, true, true>(p, cells, obs, agent_state, reward, flags, seq, st, *pol); \
        if (o16) return launch_perceive_t<LY, true, ILVV, true>(p, cells, obs, agent_state, reward, flags, seq, st);    \
        return launch_perceive_t<LY, false, ILVV, true>(p, cells, obs, agent_state, reward, flags, seq, st);            \
    }
#define PRC_GO2(LY) { if (ilv) { constexpr bool ILVV = true; PRC_GO(LY) } else { constexpr bool ILVV = false; PRC_GO(LY) } }
    if (layout == PLAYOUT_DEFAULT) PRC_GO2(PLAYOUT_DEFAULT)
    if (layout == PLAYOUT_DEFAULT_ROCKS) PRC_GO2(PLAYOUT_DEFAULT_ROCKS)
#undef PRC_GO2
#undef PRC_GO
    return hipErrorInvalidValue;
}

hipError_t antsrl_launch_meta_rebase(const KP &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_meta_rebase, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p);
    return hipGetLastError();
}
