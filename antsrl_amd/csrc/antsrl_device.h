// antsrl_device.h — device-side state layout and kernel parameter block (gfx950 only).
//
// Data layout in HBM (one batch of E environments, everything env-major):
//   ants        struct-of-arrays, float64 kinematics: x[E][N], y[E][N], theta[E][N],
//               prev_x, prev_y (Ants.ants / Ants.prev_ants, environment/ants.py:27,30);
//               holding f32, mandibles u8, seed f32, activation f32 [E][N][C]
//   pheromone   f32, channels INTERLEAVED per cell: phero[buf][E][W][H][C], two buffers
//               (ping-pong: the decay/diffuse sweep reads `cur`, writes `cur^1`, then the
//               deposit lands in `cur^1`).  One 8-byte gather returns both channels of a cell.
//   food        f32 [E][W][H]
//               With scaled pheromone units and two channels (the reference's setup: no per-step
//               sweep) the two grids are ONE array of 16-byte cell records {p0, p1, food, pad}:
//               a perception is then a single 16-byte gather per cell (KP::ps, KP::fs).
//   walls / anthill area / explored map   bit-packed, 1 bit per cell: u32 [E][ceil(W*H/32)],
//               bit index = x*H + y.  8 KiB per env at 256x256, so whole maps fit in LDS.
//   rocks       float64 SoA [E][R]
//
// Cell-meta layout (KP::meta, the reference's perception shapes: antsrl_perceive.hip).  Every cell owns a
// 32-bit META word right behind its food value — {p0, p1, food, META} in the interleaved record,
// {food, META} records (fs = 2) otherwise — so the gather that fetches a perceived cell's food also
// returns its wall / anthill / presence bits and its explored stamp:
//   bits 0..15   PRESENCE STAMP: sequence number of the last observation in which an ant stood on the cell
//                (RL_api.py:137-142): k_move stores the current number with one 2-byte store per ant, the ants
//                channel is (stamp == current number).  No clearing pass, no read-modify-write, no atomics.
//   bit 16       wall (Walls.map)          bit 17  anthill area (Anthill.area)
//   bits 18..31  EXPLORED STAMP: sequence number of the first observation whose perception covered the cell,
//                META_NEVER if none.  Observation number s counts a cell as unexplored iff stamp >= s (all
//                ants count against the map as it was BEFORE this observation, reward_custom.py:19-22) and
//                marks it by storing the upper half-word (s << 2 | wall / area bits) — a plain 2-byte store:
//                every writer of one observation stores the same value.
//   Sequence numbers run 1 .. META_NEVER - 2; the host then re-bases both stamps in one pass over the cells
//   (explored cells -> stamp 0, presence -> 0) and restarts at 1: every 16 381 observations.
#pragma once
#include <stdint.h>
#include "../../include/antsrl.h"

#define META_PRES_MASK 0xFFFFu
#define META_WALL 0x10000u
#define META_AREA 0x20000u
#define META_STAMP_SHIFT 18
#define META_NEVER 0x3FFFu

// PROF_ENV("NAME"): A/B and ablation switches exist in the profiling build only (-DANTSRL_PROFILING ->
// libantsrl_hip_prof.so, see antsrl_amd/build.py); in the product library no environment variable can
// change what a step computes.
#ifdef ANTSRL_PROFILING
#define PROF_ENV(name) getenv(name)
#else
#define PROF_ENV(name) ((const char *)nullptr)
#endif
// Compile-time ablations (bit masks, results WRONG by design: antsrl_perceive.hip, antsrl_update_one.h) and the wave
// time-line trace of k_perceive exist in variant builds of the profiling library only — a product build with any of them
// set does not compile.
#ifndef PRC_ABL
#define PRC_ABL 0
#endif
#ifndef UM_ABL
#define UM_ABL 0
#endif
#if !defined(ANTSRL_PROFILING) && (PRC_ABL != 0 || UM_ABL != 0 || defined(PRC_TRACE) || defined(UM_TRACE))
#error "PRC_ABL / UM_ABL / PRC_TRACE / UM_TRACE are profiling switches: build them with -DANTSRL_PROFILING (python -m antsrl_amd.build --variant NAME -D...)"
#endif

// -DUM_TRACE (variant build; profiles/um_trace.py): the time line of every workgroup of k_update_move's last launch — its
// thread 0 writes s_memrealtime (10 ns ticks) into g_um_trace[block][slot] at the phase boundaries.
#ifdef UM_TRACE
#define UM_TRACE_SLOTS 16
#define UM_TRACE_MAX_WGS 4096
static __device__ uint32_t g_um_trace[UM_TRACE_SLOTS * UM_TRACE_MAX_WGS]; // (one copy per translation unit; antsrl_perceive.hip's is the one read back)
#define UM_STAMP(slot)                                                                                             \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < UM_TRACE_MAX_WGS) g_um_trace[blockIdx.x * UM_TRACE_SLOTS + (slot)] = (uint32_t)wall_clock64(); \
    } while (0)
// (the stamp waits until `v` — a value that depends on the loads of interest — is in a register)
#define UM_STAMP_ON(slot, v)                                                                                       \
    do {                                                                                                           \
        asm volatile("" ::"v"(v));                                                                                 \
        UM_STAMP(slot);                                                                                            \
    } while (0)
#else
#define UM_STAMP(slot) do { } while (0)
#define UM_STAMP_ON(slot, v) do { } while (0)
#endif

struct __align__(16) AntFrame { double cx, cy, ct, st; }; // perception centre, cos/sin(theta + pi/2)
// k_update_move: what the update hands to the move of the same ant in registers — x, y after the update (= the new `prev`)
// and theta; the food value and META word of the cell the ant stands on (the deposit cell's record IS the record of the
// move's mandible decision: one 16-byte load instead of a pheromone read, a dependent food read and an area-bit read);
// and the move's own per-ant inputs, fetched in front of the update so that their round trip rides with the update's.
// Round 4 (profiles/history/r04/um_trace.txt: the workgroup's life is a chain of exposed latencies, not bytes): the move's rotation and
// its float64 sincos — ~400 VALU instructions that need theta and the action only — are evaluated by the UPDATE while its
// record load is in flight (`pre`); the env's timestep and reward_primed flag travel along, so that thread 0 of the move
// has no memory round trip of its own in front of the move's first barrier.
#define UMFWD_WIN 0x100
struct UmFwd {
    double x, y, th;
    double th_new, sn, cs; // pre: theta after the rotation, its sine and cosine (RL_api.py:190-196)
    float food, hold;
    uint32_t meta;
    int m, rot, pa;
    int rec;      // 1: food / meta are valid (interleaved records)
    int pre;      // 1: th_new / sn / cs are valid
    int has_rot;  // the step rotates (a rotation tensor was passed)
    int ts;       // the env's timestep after the update
    // m, bit 8 (UMFWD_WIN): this ant is the last-writer-wins winner of its cell in the update's hash (the highest ant index
    // standing on the cell).  The move's food exchange resolves duplicates over the SAME cells (prev := cur in the update) by
    // the same rule: the verdict is forwarded and the move builds no table of its own (round 5; it shares the mandible
    // flag's register — one more live VGPR spilled in the explicit-sweep variant of the kernel)
    uint8_t primed; // reward_primed[e] as it stood at the launch
    uint32_t frm_off; // LDS byte offset of the [T][2] doubles where the update parks cos / sin(theta + pi/2) for the frame
};

struct DState {
    double *x, *y, *theta, *prev_x, *prev_y; // [E*N]
    double *prev_dist;                       // [E*N]  All_Rewards.previous_dist
    float *holding, *seed, *prev_holding;    // [E*N]
    float *activation;                       // [E*N*C]
    uint8_t *mandibles, *reward_state;       // [E*N]
    int32_t *dirty_cell;                     // [E*N]  food cell on the anthill area this ant
                                             //        rewrote in the last step, else -1
    int32_t *walldep_cell;                   // [E*N]  (scaled mode) wall cell this ant deposited
                                             //        on in the last update, else -1
    float *phero[2];                         // [E*G*ps] each: value of channel c of cell g at [g*ps + c]
    float *food;                             // [E*G*fs]:  food of cell g at [g*fs]
                                             // (KP::ps, KP::fs: cell strides in floats.  Separate arrays:
                                             //  ps = C, fs = 1.  Interleaved record, scaled units with two
                                             //  channels: ONE array of {p0, p1, food, pad} per cell, ps = fs
                                             //  = 4, food = phero[0] + 2 — a perception is then a single
                                             //  16-byte gather per cell)
    uint32_t *walls_bits, *area_bits, *explored_bits; // [E*words]
    uint32_t *big_pres, *big_old; // [E*words] presence / pre-step explored map of k_act when the grid's bit maps do
                                  // not fit LDS (antsrl_act_needs_hbm_maps), else NULL
    int32_t *anthill_xyr;                    // [E*3]
    double *anthill_food;                    // [E]
    double *rock_cx, *rock_cy, *rock_r, *rock_w; // [E*R]
    int32_t *timestep;                       // [E]
    uint8_t *reward_primed;                  // [E]
    int32_t *gen_discs;                      // [E][ANTSRL_MAX_FOOD_DISCS][3] food discs of the device generator
    int32_t *gen_perlin;                     // [E][2] PerlinGenerator's offsets (ANTSRL_RNG_REFERENCE: drawn from the
                                             //        environment's Python stream by k_gen_mt)
    // cell-meta layout only (KP::meta)
    uint8_t *primed_cur;                     // [E]   reward_primed as the current observation must see it
    AntFrame *frames;                        // [E*N] the perception frames of the ants as k_update_move left them (ACT_FRAMES):
                                             //       valid for the k_perceive launch that follows it, and for nothing else
    unsigned char *pol_pack;                 // ANTSRL_POL_PACK_BYTES: the in-loop policy's weights as MFMA fragments (antsrl_policy.hip)
};

// Kernel parameter block, passed by value (lives in the kernarg segment; every field is
// wave-uniform so the compiler reads it with scalar loads).
// In-loop policy (antsrl_set_inloop_policy): the reference's linear DQN net evaluated by k_perceive on the rows it has just
// written (bf16), per 32-ant tile.  All pointers device memory; the packs live in DState::pol_pack (k_policy_pack).
#define ANTSRL_POL_MAX_KSTEPS 64
#define ANTSRL_POL_WPACK_BYTES (ANTSRL_POL_MAX_KSTEPS * 64 * 16)      // [ks][64 lanes] bf16x8: W1's A fragments
#define ANTSRL_POL_A2PACK_BYTES (2 * 64 * 16)                          // [2][64 lanes] bf16x8: the heads' A fragments
#define ANTSRL_POL_LANEPACK_FLOATS (48 * 64 + 8)                       // bias1 / was0 / was1 [16][64] each, then hb[6]
#define ANTSRL_POL_PACK_BYTES (ANTSRL_POL_WPACK_BYTES + ANTSRL_POL_A2PACK_BYTES + 4 * ANTSRL_POL_LANEPACK_FLOATS)
struct PolArgs {
    const unsigned char *pack; // DState::pol_pack, or NULL: no in-loop policy
    int8_t *rot, *ph;          // [E][N] next actions (ph may be NULL)
    int ks;                    // ceil(F / 16)
    int _pad;
};

struct KP {
    DState s;
    int32_t E, N, W, H, C, R, K, P, PP, r, words;
    int32_t HT;          // last-writer-wins hash table size (pow2 >= 2N)
    int32_t has_mask, has_max_val, reward_kind, max_time, filter_radius, explore_on;
    int32_t ch_kind[ANTSRL_MAX_CHANNELS], ch_arg[ANTSRL_MAX_CHANNELS];
    // The mandible walk over perceived_objects (RL_api.py:178-185) reduced on the host: Food SETS the bit (q > 0), Anthill
    // CLEARS it (on the area); both are idempotent, so the walk equals its last two distinct operations in order
    // (S C S = C S, C S C = S C).  1 = set, 2 = clear, 0 = none: no per-channel scalar load in the ants' loop.
    int32_t mand_first, mand_last;
    uint8_t mask[ANTSRL_MAX_PCELLS + 7];
    double delta, fwd_delta, max_speed, max_rot_speed, carry, backward, max_hold;
    double max_val, deposit_strength, threshold, reward_threshold;
    double fct_explore, fct_food, fct_anthill, fct_explore_holding, fct_heading;
    double filter[ANTSRL_MAX_FILTER_TAPS];
    uint64_t rng_seed;
    uint32_t env_id_base, _pad0; // AntsCfg.env_id_base: environment e of this handle is GLOBAL environment env_id_base + e (every env-keyed random stream)
    // Scaled pheromone representation (ANTSRL_PHERO_AUTO with a centre-only filter): the grid
    // holds u = v / f0^S_at_write; v_now = u * g_now with g = f0^S.  g == 1 in explicit mode.
    int32_t scaled, _pad;
    int32_t ps, fs;   // cell strides of the pheromone / food arrays, see DState
    int32_t meta;     // 1: cell-meta layout (META word at food + 1), k_move + k_perceive instead of k_act
    int32_t tiled;    // 1: the cell records are stored in BLOCKS OF 2 x 4 CELLS per 128-byte line (rec_xy below) instead
                      //    of row-major (1 x 8 cells per line): a 7x7 perception patch then touches ~18 lines instead of
                      //    ~21 and an ant's old and new cell share a line more often.  Interleaved 16-byte records on the
                      //    cell-meta path only, W even, H a multiple of 4; the bit maps and every canonical (caller-facing)
                      //    layout stay row-major.
    int32_t ftile;    // 1: the 8-byte {food, META} records of the explicit-sweep layout (fs == 2, cell-meta path) are stored
                      //    in BLOCKS OF 4 x 4 CELLS per 128-byte line (frec_xy below); the pheromone buffers of that layout
                      //    stay row-major (the marching stencils stream them).  W and H multiples of 4.
    int32_t _pad3;
    double g_now;     // f0^S          : materialises values for the perception gather / read-out
    double g_dep;     // f0^(S+1)      : at deposit time, after the conceptual sweep of this update
    double inv_g_dep; // 1 / g_dep
    // DIFFUSE_FILTER taps split into fp32 hi + lo (hi + lo == the float64 tap to ~1e-15): the
    // stencil runs in fp32 FMAs without the systematic per-step bias a rounded tap would add
    // layout [b][a]{hi,lo} (b = tap column, a = tap row) so one column's taps are contiguous
    float ftap[2 * ANTSRL_MAX_FILTER_TAPS];
    // Rank-1 filters F[a][b] = u[a] * v[b] (a Gaussian, the reference's 3x3 when DIFFUSE_FACTOR = 0 ...):
    // {hi, lo} pairs of u (tap row index a: the marching direction) and v (tap column index b: across lanes)
    int32_t filter_sep, _pad2;
    float fsep_u[2 * (2 * ANTSRL_MAX_FILTER_RADIUS + 1)], fsep_v[2 * (2 * ANTSRL_MAX_FILTER_RADIUS + 1)];
};

// k_act flags
#define ACT_STEP 1        // run RLApi.step's action phases before observing
#define ACT_HAS_OBS 2     // obs pointer valid
#define ACT_FUSED_UPDATE 4 // run Environment.update of the same step at the tail of the launch
#define ACT_OBS_BF16 8     // `obs` is a bfloat16 tensor (antsrl_set_obs_format): same values, rounded to nearest even
#define ACT_FRAMES 16      // k_perceive: DState::frames holds this observation's frames (written by the k_update_move in front of it)
// profiling ablations (env ANTSRL_ABLATE, results are WRONG with any of them set; bench/tests never set it)
#define ACT_ABL_NO_ITEMS 256   // skip the perception phase
#define ACT_ABL_NO_GATHER 512  // no pheromone/food gathers
#define ACT_ABL_NO_STORE 1024  // no observation stores
#define ACT_ABL_NO_EXPLORE 2048 // no explored-map test/mark
#define ACT_ABL_TRACE 32768      // (results stay valid) record the per-workgroup phase timeline, see act_trace
