// antsrl_capi.hip — the C-ABI of libantsrl_hip.so (include/antsrl.h).
//
// Host-side only: validates the configuration, carves the caller's device workspace into the
// state arrays of antsrl_device.h, and enqueues the kernels of antsrl_act / _update / _sweep / _state.hip on the
// caller's stream.  No allocation, no synchronisation, no exceptions across the ABI.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "antsrl_device.h"

// launchers (antsrl_act.hip, antsrl_update.hip, antsrl_sweep.hip, antsrl_state.hip)
hipError_t antsrl_launch_act(const KP &p, const int8_t *rot, const int8_t *ph, int cur, float *obs,
                             float *agent_state, float *reward, uint8_t *done, int flags,
                             const double *jitter, int out_buf, hipStream_t st);
bool antsrl_act_fits(const KP &p);
bool antsrl_act_needs_hbm_maps(const KP &p);
hipError_t antsrl_launch_sweep(const KP &p, int cur, hipStream_t st);
hipError_t antsrl_launch_update(const KP &p, const double *jitter, int out_buf, hipStream_t st, int phases = 15);
hipError_t antsrl_launch_collect_full(const KP &p, hipStream_t st);
hipError_t antsrl_launch_reset(const KP &p, const AntsInit *in, hipStream_t st);
hipError_t antsrl_launch_set_activation(const KP &p, const float *act, hipStream_t st);
hipError_t antsrl_launch_read_state(const KP &p, int which, int cur, void *dst, hipStream_t st);
hipError_t antsrl_launch_perceptive_field(const KP &p, uint8_t *dst, hipStream_t st);
hipError_t antsrl_launch_generate(const KP &p, const AntsGen &g, uint64_t seed, hipStream_t st);
hipError_t antsrl_launch_phero_wall_clear(const KP &p, hipStream_t st, int buf = 0);
hipError_t antsrl_launch_phero_renorm(const KP &p, hipStream_t st);
// cell-meta path (antsrl_perceive.hip)
bool antsrl_meta_supported(const KP &p);
hipError_t antsrl_launch_move(const KP &p, const int8_t *rot, const int8_t *ph, uint8_t *done, int do_step, uint32_t seq,
                              hipStream_t st);
hipError_t antsrl_launch_perceive(const KP &p, int cur, float *obs, float *agent_state, float *reward, int flags,
                                  uint32_t seq, hipStream_t st, const PolArgs *pol, uint32_t obs_pitch);
bool antsrl_inloop_policy_supported(const KP &p);
hipError_t antsrl_launch_policy_pack(unsigned char *pack, const float *w1, const float *b1, const float *w2, const float *b2,
                                     const float *w3, const float *b3, int F, hipStream_t st);
hipError_t antsrl_launch_meta_rebase(const KP &p, hipStream_t st);
bool antsrl_update_move_supported(const KP &p);
hipError_t antsrl_launch_update_move(const KP &p, int out_buf, double g_dep, double inv_g_dep, const int8_t *rot,
                                     const int8_t *ph, uint8_t *done, uint32_t seq, hipStream_t st, bool with_frames);
int antsrl_perceive_run(const KP &p);
hipError_t antsrl_launch_copy16(void *dst, const void *src, size_t bytes, hipStream_t st);

struct AntsHandle {
    AntsCfg cfg;
    KP p;
    int cur;               // pheromone buffer holding the current grid
    int steps_since_update;// RLApi.step calls since the last Environment.update
    bool need_full_collect;// next update must run Anthill.update over the whole grid
    bool is_reset;
    bool episode_over;     // a refused auto-reset left a finished episode behind (see antsrl_step_update)
    size_t ws_bytes;
    AntsGen gen;           // device generator parameters (antsrl_generate)
    bool has_gen;
    uint64_t episode_seed; // seed of the current episode
    int host_timestep;     // mirrors Environment.timestep (all envs step in lockstep)
    long long sweeps;      // scaled mode: updates since the units were last re-based
    bool obs_bf16;         // observation buffers are bfloat16 (antsrl_set_obs_format)
    uint32_t obs_pitch;    // elements between two ants' observation rows (antsrl_set_obs_row_stride), 0 = dense
    bool need_wall_clear;  // scaled mode: initial grid may hold pheromone on wall cells
    hipEvent_t ev[ANTSRL_TIMING_EVENTS]; // measurement hook (antsrl_set_timing_events)
    bool ev_armed;
    uint32_t obs_seq;      // cell-meta path: observations since the explored stamps were last re-based
    PolArgs pol;           // in-loop policy (antsrl_set_inloop_policy); pol.pack == NULL: none
    // Deferred update (include/antsrl.h): Environment.update was requested, its host-side bookkeeping is done, its
    // kernel has NOT been enqueued yet — the next antsrl_step runs it fused with the move (k_update_move); every other
    // entry point that touches the state enqueues it first (flush_pending).
    bool pend_update;
    KP pend_p;             // kernel parameters as they stood at the update's call (g_dep / inv_g_dep differ afterwards)
    int pend_out_buf;
    int phase_next;        // antsrl_update_phase: the next phase of an Environment.update in progress (0: none in progress)
};

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char *what)
{
    return fail(ANTSRL_E_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

extern "C" int antsrl_abi_version(void) { return ANTSRL_ABI_VERSION; }
extern "C" size_t antsrl_cfg_size(void) { return sizeof(AntsCfg); }
extern "C" const char *antsrl_last_error(void) { return g_err; }

static void fill_kp(const AntsCfg *c, KP *p);

static int validate(const AntsCfg *c)
{
    if (!c) return fail(ANTSRL_E_INVALID, "cfg is NULL");
    if (c->abi_version != ANTSRL_ABI_VERSION)
        return fail(ANTSRL_E_INVALID, "AntsCfg.abi_version %d != library %d", c->abi_version, ANTSRL_ABI_VERSION);
    if (c->n_envs < 1 || c->n_ants < 1 || c->w < 1 || c->h < 1)
        return fail(ANTSRL_E_INVALID, "n_envs, n_ants, w, h must be >= 1");
    if ((long long)c->w * c->h > (1ll << 30)) return fail(ANTSRL_E_INVALID, "grid too large");
    if (c->n_envs > 65535) // the sweep / stencil kernels index envs with blockIdx.y
        return fail(ANTSRL_E_INVALID, "n_envs must be <= 65535 per handle (shard larger batches over handles / GPUs)");
    if (c->n_phero < 1 || c->n_phero > ANTSRL_MAX_PHERO)
        return fail(ANTSRL_E_INVALID, "n_phero must be in 1..%d", ANTSRL_MAX_PHERO);
    if (c->n_rocks < 0 || c->n_rocks > 32) return fail(ANTSRL_E_INVALID, "n_rocks must be in 0..32");
    if (c->perception_radius < 0 || 2 * c->perception_radius + 1 > ANTSRL_MAX_PSIDE)
        return fail(ANTSRL_E_INVALID, "perception side must be <= %d", ANTSRL_MAX_PSIDE);
    if (c->n_channels < 1 || c->n_channels > ANTSRL_MAX_CHANNELS)
        return fail(ANTSRL_E_INVALID, "n_channels must be in 1..%d", ANTSRL_MAX_CHANNELS);
    for (int k = 0; k < c->n_channels; ++k) {
        const int kind = c->channel_kind[k];
        if (kind < ANTSRL_CH_ANTS || kind > ANTSRL_CH_ROCKS) return fail(ANTSRL_E_INVALID, "bad channel kind");
        if (kind == ANTSRL_CH_PHERO && (c->channel_arg[k] < 0 || c->channel_arg[k] >= c->n_phero))
            return fail(ANTSRL_E_INVALID, "pheromone channel index out of range");
        if (kind == ANTSRL_CH_PHERO && !c->has_max_val) // phero / None raises TypeError, RL_api.py:125
            return fail(ANTSRL_E_INVALID, "a perceived pheromone needs max_val (RL_api.py:125 divides by it)");
        if (kind == ANTSRL_CH_ROCKS && c->n_rocks == 0)
            return fail(ANTSRL_E_INVALID, "rocks perceived but n_rocks == 0");
    }
    if (c->filter_radius < 0 || c->filter_radius > ANTSRL_MAX_FILTER_RADIUS)
        return fail(ANTSRL_E_INVALID, "filter_radius must be in 0..%d", ANTSRL_MAX_FILTER_RADIUS);
    if (c->reward_kind < ANTSRL_REWARD_NONE || c->reward_kind > ANTSRL_REWARD_ALL)
        return fail(ANTSRL_E_INVALID, "bad reward_kind");
    if (c->has_max_val && !(c->phero_max_val > 0)) return fail(ANTSRL_E_INVALID, "phero_max_val must be > 0");
    if (c->act_path < ANTSRL_ACT_AUTO || c->act_path > ANTSRL_ACT_SINGLE_KERNEL) return fail(ANTSRL_E_INVALID, "bad act_path");
    if (c->env_id_base < 0 || (long long)c->env_id_base + c->n_envs > 0x7fffffffll)
        return fail(ANTSRL_E_INVALID, "env_id_base must be >= 0 and env_id_base + n_envs must fit 31 bits");
    if (c->n_envs_total != 0 && (long long)c->n_envs_total < (long long)c->env_id_base + c->n_envs)
        return fail(ANTSRL_E_INVALID, "n_envs_total %d < env_id_base + n_envs = %lld (0 = that sum)", c->n_envs_total,
                    (long long)c->env_id_base + c->n_envs);
    if (c->act_path == ANTSRL_ACT_CELL_META) {
        KP kp;
        fill_kp(c, &kp);
        if (!kp.meta)
            return fail(ANTSRL_E_UNSUPPORTED, "ANTSRL_ACT_CELL_META needs 2 pheromone channels, the generator's channel order "
                                              "([Ants, Phero0, Phero1, Anthill, Walls, Food(, Rocks)]), a perception of 128..368 values "
                                              "per ant over at most 64 cells, and at most 4096 ants per env");
    }
    return ANTSRL_OK;
}

// Scaled pheromone units (see ANTSRL_PHERO_AUTO in antsrl.h) apply to a centre-only filter whose
// coefficient is a genuine decay; anything else takes the explicit sweep.
static bool use_scaled(const AntsCfg *c)
{
    return c->phero_mode == ANTSRL_PHERO_AUTO && c->filter_radius == 0 && c->filter[0] > 0.5 &&
           c->filter[0] <= 1.0 && c->phero_threshold >= 0.0;
}

// Interleaved cell record {p0, p1, food, pad}: with scaled units there is no per-step sweep whose traffic the
// wider record would inflate, and a perception becomes ONE 16-byte gather per cell instead of an 8-byte
// and a 4-byte one (profiles/history/obs_write_probe.hip: 48-57 cycles per ant per CU against 70-77 on L2 hits,
// 180-197 against 275 from HBM).  ANTSRL_NO_INTERLEAVE keeps the separate arrays (A/B).
static bool use_interleaved(const AntsCfg *c)
{
    static const bool off = PROF_ENV("ANTSRL_NO_INTERLEAVE") != nullptr;
    return !off && use_scaled(c) && c->n_phero == 2;
}

// Carves the workspace; with base == NULL only computes the size.
static size_t carve(const AntsCfg *c, DState *s, unsigned char *base)
{
    const size_t E = c->n_envs, N = c->n_ants, G = (size_t)c->w * c->h, Cn = c->n_phero, R = c->n_rocks;
    const size_t words = (G + 31) / 32;
    size_t off = 0;
    auto take = [&](size_t bytes) -> unsigned char * {
        unsigned char *ptr = base ? base + off : nullptr;
        off = (off + bytes + 255) / 256 * 256;
        return ptr;
    };
    KP kp;
    fill_kp(c, &kp);
    DState d;
    d.x = (double *)take(8 * E * N); d.y = (double *)take(8 * E * N); d.theta = (double *)take(8 * E * N);
    d.prev_x = (double *)take(8 * E * N); d.prev_y = (double *)take(8 * E * N);
    d.prev_dist = (double *)take(8 * E * N);
    d.holding = (float *)take(4 * E * N); d.seed = (float *)take(4 * E * N);
    d.prev_holding = (float *)take(4 * E * N);
    d.activation = (float *)take(4 * E * N * Cn);
    d.mandibles = (uint8_t *)take(E * N); d.reward_state = (uint8_t *)take(E * N);
    d.dirty_cell = (int32_t *)take(4 * E * N);
    d.walldep_cell = (int32_t *)take(4 * E * N);
    if (use_interleaved(c)) { // one {p0, p1, food, pad} record per cell
        d.phero[0] = d.phero[1] = (float *)take(16 * E * G);
        d.food = d.phero[0] ? d.phero[0] + 2 : nullptr;
    } else {
        d.phero[0] = (float *)take(4 * E * G * Cn);
        d.phero[1] = use_scaled(c) ? d.phero[0] : (float *)take(4 * E * G * Cn); // no ping-pong when scaled
        d.food = (float *)take(4 * E * G * (size_t)kp.fs); // {food, META} records on the cell-meta path
    }
    d.walls_bits = (uint32_t *)take(4 * E * words); d.area_bits = (uint32_t *)take(4 * E * words);
    d.explored_bits = (uint32_t *)take(4 * E * words);
    d.big_pres = d.big_old = nullptr;
    if (!kp.meta && antsrl_act_needs_hbm_maps(kp)) {
        d.big_pres = (uint32_t *)take(4 * E * words);
        d.big_old = (uint32_t *)take(4 * E * words);
    }
    d.primed_cur = kp.meta ? (uint8_t *)take(E) : nullptr;
    d.frames = kp.meta ? (AntFrame *)take(sizeof(AntFrame) * E * N) : nullptr;
    d.anthill_xyr = (int32_t *)take(4 * E * 3);
    d.anthill_food = (double *)take(8 * E);
    d.rock_cx = (double *)take(8 * E * (R ? R : 1)); d.rock_cy = (double *)take(8 * E * (R ? R : 1));
    d.rock_r = (double *)take(8 * E * (R ? R : 1)); d.rock_w = (double *)take(8 * E * (R ? R : 1));
    d.timestep = (int32_t *)take(4 * E);
    d.reward_primed = (uint8_t *)take(E);
    d.gen_discs = (int32_t *)take(4 * E * ANTSRL_MAX_FOOD_DISCS * 3);
    d.gen_perlin = (int32_t *)take(4 * E * 2);
    d.pol_pack = (unsigned char *)take(ANTSRL_POL_PACK_BYTES);
    if (s) *s = d;
    return off;
}

static void fill_kp(const AntsCfg *c, KP *p)
{
    memset(p, 0, sizeof(*p));
    p->E = c->n_envs; p->N = c->n_ants; p->W = c->w; p->H = c->h; p->C = c->n_phero; p->R = c->n_rocks;
    p->K = c->n_channels; p->r = c->perception_radius; p->P = 2 * p->r + 1; p->PP = p->P * p->P;
    p->words = (int)(((size_t)c->w * c->h + 31) / 32);
    int ht = 64;
    while (ht < 2 * c->n_ants) ht <<= 1;
    p->HT = ht;
    p->has_mask = c->has_mask; p->has_max_val = c->has_max_val; p->reward_kind = c->reward_kind;
    p->max_time = c->max_time; p->filter_radius = c->filter_radius;
    p->explore_on = c->reward_kind == ANTSRL_REWARD_EXPLORATION ||
                    (c->reward_kind == ANTSRL_REWARD_ALL &&
                     (c->fct_explore != 0.0 || c->fct_explore_holding != 0.0)); // reward_custom.py:87
    for (int k = 0; k < ANTSRL_MAX_CHANNELS; ++k) {
        p->ch_kind[k] = c->channel_kind[k];
        p->ch_arg[k] = c->channel_arg[k];
    }
    p->mand_first = p->mand_last = 0;
    for (int k = 0; k < c->n_channels; ++k) {
        const int op = c->channel_kind[k] == ANTSRL_CH_FOOD ? 1 : c->channel_kind[k] == ANTSRL_CH_ANTHILL ? 2 : 0;
        if (op && op != p->mand_last) {
            p->mand_first = p->mand_last;
            p->mand_last = op;
        }
    }
    memcpy(p->mask, c->mask, ANTSRL_MAX_PCELLS);
    p->delta = c->delta; p->fwd_delta = c->fwd_delta; p->max_speed = c->max_speed;
    p->max_rot_speed = c->max_rot_speed; p->carry = c->carry_speed_reduction;
    p->backward = c->backward_speed_reduction; p->max_hold = c->max_hold;
    p->max_val = c->phero_max_val; p->deposit_strength = c->deposit_strength;
    p->threshold = c->phero_threshold; p->reward_threshold = c->reward_threshold;
    p->fct_explore = c->fct_explore; p->fct_food = c->fct_food; p->fct_anthill = c->fct_anthill;
    p->fct_explore_holding = c->fct_explore_holding; p->fct_heading = c->fct_headinganthill;
    memcpy(p->filter, c->filter, sizeof(p->filter));
    {
        const int S = 2 * c->filter_radius + 1;
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                const double f = c->filter[a * S + b];
                const float hi = (float)f;
                p->ftap[(b * S + a) * 2 + 0] = hi;
                p->ftap[(b * S + a) * 2 + 1] = (float)(f - (double)hi);
            }
    }
    {
        // Rank-1 test: F == u v^T / F[i0][j0] around the largest tap, to 4 ulp of the largest tap.  Then the
        // stencil runs as a pass across the lanes followed by a pass along the march (2 x S taps instead of
        // S x S).  The taps u[a] * v[b] differ from F[a][b] by ~1e-16 relative: far inside the fp32 grid's
        // rounding, and unbiased like the hi + lo split of the full filter.
        const int S = 2 * c->filter_radius + 1;
        p->filter_sep = 0;
        if (S > 1 && !PROF_ENV("ANTSRL_NO_SEPARABLE")) {
            int i0 = 0, j0 = 0;
            double big = 0.0;
            for (int a = 0; a < S; ++a)
                for (int b = 0; b < S; ++b)
                    if (fabs(c->filter[a * S + b]) > big) { big = fabs(c->filter[a * S + b]); i0 = a; j0 = b; }
            bool sep = big > 0.0;
            double u[2 * ANTSRL_MAX_FILTER_RADIUS + 1], v[2 * ANTSRL_MAX_FILTER_RADIUS + 1];
            for (int a = 0; a < S && sep; ++a) u[a] = c->filter[a * S + j0];
            for (int b = 0; b < S && sep; ++b) v[b] = c->filter[i0 * S + b] / c->filter[i0 * S + j0];
            for (int a = 0; a < S && sep; ++a)
                for (int b = 0; b < S; ++b)
                    if (fabs(u[a] * v[b] - c->filter[a * S + b]) > 1e-15 * big) sep = false;
            if (sep) {
                p->filter_sep = 1;
                for (int k = 0; k < S; ++k) {
                    const float uh = (float)u[k], vh = (float)v[k];
                    p->fsep_u[2 * k] = uh; p->fsep_u[2 * k + 1] = (float)(u[k] - (double)uh);
                    p->fsep_v[2 * k] = vh; p->fsep_v[2 * k + 1] = (float)(v[k] - (double)vh);
                }
            }
        }
    }
    p->rng_seed = c->rng_seed;
    p->env_id_base = (uint32_t)c->env_id_base;
    p->scaled = use_scaled(c) ? 1 : 0;
    p->ps = use_interleaved(c) ? 4 : c->n_phero;
    p->fs = use_interleaved(c) ? 4 : 1;
    p->g_now = p->g_dep = p->inv_g_dep = 1.0;
    // cell-meta path (k_move + k_perceive) for the reference's perception shapes (AntsCfg.act_path pins either).
    // (Round 2 first kept k_act for batches under 8192 ants — one launch fewer; with the deferred update the cell-meta
    // path is two launches per step as well and wins at every size: c1 13.8 against 15.3 us/step, 16 envs x 256 ants
    // 16.2 against 30.1 — profiles/r02/tiny_ab.txt.)
    p->meta = antsrl_meta_supported(*p) && c->act_path != ANTSRL_ACT_SINGLE_KERNEL ? 1 : 0;
    if (p->meta && p->fs == 1) p->fs = 2;
    // 2 x 4-cell blocks per line for the interleaved records (antsrl_device.h: KP::tiled); ANTSRL_NO_TILED: A/B (profiling build)
    p->tiled = (p->meta && p->ps == 4 && p->fs == 4 && (c->w & 1) == 0 && (c->h & 3) == 0 && !PROF_ENV("ANTSRL_NO_TILED")) ? 1 : 0;
    // 4 x 4-cell blocks per line for the 8-byte {food, META} records beside separate pheromone buffers (c4's layout: KP::ftile)
    p->ftile = (p->meta && p->fs == 2 && (c->w & 3) == 0 && (c->h & 3) == 0 && !PROF_ENV("ANTSRL_NO_TILED")) ? 1 : 0;
}

// f0^S for the next observation, f0^(S+1) for the next deposit
static void set_decay(AntsHandle *h)
{
    if (!h->p.scaled) return;
    const double f0 = h->cfg.filter[0];
    h->p.g_now = pow(f0, (double)h->sweeps);
    h->p.g_dep = pow(f0, (double)(h->sweeps + 1));
    h->p.inv_g_dep = 1.0 / h->p.g_dep;
}

extern "C" int antsrl_workspace_bytes(const AntsCfg *cfg, size_t *bytes)
{
    int rc = validate(cfg);
    if (rc) return rc;
    if (!bytes) return fail(ANTSRL_E_INVALID, "bytes is NULL");
    *bytes = carve(cfg, nullptr, nullptr);
    return ANTSRL_OK;
}

extern "C" int antsrl_create(const AntsCfg *cfg, void *workspace, size_t workspace_bytes, AntsHandle **out)
{
    int rc = validate(cfg);
    if (rc) return rc;
    if (!out) return fail(ANTSRL_E_INVALID, "out is NULL");
    if (!workspace) return fail(ANTSRL_E_INVALID, "workspace is NULL");
    if (((uintptr_t)workspace & 255) != 0) return fail(ANTSRL_E_INVALID, "workspace must be 256-byte aligned");
    const size_t need = carve(cfg, nullptr, nullptr);
    if (workspace_bytes < need)
        return fail(ANTSRL_E_NOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    AntsHandle *h = new (std::nothrow) AntsHandle();
    if (!h) return fail(ANTSRL_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    fill_kp(cfg, &h->p);
    carve(cfg, &h->p.s, (unsigned char *)workspace);
    h->cur = 0; h->steps_since_update = 0; h->need_full_collect = true; h->is_reset = false; h->episode_over = false; h->pend_update = false;
    h->phase_next = 0;
    h->pol = PolArgs{};
    h->sweeps = 0; h->need_wall_clear = false;
    h->has_gen = false; h->episode_seed = 0; h->host_timestep = 1; h->obs_bf16 = false; h->obs_pitch = 0;
    h->ws_bytes = need;
    h->ev_armed = false;
    h->obs_seq = 0;
    if (!h->p.meta && !antsrl_act_fits(h->p)) {
        delete h;
        return fail(ANTSRL_E_UNSUPPORTED,
                    "%d ants with a %dx%d perception need more than 160 KiB of LDS per workgroup "
                    "(per-ant frames, the collision hash and the row staging are kept in LDS)", cfg->n_ants,
                    2 * cfg->perception_radius + 1, 2 * cfg->perception_radius + 1);
    }
    *out = h;
    return ANTSRL_OK;
}

extern "C" void antsrl_destroy(AntsHandle *h) { delete h; }

#ifdef ANTSRL_PROFILING
// Zone probe (profiles/r05/split_workspace_probe.py): the interleaved cell records at a caller-supplied address instead of
// inside the workspace — call right after antsrl_create, before antsrl_reset / antsrl_generate.  16 * E * W * H bytes.
extern "C" int antsrl_debug_set_cells_base(AntsHandle *h, void *cells)
{
    if (!h || !cells || h->p.ps != 4 || h->p.fs != 4) return ANTSRL_E_INVALID;
    h->p.s.phero[0] = h->p.s.phero[1] = (float *)cells;
    h->p.s.food = (float *)cells + 2;
    return ANTSRL_OK;
}
#endif

extern "C" int antsrl_reset(AntsHandle *h, const AntsInit *init, void *stream)
{
    if (!h || !init) return fail(ANTSRL_E_INVALID, "NULL handle or init");
    if (!init->ants_xyt || !init->seed || !init->walls || !init->food || !init->anthill_xyr)
        return fail(ANTSRL_E_INVALID, "AntsInit: ants_xyt, seed, walls, food, anthill_xyr are required");
    if (h->p.R > 0 && !init->rocks) return fail(ANTSRL_E_INVALID, "AntsInit.rocks is NULL but n_rocks > 0");
    h->pend_update = false; // (a deferred update of the state being replaced)
    h->phase_next = 0;
    hipError_t e = antsrl_launch_reset(h->p, init, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "reset");
    h->p.deposit_strength = h->cfg.deposit_strength;
    h->cur = 0; h->steps_since_update = 0; h->need_full_collect = true; h->is_reset = true; h->episode_over = false;
    h->sweeps = 0; h->need_wall_clear = h->p.scaled && init->phero != nullptr;
    h->host_timestep = 1; h->obs_seq = 0;
    set_decay(h);
    return ANTSRL_OK;
}

// np.random.seed(seed * 5) takes 32 bits (environment_generator.py:55): env e draws from seed + env_id_base + e.  Checked at
// every (auto-)reset: numpy raises past 2^32, a wrapped seed would silently replay an earlier episode.  (Written so that
// no intermediate can wrap in uint64.)
static int check_gen_seed(const AntsHandle *h, const AntsGen &g, uint64_t seed)
{
    const uint64_t top = (uint64_t)h->p.env_id_base + (uint64_t)h->p.E; // < 2^31
    const uint64_t lim = 0xFFFFFFFFull / 5u;                            // (top may exceed it: compare before subtracting)
    if (g.rng_kind == ANTSRL_RNG_REFERENCE && (top > lim || seed > lim - top))
        return fail(ANTSRL_E_INVALID, "ANTSRL_RNG_REFERENCE: (episode_seed + env_id_base + n_envs) * 5 must stay below 2^32 (np.random.seed)");
    return ANTSRL_OK;
}

static int do_generate(AntsHandle *h, uint64_t seed, hipStream_t st)
{
    int rc = check_gen_seed(h, h->gen, seed);
    if (rc) return rc;
    h->pend_update = false; // (a deferred update of the state being replaced)
    h->phase_next = 0;
    hipError_t e = antsrl_launch_generate(h->p, h->gen, seed, st);
    if (e != hipSuccess) return hip_fail(e, "generate");
    h->p.deposit_strength = h->cfg.deposit_strength;
    h->cur = 0; h->steps_since_update = 0; h->need_full_collect = true; h->is_reset = true; h->episode_over = false;
    h->sweeps = 0; h->need_wall_clear = false; h->host_timestep = 1; h->obs_seq = 0;
    h->episode_seed = seed;
    set_decay(h);
    return ANTSRL_OK;
}

extern "C" int antsrl_generate(AntsHandle *h, const AntsGen *gen, uint64_t episode_seed, void *stream)
{
    if (!h || !gen) return fail(ANTSRL_E_INVALID, "NULL handle or gen");
    if (gen->n_food_discs < 0 || gen->n_food_discs > ANTSRL_MAX_FOOD_DISCS)
        return fail(ANTSRL_E_INVALID, "n_food_discs must be in 0..%d", ANTSRL_MAX_FOOD_DISCS);
    if (gen->food_rmin < 0 || gen->food_rmax < gen->food_rmin) return fail(ANTSRL_E_INVALID, "bad food radii");
    if (gen->wall_kind == ANTSRL_WALLS_BERNOULLI) {
        if (!(gen->wall_density >= 0.0 && gen->wall_density <= 1.0)) return fail(ANTSRL_E_INVALID, "bad wall_density");
    } else if (gen->wall_kind == ANTSRL_WALLS_PERLIN) {
        if (!(gen->wall_density >= -1.0 && gen->wall_density <= 1.0) || gen->perlin_octaves < 1 || gen->perlin_octaves > 8 ||
            !(gen->perlin_scale > 0.0) || !(gen->perlin_persistence > 0.0) || !(gen->perlin_lacunarity > 0.0))
            return fail(ANTSRL_E_INVALID, "bad Perlin wall parameters (density in [-1, 1], 1..8 octaves, positive "
                                          "scale / persistence / lacunarity)");
    } else if (gen->wall_kind == ANTSRL_WALLS_INPUT) {
        if (!gen->walls_input) return fail(ANTSRL_E_INVALID, "ANTSRL_WALLS_INPUT needs AntsGen.walls_input");
    } else {
        return fail(ANTSRL_E_INVALID, "bad wall_kind %d", gen->wall_kind);
    }
    if (gen->rng_kind != ANTSRL_RNG_COUNTER && gen->rng_kind != ANTSRL_RNG_REFERENCE)
        return fail(ANTSRL_E_INVALID, "bad rng_kind %d", gen->rng_kind);
    if (gen->rng_kind == ANTSRL_RNG_REFERENCE) {
        if (gen->wall_kind == ANTSRL_WALLS_BERNOULLI && gen->wall_density != 0.0)
            return fail(ANTSRL_E_UNSUPPORTED, "the reference has no Bernoulli walls generator: with ANTSRL_RNG_REFERENCE use "
                                              "ANTSRL_WALLS_PERLIN, ANTSRL_WALLS_INPUT or wall_density 0");
        // (the 32-bit limit of np.random.seed(seed * 5) is checked in do_generate, for this and every auto-reset episode)
    }
    int rc = check_gen_seed(h, *gen, episode_seed); // (before anything of the handle changes: a refused call leaves it as it was)
    if (rc) return rc;
    h->gen = *gen;
    h->has_gen = true;
    return do_generate(h, episode_seed, (hipStream_t)stream);
}

// (only k_act can refuse the format: the cell-meta path takes bfloat16 observations at any grid size)
static int bf16_unsupported()
{
    return fail(ANTSRL_E_UNSUPPORTED, "bfloat16 observations on the single-kernel path (k_act) need 2 pheromone channels, the "
                                      "generator's channel order ([Ants, Phero0, Phero1, Anthill, Walls, Food(, Rocks)]), a perception of "
                                      "at most 64 cells and a grid whose bit maps fit LDS (up to ~600k cells); the cell-meta path "
                                      "(ANTSRL_Q_CELL_META) has no grid limit");
}

extern "C" int antsrl_set_obs_format(AntsHandle *h, int format)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (format != ANTSRL_OBS_F32 && format != ANTSRL_OBS_BF16) return fail(ANTSRL_E_INVALID, "bad observation format %d", format);
    if ((format == ANTSRL_OBS_BF16) != h->obs_bf16) h->obs_pitch = 0; // (a row stride is a whole number of lines of ONE element size)
    h->obs_bf16 = format == ANTSRL_OBS_BF16;
    if (!h->obs_bf16) h->pol = PolArgs{}; // (the in-loop policy reads bfloat16 rows)
    return ANTSRL_OK;
}

extern "C" int antsrl_set_obs_row_stride(AntsHandle *h, int32_t stride_elems)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    const int32_t row = h->p.PP * h->p.K, esz = h->obs_bf16 ? 2 : 4;
    if (stride_elems == 0 || stride_elems == row) {
        h->obs_pitch = 0;
        return ANTSRL_OK;
    }
    if (stride_elems < row || ((long long)stride_elems * esz) % 128 != 0 || stride_elems > row + 128 / esz)
        return fail(ANTSRL_E_INVALID, "obs row stride %d: 0 (dense) or the row's %d elements rounded up to whole 128-byte lines (%d)",
                    stride_elems, row, (row * esz + 127) / 128 * 128 / esz);
    if (!h->p.meta) return fail(ANTSRL_E_UNSUPPORTED, "antsrl_set_obs_row_stride needs the cell-meta path (ANTSRL_Q_CELL_META)");
    if (h->pol.pack) // (refused HERE, not in the middle of a step whose move has already been enqueued)
        return fail(ANTSRL_E_UNSUPPORTED, "a padded observation row stride and the in-loop policy exclude each other (the net reads a "
                                          "dense tile image): switch the policy off first (antsrl_set_inloop_policy with w1 == NULL)");
    h->obs_pitch = (uint32_t)stride_elems;
    return ANTSRL_OK;
}

static int hip_fail(hipError_t e, const char *what);
// Enqueues a deferred update on its own (k_update_one), e.g. ahead of a state read.
static int flush_pending(AntsHandle *h, hipStream_t st)
{
    if (!h->pend_update) return ANTSRL_OK;
    h->pend_update = false;
    hipError_t e = antsrl_launch_update(h->pend_p, nullptr, h->pend_out_buf, st);
    if (e != hipSuccess) return hip_fail(e, "deferred update");
    return ANTSRL_OK;
}

static int not_reset(const AntsHandle *h)
{
    if (h->episode_over)
        return fail(ANTSRL_E_INVALID, "the episode is over and its auto-reset was refused (episode seed past np.random.seed's 32 bits): "
                                      "call antsrl_reset or antsrl_generate");
    return fail(ANTSRL_E_INVALID, "antsrl_reset has not been called on this handle");
}

static int mid_update(const AntsHandle *h)
{
    return fail(ANTSRL_E_INVALID, "an Environment.update is in progress (antsrl_update_phase %d of 4 is next): finish it first", h->phase_next);
}

// One observation on the cell-meta path: k_move (with the action phases when `stepping`) then k_perceive.
static int meta_observe(AntsHandle *h, const int8_t *rot, const int8_t *ph, float *obs, float *agent_state,
                        float *reward, uint8_t *done, bool stepping, hipStream_t st, bool timed)
{
    hipError_t e;
    if (obs && h->obs_pitch != 0 && h->obs_pitch != (uint32_t)(h->p.PP * h->p.K) && h->pol.pack && h->obs_bf16) // (both setters refuse the
        return fail(ANTSRL_E_UNSUPPORTED, "a padded observation row stride and the in-loop policy exclude each other"); // pair; before any launch)
    if (h->obs_seq >= META_NEVER - 2) { // explored stamps: re-base long before the counter can reach "never"
        e = antsrl_launch_meta_rebase(h->p, st);
        if (e != hipSuccess) return hip_fail(e, "explored-stamp rebase");
        h->obs_seq = 0;
    }
    h->obs_seq++;
    bool frames = false; // k_update_move leaves the ants' perception frames for the k_perceive right behind it
    if (h->pend_update && stepping) { // the previous step's update and this step's move in one launch
        h->pend_update = false;
        // Frames from k_update_move pay where k_perceive is latency-bound — the in-loop policy's launches, whose rows are
        // bfloat16 or stay in LDS: c5 -2.5 %, k_perceive -5 % — and cost 1-2 us of a second sincos where it is bound by its
        // float32 store stream, which hides the prologue anyway (c3 / c2 / c4: +1 %; profiles/r04/frames_ab.txt).
        // ANTSRL_FRAMES=0 / 1 (profiling library): never / always.
        frames = h->pol.pack != nullptr;
        if (const char *s = PROF_ENV("ANTSRL_FRAMES")) frames = atoi(s) != 0;
        e = antsrl_launch_update_move(h->p, h->pend_out_buf, h->pend_p.g_dep, h->pend_p.inv_g_dep, rot, ph, done, h->obs_seq, st, frames);
        if (e != hipSuccess) return hip_fail(e, "update + move");
    } else {
        int rc = flush_pending(h, st);
        if (rc) return rc;
        e = antsrl_launch_move(h->p, rot, ph, done, stepping ? 1 : 0, h->obs_seq, st);
        if (e != hipSuccess) return hip_fail(e, "move");
    }
    if (timed) (void)hipEventRecord(h->ev[2], st);
    e = antsrl_launch_perceive(h->p, h->cur, obs, agent_state, reward,
                               (stepping ? ACT_STEP : 0) | (obs ? ACT_HAS_OBS : 0) | (frames ? ACT_FRAMES : 0) |
                                   ((obs || h->pol.pack) && h->obs_bf16 ? ACT_OBS_BF16 : 0), // (act-only: rows in LDS, bf16)
                               h->obs_seq, st, &h->pol, h->obs_pitch);
    if (e == hipErrorNotSupported)
        return fail(ANTSRL_E_UNSUPPORTED, "a padded observation row stride and the in-loop policy exclude each other (the net reads a dense tile image)");
    if (e != hipSuccess) return hip_fail(e, "perceive");
    return ANTSRL_OK;
}

static int do_step(AntsHandle *h, const int8_t *rot, const int8_t *ph, float *obs, float *agent_state,
                   float *reward, uint8_t *done, hipStream_t st, bool fused_update = false,
                   const double *jitter = nullptr, bool timed = false)
{
    if (ph && h->p.C != 2) // Ants.activate_pheromone hard-codes two channels, ants.py:89-96
        return fail(ANTSRL_E_INVALID, "pheromone actions need exactly 2 pheromone channels (ants.py:89-96)");
    if (h->steps_since_update > 0) h->need_full_collect = true; // dirty-cell list would be overwritten
    if (h->p.meta) {
        int rc = meta_observe(h, rot, ph, obs, agent_state, reward, done, true, st, timed);
        if (rc) return rc;
        h->steps_since_update++;
        return ANTSRL_OK;
    }
    if (obs && h->obs_pitch) return fail(ANTSRL_E_UNSUPPORTED, "antsrl_set_obs_row_stride needs the cell-meta path (ANTSRL_Q_CELL_META)");
    if (timed) (void)hipEventRecord(h->ev[2], st);
    static const int ablate = PROF_ENV("ANTSRL_ABLATE") ? atoi(PROF_ENV("ANTSRL_ABLATE")) & ~15 : 0; // profiling build only
    hipError_t e = antsrl_launch_act(h->p, rot, ph, h->cur, obs, agent_state, reward, done,
                                     ACT_STEP | (obs ? ACT_HAS_OBS : 0) | (fused_update ? ACT_FUSED_UPDATE : 0) | ablate |
                                         (obs && h->obs_bf16 ? ACT_OBS_BF16 : 0),
                                     jitter, h->p.scaled ? 0 : h->cur ^ 1, st);
    if (e == hipErrorNotSupported) return bf16_unsupported();
    if (e != hipSuccess) return hip_fail(e, "step");
    h->steps_since_update++;
    return ANTSRL_OK;
}

static int do_update(AntsHandle *h, const double *jitter, hipStream_t st, bool sweep_done,
                     bool update_fused = false, bool may_defer = false)
{
    hipError_t e;
    int frc = flush_pending(h, st); // (two updates in a row)
    if (frc) return frc;
    if (h->p.scaled) {
        if (h->p.g_dep < 1e-20) { // re-base the units long before u = v / f0^S can overflow fp32
            e = antsrl_launch_phero_renorm(h->p, st);
            if (e != hipSuccess) return hip_fail(e, "pheromone renorm");
            h->sweeps = 0;
            set_decay(h);
        }
        if (h->need_wall_clear) { // Walls pass of this update (walls.py:30) on the initial grid,
                                  // before this update's deposits land
            e = antsrl_launch_phero_wall_clear(h->p, st);
            if (e != hipSuccess) return hip_fail(e, "pheromone wall clear");
            h->need_wall_clear = false;
        }
    } else if (!sweep_done) {
        e = antsrl_launch_sweep(h->p, h->cur, st);
        if (e != hipSuccess) return hip_fail(e, "pheromone sweep");
    }
    if (!update_fused) {
        // Deferred: with the library's own wall jitter (no caller buffer to outlive the call) and nothing that has to
        // run after the update kernel in this call, the kernel is left for the next step's k_update_move.
        if (may_defer && !jitter && !h->need_full_collect && antsrl_update_move_supported(h->p)) {
            h->pend_update = true;
            h->pend_p = h->p;
            h->pend_out_buf = h->p.scaled ? 0 : h->cur ^ 1;
        } else {
            e = antsrl_launch_update(h->p, jitter, h->p.scaled ? 0 : h->cur ^ 1, st);
            if (e != hipSuccess) return hip_fail(e, "update");
        }
    }
    if (h->p.scaled) {
        h->sweeps++;
        set_decay(h);
    }
    if (h->need_full_collect) {
        e = antsrl_launch_collect_full(h->p, st);
        if (e != hipSuccess) return hip_fail(e, "anthill collect");
        h->need_full_collect = false;
    }
    if (!h->p.scaled) h->cur ^= 1;
    h->steps_since_update = 0;
    h->host_timestep++;
    return ANTSRL_OK;
}

extern "C" int antsrl_step(AntsHandle *h, const int8_t *rotation, const int8_t *phero, float *obs,
                           float *agent_state, float *reward, uint8_t *done, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!h->is_reset) return not_reset(h);
    if (h->phase_next) return mid_update(h);
    if (!agent_state || !reward || !done) return fail(ANTSRL_E_INVALID, "agent_state, reward, done are required");
    return do_step(h, rotation, phero, obs, agent_state, reward, done, (hipStream_t)stream);
}

extern "C" int antsrl_observe(AntsHandle *h, float *obs, float *agent_state, float *reward, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!h->is_reset) return not_reset(h);
    if (h->phase_next) return mid_update(h);
    if (h->p.meta)
        return meta_observe(h, nullptr, nullptr, obs, agent_state, reward, nullptr, false, (hipStream_t)stream, false);
    if (obs && h->obs_pitch) return fail(ANTSRL_E_UNSUPPORTED, "antsrl_set_obs_row_stride needs the cell-meta path (ANTSRL_Q_CELL_META)");
    hipError_t e = antsrl_launch_act(h->p, nullptr, nullptr, h->cur, obs, agent_state, reward, nullptr,
                                     obs ? ACT_HAS_OBS | (h->obs_bf16 ? ACT_OBS_BF16 : 0) : 0, nullptr, 0,
                                     (hipStream_t)stream);
    if (e == hipErrorNotSupported) return bf16_unsupported();
    if (e != hipSuccess) return hip_fail(e, "observe");
    return ANTSRL_OK;
}

extern "C" int antsrl_update(AntsHandle *h, const double *wall_jitter, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!h->is_reset) return not_reset(h);
    if (h->phase_next) return mid_update(h);
    return do_update(h, wall_jitter, (hipStream_t)stream, false, false, true);
}

// Environment.update one reference step at a time (include/antsrl.h).  Bit-identical to antsrl_update: the same device
// functions in the same order, cut at the launch boundaries.
extern "C" int antsrl_update_phase(AntsHandle *h, int phase, const double *wall_jitter, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!h->is_reset) return not_reset(h);
    if (phase < ANTSRL_PHASE_WALLS || phase > ANTSRL_PHASE_ANTHILL) return fail(ANTSRL_E_INVALID, "bad phase %d", phase);
    if (phase != h->phase_next)
        return fail(ANTSRL_E_INVALID, "antsrl_update_phase: phase %d is next, got %d (WALLS, ROCKS_PHEROMONE, ANTS, ANTHILL in order)", h->phase_next, phase);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (phase == ANTSRL_PHASE_WALLS) { // walls.py:22-30 (update_step -1)
        int frc = flush_pending(h, st);
        if (frc) return frc;
        if (h->p.scaled) {
            if (h->p.g_dep < 1e-20) { // re-base the units long before u = v / f0^S can overflow fp32 (value-preserving)
                e = antsrl_launch_phero_renorm(h->p, st);
                if (e != hipSuccess) return hip_fail(e, "pheromone renorm");
                h->sweeps = 0;
                set_decay(h);
            }
            if (h->need_wall_clear) {
                e = antsrl_launch_phero_wall_clear(h->p, st);
                if (e != hipSuccess) return hip_fail(e, "pheromone wall clear");
                h->need_wall_clear = false;
            }
        } else { // `phero[map] = 0` as a step of its own (the sweep of the next phase does it again, to the same effect)
            e = antsrl_launch_phero_wall_clear(h->p, st, h->cur);
            if (e != hipSuccess) return hip_fail(e, "pheromone wall clear");
        }
        e = antsrl_launch_update(h->p, wall_jitter, h->p.scaled ? 0 : h->cur, st, 1 /* UPD_WALLS */);
        if (e != hipSuccess) return hip_fail(e, "update: walls");
    } else if (phase == ANTSRL_PHASE_ROCKS_PHEROMONE) { // circle_obstacles.py:32-58, pheromone.py:43-45 (update_step 0)
        e = antsrl_launch_update(h->p, nullptr, h->p.scaled ? 0 : h->cur, st, 2 /* UPD_ROCKS */);
        if (e != hipSuccess) return hip_fail(e, "update: rocks");
        if (h->p.scaled) { // the conceptual sweep: readers materialise one more decay from here on; the deposit of the
            h->sweeps++;   // ANTS phase lands "after the sweep of this update" = at the new g_now
            set_decay(h);
        } else {
            e = antsrl_launch_sweep(h->p, h->cur, st);
            if (e != hipSuccess) return hip_fail(e, "pheromone sweep");
            h->cur ^= 1; // readers (and the deposit) see the swept grid from here on
        }
    } else if (phase == ANTSRL_PHASE_ANTS) { // ants.py:123-130 (update_step 999)
        KP kp = h->p;
        if (kp.scaled) { // (the units were advanced in the previous phase: the deposit factor is the CURRENT decay)
            kp.g_dep = kp.g_now;
            kp.inv_g_dep = 1.0 / kp.g_dep;
        }
        e = antsrl_launch_update(kp, nullptr, kp.scaled ? 0 : h->cur, st, 4 /* UPD_ANTS */);
        if (e != hipSuccess) return hip_fail(e, "update: ants");
    } else { // anthill.py:41-46 (update_step 1000); the environment's timestep advances here
        e = antsrl_launch_update(h->p, nullptr, h->p.scaled ? 0 : h->cur, st, 8 /* UPD_ANTHILL */);
        if (e != hipSuccess) return hip_fail(e, "update: anthill");
        if (h->need_full_collect) {
            e = antsrl_launch_collect_full(h->p, st);
            if (e != hipSuccess) return hip_fail(e, "anthill collect");
            h->need_full_collect = false;
        }
        h->steps_since_update = 0;
        h->host_timestep++;
    }
    h->phase_next = (phase + 1) % 4;
    return ANTSRL_OK;
}

extern "C" int antsrl_flush(AntsHandle *h, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    return flush_pending(h, (hipStream_t)stream);
}

extern "C" int antsrl_step_update(AntsHandle *h, const int8_t *rotation, const int8_t *phero,
                                  const double *wall_jitter, float *obs, float *agent_state, float *reward,
                                  uint8_t *done, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!h->is_reset) return not_reset(h);
    if (h->phase_next) return mid_update(h);
    if (!agent_state || !reward || !done) return fail(ANTSRL_E_INVALID, "agent_state, reward, done are required");
    hipStream_t st = (hipStream_t)stream;
    const bool timed = h->ev_armed;
    h->ev_armed = false;
    if (timed) (void)hipEventRecord(h->ev[0], st);
    // The sweep only reads phero[cur] and the wall bitmap, so it can be enqueued first: the
    // perception gather of the step reads the same (pre-update) buffer.  Except behind a DEFERRED update: its deposits land
    // in phero[cur] — this sweep's input — with k_update_move, so the sweep follows the step's kernels (the perception
    // reads phero[cur], too, and nothing of the step reads the sweep's output).
    const bool sweep_late = !h->p.scaled && h->pend_update;
    if (!h->p.scaled && !sweep_late) {
        hipError_t e = antsrl_launch_sweep(h->p, h->cur, st);
        if (e != hipSuccess) return hip_fail(e, "pheromone sweep");
    }
    if (timed) (void)hipEventRecord(h->ev[1], st);
    // Fuse Environment.update into the same launch unless the one-off wall clear of an initial
    // pheromone grid has to run between this step's observation and this update's deposits.
    // Measured on MI355X (c3): fusing is within 1 % of two launches either way (0.325 vs 0.328 ms/step with
    // the loop form of the update; the separate k_update_one has since become the faster kernel) — with
    // two workgroups per CU the update's latency-bound phases idle half the CU.  Opt-in
    // (ANTSRL_FUSE_UPDATE=1) for re-evaluation.
    static const bool want_fuse = PROF_ENV("ANTSRL_FUSE_UPDATE") && atoi(PROF_ENV("ANTSRL_FUSE_UPDATE")) != 0;
    const bool fuse = want_fuse && !h->p.meta && !(h->p.scaled && h->need_wall_clear);
    if (fuse && h->p.scaled && h->p.g_dep < 1e-20) { // re-base before the launch (value-preserving)
        hipError_t e = antsrl_launch_phero_renorm(h->p, st);
        if (e != hipSuccess) return hip_fail(e, "pheromone renorm");
        h->sweeps = 0;
        set_decay(h);
    }
    int rc = do_step(h, rotation, phero, obs, agent_state, reward, done, st, fuse, wall_jitter, timed);
    if (rc) return rc;
    if (timed) (void)hipEventRecord(h->ev[3], st);
    if (sweep_late) {
        hipError_t e = antsrl_launch_sweep(h->p, h->cur, st);
        if (e != hipSuccess) return hip_fail(e, "pheromone sweep");
    }
    const bool was_done = h->host_timestep == h->cfg.max_time; // RL_api.py:200, same for every env
    const bool regen = was_done && h->has_gen && h->gen.auto_reset;
    rc = do_update(h, wall_jitter, st, true, fuse, !regen);
    if (timed) (void)hipEventRecord(h->ev[4], st);
    if (rc == ANTSRL_OK && regen) {
        // next episode, like main.py:69-79 does per episode (reference streams: global env g takes seed + g, so the next
        // episode starts one whole batch — AntsCfg.n_envs_total, all shards — of seeds further)
        const uint64_t total = h->cfg.n_envs_total ? (uint64_t)h->cfg.n_envs_total : (uint64_t)h->p.env_id_base + (uint64_t)h->p.E;
        rc = do_generate(h, h->episode_seed + (h->gen.rng_kind == ANTSRL_RNG_REFERENCE ? total : 1u), st);
        if (rc != ANTSRL_OK) {
            // The step and the update have run, the next episode could not be drawn (the 32-bit seed limit): the handle
            // holds a finished episode.  Every later step fails loudly until antsrl_reset / antsrl_generate gives it a
            // new one, instead of running past max_time with `done` never firing again.
            h->is_reset = false;
            h->episode_over = true;
        }
    }
    return rc;
}

extern "C" int antsrl_query(const AntsHandle *h, int what, long long *value)
{
    if (!h || !value) return fail(ANTSRL_E_INVALID, "NULL handle or value");
    switch (what) {
    case ANTSRL_Q_CELL_META: *value = h->p.meta; break;
    case ANTSRL_Q_SCALED_UNITS: *value = h->p.scaled; break;
    case ANTSRL_Q_INTERLEAVED: *value = h->p.ps == 4 && h->p.fs == 4; break;
    case ANTSRL_Q_FILTER_SEPARABLE: *value = h->p.filter_sep; break;
    case ANTSRL_Q_PERCEIVE_RUN: *value = h->p.meta ? antsrl_perceive_run(h->p) : 0; break;
    case ANTSRL_Q_TIMESTEP: *value = h->host_timestep; break;
    case ANTSRL_Q_DEFERRED_UPDATE: *value = antsrl_update_move_supported(h->p); break;
    default: return fail(ANTSRL_E_INVALID, "bad query selector %d", what);
    }
    return ANTSRL_OK;
}

extern "C" int antsrl_bench_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!dst || !src || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(ANTSRL_E_INVALID, "bench_copy: 16-byte aligned pointers and a multiple of 16 bytes");
    hipError_t e = antsrl_launch_copy16(dst, src, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "bench_copy");
    return ANTSRL_OK;
}

extern "C" int antsrl_set_timing_events(AntsHandle *h, void *const *events)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    h->ev_armed = events != nullptr;
    if (events)
        for (int i = 0; i < ANTSRL_TIMING_EVENTS; ++i) h->ev[i] = (hipEvent_t)events[i];
    return ANTSRL_OK;
}

extern "C" int antsrl_set_activation(AntsHandle *h, const float *act, double new_deposit_strength, void *stream)
{
    if (!h || !act) return fail(ANTSRL_E_INVALID, "NULL handle or act");
    if (!h->is_reset) return not_reset(h);
    if (h->phase_next) return mid_update(h);
    int frc = flush_pending(h, (hipStream_t)stream); // (the deferred update deposits with the activation as it was)
    if (frc) return frc;
    hipError_t e = antsrl_launch_set_activation(h->p, act, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "set_activation");
    if (new_deposit_strength > 0) h->p.deposit_strength = new_deposit_strength;
    return ANTSRL_OK;
}

hipError_t antsrl_launch_policy(const float *obs, const float *agent_state, const float *w1, const float *b1,
                                const float *w2, const float *b2, const float *w3, const float *b3, int8_t *rot,
                                int8_t *ph, float *logits, int M, int F, hipStream_t st, bool obs_bf16);

extern "C" int antsrl_policy_mlp(AntsHandle *h, const float *obs, const float *agent_state, int64_t n_ants,
                                 int32_t n_features, const float *w1, const float *b1, const float *w2,
                                 const float *b2, const float *w3, const float *b3, int8_t *rotation,
                                 int8_t *pheromone, float *logits, void *stream)
{
    // h may be NULL (float32 observations); a handle supplies the observation format (antsrl_set_obs_format)
    if (!obs || !agent_state || !w1 || !b1 || !w2 || !b2 || !rotation)
        return fail(ANTSRL_E_INVALID, "policy_mlp: obs, agent_state, w1, b1, w2, b2, rotation are required");
    if ((w3 == nullptr) != (b3 == nullptr) || (pheromone && !w3))
        return fail(ANTSRL_E_INVALID, "policy_mlp: w3/b3 go together and are needed for a pheromone output");
    if (n_ants < 1 || n_ants > 0x7fffffff || n_features < 1 || n_features + 2 > 1024)
        return fail(ANTSRL_E_INVALID, "policy_mlp: n_ants >= 1 and 1 <= n_features <= 1022");
    hipError_t e = antsrl_launch_policy(obs, agent_state, w1, b1, w2, b2, w3, b3, rotation, pheromone, logits,
                                        (int)n_ants, n_features, (hipStream_t)stream, h && h->obs_bf16);
    if (e != hipSuccess) return hip_fail(e, "policy_mlp");
    return ANTSRL_OK;
}

extern "C" int antsrl_set_inloop_policy(AntsHandle *h, int32_t n_features, const float *w1, const float *b1, const float *w2,
                                        const float *b2, const float *w3, const float *b3, int8_t *rotation_next,
                                        int8_t *pheromone_next, void *stream)
{
    if (!h) return fail(ANTSRL_E_INVALID, "NULL handle");
    if (!w1) { // switch it off
        h->pol = PolArgs{};
        return ANTSRL_OK;
    }
    if (!b1 || !w2 || !b2 || !rotation_next)
        return fail(ANTSRL_E_INVALID, "inloop_policy: w1, b1, w2, b2, rotation_next are required");
    if ((w3 == nullptr) != (b3 == nullptr) || (pheromone_next && !w3))
        return fail(ANTSRL_E_INVALID, "inloop_policy: w3/b3 go together and are needed for a pheromone output");
    if (n_features != h->p.PP * h->p.K)
        return fail(ANTSRL_E_INVALID, "inloop_policy: n_features must be P*P*K = %d", h->p.PP * h->p.K);
    if (!h->obs_bf16 || !antsrl_inloop_policy_supported(h->p))
        return fail(ANTSRL_E_UNSUPPORTED, "inloop_policy needs the cell-meta path (ANTSRL_Q_CELL_META) with bfloat16 "
                                          "observations (antsrl_set_obs_format) — use antsrl_policy_mlp otherwise");
    if (h->obs_pitch != 0 && h->obs_pitch != (uint32_t)(h->p.PP * h->p.K))
        return fail(ANTSRL_E_UNSUPPORTED, "the in-loop policy and a padded observation row stride exclude each other (the net reads a "
                                          "dense tile image): antsrl_set_obs_row_stride(h, 0) first");
    hipError_t e = antsrl_launch_policy_pack(h->p.s.pol_pack, w1, b1, w2, b2, w3, b3, n_features, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "inloop_policy pack");
    h->pol.pack = h->p.s.pol_pack;
    h->pol.rot = rotation_next;
    h->pol.ph = pheromone_next;
    h->pol.ks = (n_features + 15) / 16;
    return ANTSRL_OK;
}

static size_t state_bytes(const AntsHandle *h, int which)
{
    const size_t E = h->p.E, N = h->p.N, G = (size_t)h->p.W * h->p.H, C = h->p.C, R = h->p.R;
    switch (which) {
    case ANTSRL_S_ANTS_XYT: return 8 * E * N * 3;
    case ANTSRL_S_PREV_XY: return 8 * E * N * 2;
    case ANTSRL_S_HOLDING: case ANTSRL_S_SEED: return 4 * E * N;
    case ANTSRL_S_MANDIBLES: case ANTSRL_S_REWARD_STATE: return E * N;
    case ANTSRL_S_ACTIVATION: return 4 * E * N * C;
    case ANTSRL_S_PHERO: return 4 * E * C * G;
    case ANTSRL_S_FOOD: return 4 * E * G;
    case ANTSRL_S_EXPLORED: case ANTSRL_S_WALLS: case ANTSRL_S_ANTHILL_AREA: return E * G;
    case ANTSRL_S_ANTHILL_FOOD: return 8 * E;
    case ANTSRL_S_ROCK_CENTERS: return 8 * E * R * 2;
    case ANTSRL_S_TIMESTEP: return 4 * E;
    case ANTSRL_S_ANTHILL_XYR: return 4 * E * 3;
    case ANTSRL_S_ROCK_RW: return 8 * E * R * 2;
    case ANTSRL_S_PHERO_C0: case ANTSRL_S_PHERO_C1: case ANTSRL_S_PHERO_C2: case ANTSRL_S_PHERO_C3:
        return (size_t)(which - ANTSRL_S_PHERO_C0) < C ? 4 * E * G : 0;
    default: return 0;
    }
}

extern "C" int antsrl_state_bytes(const AntsHandle *h, int which, size_t *bytes)
{
    if (!h || !bytes) return fail(ANTSRL_E_INVALID, "NULL handle or bytes");
    if (which < 0 || which >= ANTSRL_S_COUNT_) return fail(ANTSRL_E_INVALID, "bad state selector %d", which);
    *bytes = state_bytes(h, which);
    return ANTSRL_OK;
}

extern "C" int antsrl_perceptive_field(AntsHandle *h, uint8_t *dst, void *stream)
{
    if (!h || !dst) return fail(ANTSRL_E_INVALID, "NULL handle or dst");
    if (!h->is_reset) return not_reset(h);
    // (no flush: a DEFERRED update has not touched the positions yet — the field is the last observation's, as in the
    //  reference, where RLApi.observation computes it; behind an update that has run it is the moved ants')
    hipError_t e = antsrl_launch_perceptive_field(h->p, dst, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "perceptive_field");
    return ANTSRL_OK;
}

extern "C" int antsrl_read_state(AntsHandle *h, int which, void *dst, void *stream)
{
    if (!h || !dst) return fail(ANTSRL_E_INVALID, "NULL handle or dst");
    if (!h->is_reset) return not_reset(h);
    if (which < 0 || which >= ANTSRL_S_COUNT_) return fail(ANTSRL_E_INVALID, "bad state selector %d", which);
    if (state_bytes(h, which) == 0) return ANTSRL_OK;
    int frc = flush_pending(h, (hipStream_t)stream);
    if (frc) return frc;
    hipError_t e = antsrl_launch_read_state(h->p, which, h->cur, dst, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "read_state");
    return ANTSRL_OK;
}
