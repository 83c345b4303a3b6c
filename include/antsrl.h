/*
 * antsrl.h — C-ABI of libantsrl_hip.so: the MI355X (gfx950) implementation of the
 * AntsRL environment step loop (RLApi.step + Environment.update), batched over
 * many independent environments.
 *
 * The reference (SelennLamson/AntsRL) is pure Python and has no FFI of its own:
 * the boundary it exposes is the Python surface of `RLApi` / `Environment`
 * (SURVEY.md §8(b)).  Each entry point below names the reference interface it
 * replaces (paths relative to the reference checkout).  The Python shim
 * `antsrl_amd.RLApi` binds these through ctypes, passing `tensor.data_ptr()`
 * of torch-ROCm tensors; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - plain C types only; every buffer is caller-owned DEVICE memory, env-major
 *    and contiguous; the library never allocates, frees or synchronises inside
 *    step/update/observe (it only enqueues kernels on the caller's stream);
 *  - all persistent state lives in ONE caller-provided device workspace whose
 *    size is reported by antsrl_workspace_bytes();
 *  - return value 0 = success, negative = error (see ANTSRL_E_*); nothing is
 *    thrown across the ABI; antsrl_last_error() gives a message for the last
 *    failure on the calling thread;
 *  - a handle belongs to the device its workspace lives on (any number of handles per device), holds up
 *    to 65535 environments and is not thread-safe;
 *  - sizes: any grid; up to 4096 ants per env with the reference's perception shapes (2 pheromone channels,
 *    the generator's channel order, up to 64 perceived cells: the cell-meta path, k_move + k_perceive),
 *    ~2400 with other channel lists (k_act keeps per-ant frames in LDS); antsrl_create() returns
 *    ANTSRL_E_UNSUPPORTED beyond that.
 */
#ifndef ANTSRL_H
#define ANTSRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANTSRL_ABI_VERSION 5

#define ANTSRL_MAX_CHANNELS 16
#define ANTSRL_MAX_PSIDE 15                                      /* 2*radius+1 <= 15 */
#define ANTSRL_MAX_PCELLS (ANTSRL_MAX_PSIDE * ANTSRL_MAX_PSIDE)
#define ANTSRL_MAX_FILTER_RADIUS 3
#define ANTSRL_MAX_FILTER_TAPS 49
#define ANTSRL_MAX_PHERO 4

/* error codes */
#define ANTSRL_OK 0
#define ANTSRL_E_INVALID (-1)   /* bad argument / configuration */
#define ANTSRL_E_NOMEM (-2)     /* workspace too small */
#define ANTSRL_E_DEVICE (-3)    /* HIP runtime error (launch / no device) */
#define ANTSRL_E_UNSUPPORTED (-4)

/* perceived-object kinds: one perception channel per entry of
 * RLApi.perceived_objects (environment/RL_api.py:123-142) */
enum {
    ANTSRL_CH_ANTS = 0,    /* RL_api.py:136-142  presence 0/1            */
    ANTSRL_CH_PHERO = 1,   /* RL_api.py:124-125  phero/max_val (arg = i) */
    ANTSRL_CH_ANTHILL = 2, /* RL_api.py:130-131  area                    */
    ANTSRL_CH_WALLS = 3,   /* RL_api.py:128-129  map                     */
    ANTSRL_CH_FOOD = 4,    /* RL_api.py:126-127  qte                     */
    ANTSRL_CH_ROCKS = 5    /* RL_api.py:132-135  any(dist < radius)      */
};

/* Pheromone.update strategy.  With the shipped centre-only DIFFUSE_FILTER (DIFFUSE_FACTOR = 0,
 * pheromone.py:5-10) the update is a per-cell multiply by f0 = 1 - EVAP_FACTOR plus a cut at
 * 0.01.  AUTO then stores the grid in units of f0^S (S = updates so far), so evaporation costs no
 * per-step pass over the grid at all; values are materialised (v = u * f0^S, zero below the cut)
 * wherever they are read.  EXPLICIT_SWEEP forces the streaming sweep kernel (k_sweep0); filters with a
 * radius always take an explicit per-step sweep: a register-marching stencil (k_sweep_r1x2 / k_sweep_sep2 /
 * k_sweep_march). */
enum { ANTSRL_PHERO_AUTO = 0, ANTSRL_PHERO_EXPLICIT_SWEEP = 1 };

/* Which kernels run RLApi.step / RLApi.observation.  AUTO picks by measurement: the cell-meta path (k_move +
 * k_perceive) wherever it is supported (the generator's channel order, two pheromone channels, a perception of 128..368
 * values per ant, at most 4096 ants per environment), the single kernel k_act elsewhere.  The other two values pin a
 * path (antsrl_create refuses a configuration the pinned path does not support): results are identical, both are
 * parity-tested. */
enum { ANTSRL_ACT_AUTO = 0, ANTSRL_ACT_CELL_META = 1, ANTSRL_ACT_SINGLE_KERNEL = 2 };

/* reward kinds (environment/rewards/) */
enum {
    ANTSRL_REWARD_NONE = 0,        /* Reward base: zeros, reward.py:19,38       */
    ANTSRL_REWARD_EXPLORATION = 1, /* ExplorationReward, reward_custom.py:8-25  */
    ANTSRL_REWARD_FOOD = 2,        /* Food_Reward, reward_custom.py:28-40       */
    ANTSRL_REWARD_ALL = 3          /* All_Rewards, reward_custom.py:43-109      */
};

/* Static configuration of a batch of environments.  Gathers the constructor
 * arguments and module constants the reference spreads over RLApi.__init__
 * (RL_api.py:23), RLApi.setup_perception (RL_api.py:80-93), DELTA (RL_api.py:15),
 * EnvironmentGenerator.__init__ (generator/environment_generator.py:20-46),
 * Ants.__init__ (ants.py:18), Pheromone (pheromone.py:5-10,21) and the reward
 * constructors (rewards/reward_custom.py:44). */
typedef struct AntsCfg {
    int32_t abi_version; /* = ANTSRL_ABI_VERSION */
    int32_t n_envs;      /* E : independent environments in this batch   */
    int32_t n_ants;      /* N : ants per environment                      */
    int32_t w, h;        /* grid; cell (x,y) is element [x][y], y fastest */
    int32_t n_phero;     /* C : pheromone channels (<= ANTSRL_MAX_PHERO)  */
    int32_t n_rocks;     /* R : circle obstacles per env (0 = none)       */
    int32_t max_time;    /* Environment.max_time, environment.py:26       */

    /* perception (RLApi.setup_perception) */
    int32_t perception_radius;                  /* r, side P = 2r+1             */
    int32_t n_channels;                         /* K = len(perceived_objects)   */
    int32_t channel_kind[ANTSRL_MAX_CHANNELS];  /* ANTSRL_CH_*                  */
    int32_t channel_arg[ANTSRL_MAX_CHANNELS];   /* pheromone index for CH_PHERO */
    int32_t has_mask;                           /* 0: perception_mask is None   */
    uint8_t mask[ANTSRL_MAX_PCELLS];            /* [P][P] row-major, 1=visible  */
    uint8_t _pad0[7];
    double delta;     /* DELTA = 1.1, RL_api.py:15                      */
    double fwd_delta; /* perception_shift = 4, environment_generator.py:43 */

    /* kinematics (RLApi.__init__, main.py:45-50) */
    double max_speed, max_rot_speed, carry_speed_reduction, backward_speed_reduction;
    double max_hold; /* Ants.max_hold = 5, environment_generator.py:93 */

    /* pheromone (pheromone.py:5-10, 36-45) */
    int32_t has_max_val;      /* Pheromone.max_val is not None                   */
    int32_t filter_radius;    /* 0..3 ; DIFFUSE_FILTER side = 2*radius+1         */
    double phero_max_val;     /* 255                                             */
    double deposit_strength;  /* what activate_pheromone's 256 becomes: 1.0 while
                                 phero_activation is bool (ants.py:83), 256.0 once
                                 an agent called activate_all_pheromones(float)   */
    double phero_threshold;   /* 0.01, pheromone.py:45                           */
    double filter[ANTSRL_MAX_FILTER_TAPS]; /* DIFFUSE_FILTER [side][side] row-major
                                 (first index along x), applied as scipy's
                                 convolve2d(.., 'same', 'fill', 0), pheromone.py:44 */

    /* reward */
    int32_t reward_kind; /* ANTSRL_REWARD_* */
    int32_t phero_mode;  /* ANTSRL_PHERO_AUTO (0) or ANTSRL_PHERO_EXPLICIT_SWEEP (1), see below */
    double reward_threshold; /* RLApi.reward_threshold, RL_api.py:32,203 */
    double fct_explore, fct_food, fct_anthill, fct_explore_holding, fct_headinganthill;

    /* Walls.update jitter (walls.py:28) when no explicit draws are supplied:
     * counter-based generator keyed on (rng_seed, GLOBAL env id, timestep, ant). */
    uint64_t rng_seed;

    int32_t act_path; /* ANTSRL_ACT_* (no reference counterpart) */

    /* GLOBAL ENVIRONMENT IDENTITY (ABI 5).  The reference seeds every environment by itself
     * (generator/environment_generator.py:53-55; the np.random stream Walls.update draws from, walls.py:28, belongs
     * to that environment), so an environment's trajectory must not depend on which handle / rank / batch position
     * it lands on.  Environment e of this handle IS global environment env_id_base + e: every random stream the
     * library keys on an environment — the built-in wall jitter, both generators of antsrl_generate — takes the
     * global id.  A handle over envs [lo, hi) of a sharded batch with env_id_base = lo reproduces rows lo:hi of the
     * whole-batch handle bit for bit (tests/test_gpu_shard_identity.py).  n_envs_total: environments in the whole
     * (sharded) batch, the stride between the auto-reset episodes of ANTSRL_RNG_REFERENCE (env g of episode k draws
     * from episode_seed + k * n_envs_total + g); 0 = env_id_base + n_envs. */
    int32_t env_id_base;
    int32_t n_envs_total;
    int32_t _pad1;
} AntsCfg;

/* Initial state of every environment = what EnvironmentGenerator.generate
 * (generator/environment_generator.py:52-106) builds.  All device pointers. */
typedef struct AntsInit {
    const double *ants_xyt;    /* [E][N][3]  Ants.ants, ants.py:27              */
    const double *seed;        /* [E][N]     Ants.seed, ants.py:41              */
    const uint8_t *walls;      /* [E][W][H]  Walls.map, walls.py:14             */
    const float *food;         /* [E][W][H]  Food.qte, food.py:15               */
    const int32_t *anthill_xyr;/* [E][3]     Anthill x,y,radius, anthill.py:17  */
    const double *rocks;       /* [E][R][4]  cx,cy,radius,weight (NULL if R=0),
                                              circle_obstacles.py:16             */
    const float *phero;        /* [E][C][W][H] or NULL (= zeros), pheromone.py:28-31 */
} AntsInit;

/* Parameters of the device-side episode generator (antsrl_generate): the knobs of
 * EnvironmentGenerator.__init__ / CirclesGenerator (generator/environment_generator.py:20-46,
 * generator/map_generators.py:28-33, main.py:70-77) that are not already in AntsCfg. */
#define ANTSRL_MAX_FOOD_DISCS 64
#define ANTSRL_WALLS_BERNOULLI 0 /* independent wall cells with probability wall_density */
#define ANTSRL_WALLS_PERLIN 1    /* PerlinGenerator (generator/map_generators.py:9-25, main.py:75):
                                    wall = pnoise2((x + ox) / scale, (y + oy) / scale, octaves, persistence,
                                    lacunarity) > wall_density, per-env offsets ox, oy uniform in
                                    [-10000, 10000]; improved Perlin noise restated from the published
                                    algorithm in float32 (the `noise` package is absent: unpinned) */
#define ANTSRL_WALLS_INPUT 2     /* the caller's bitmap (AntsGen.walls_input, uint8 [E][W][H], device memory): what any
                                    walls_generator.generate(w, h) returned; cleared on the anthill area
                                    like environment_generator.py:66-67 */
/* random streams of the device generator */
#define ANTSRL_RNG_COUNTER 0     /* counter-based, keyed on (episode_seed, global env id, item): same distributions as
                                    the reference's generator, different maps (the oracle restates it) */
#define ANTSRL_RNG_REFERENCE 1   /* the reference's own streams: env e (global id g = AntsCfg.env_id_base + e) is drawn
                                    like EnvironmentGenerator(seed = episode_seed + g).generate — random.seed(seed) / np.random.seed(seed * 5)
                                    (environment_generator.py:53-55), i.e. two MT19937 generators per env with
                                    Python's and numpy's seeding and 53-bit doubles, consumed in the reference's order
                                    (anthill :60-63, PerlinGenerator's two randints map_generators.py:19-20,
                                    CirclesGenerator :37-39, rocks :77-85, ants :87-91, Ants.seed ants.py:41): equal
                                    seeds give the reference's anthill, food discs, ants and seeds bit for bit.
                                    Walls: ANTSRL_WALLS_PERLIN or ANTSRL_WALLS_INPUT.
                                    (episode_seed + env_id_base + E) * 5 < 2^32. */
typedef struct AntsGen {
    double wall_density;   /* Bernoulli: probability of a wall cell; Perlin: PerlinGenerator.density (threshold) */
    int32_t n_food_discs;  /* CirclesGenerator.n_circles, main.py:74 uses 20 (<= ANTSRL_MAX_FOOD_DISCS) */
    int32_t food_rmin, food_rmax; /* CirclesGenerator min/max radius, main.py:74 uses 5, 10 */
    int32_t auto_reset;    /* 1: antsrl_step_update regenerates every env right after the update of
                              the step that reported done (RL_api.py:200), with the next episode seed */
    int32_t wall_kind;     /* ANTSRL_WALLS_* */
    int32_t perlin_octaves;      /* PerlinGenerator defaults: 2 (1..8) */
    double perlin_scale;         /* 22.0 */
    double perlin_persistence;   /* 0.5 */
    double perlin_lacunarity;    /* 2.0 */
    int32_t rng_kind;            /* ANTSRL_RNG_* */
    int32_t _pad;
    const uint8_t *walls_input;  /* ANTSRL_WALLS_INPUT: uint8 [E][W][H] device memory, else NULL */
} AntsGen;

typedef struct AntsHandle AntsHandle;

/* selectors for antsrl_read_state: canonical (reference-shaped) layouts */
enum {
    ANTSRL_S_ANTS_XYT = 0,     /* double  [E][N][3]                     */
    ANTSRL_S_PREV_XY = 1,      /* double  [E][N][2]   Ants.prev_ants    */
    ANTSRL_S_HOLDING = 2,      /* float   [E][N]                        */
    ANTSRL_S_MANDIBLES = 3,    /* uint8   [E][N]                        */
    ANTSRL_S_ACTIVATION = 4,   /* float   [E][N][C]   phero_activation  */
    ANTSRL_S_PHERO = 5,        /* float   [E][C][W][H]                  */
    ANTSRL_S_FOOD = 6,         /* float   [E][W][H]                     */
    ANTSRL_S_EXPLORED = 7,     /* uint8   [E][W][H]   reward explored_map */
    ANTSRL_S_ANTHILL_FOOD = 8, /* double  [E]         Anthill.food      */
    ANTSRL_S_ROCK_CENTERS = 9, /* double  [E][R][2]                     */
    ANTSRL_S_TIMESTEP = 10,    /* int32   [E]                           */
    ANTSRL_S_REWARD_STATE = 11,/* uint8   [E][N]      Ants.reward_state */
    ANTSRL_S_WALLS = 12,       /* uint8   [E][W][H]                     */
    ANTSRL_S_ANTHILL_AREA = 13,/* uint8   [E][W][H]   Anthill.area      */
    ANTSRL_S_SEED = 14,        /* float   [E][N]                        */
    ANTSRL_S_ANTHILL_XYR = 15, /* int32   [E][3]      Anthill.x, .y, .radius (anthill.py:21-23) */
    ANTSRL_S_ROCK_RW = 16,     /* double  [E][R][2]   CircleObstacles.radiuses, .weights (circle_obstacles.py:19-20) */
    ANTSRL_S_PHERO_C0 = 17,    /* float   [E][W][H]   one pheromone channel: Pheromone.phero of pheromone 0 ...   */
    ANTSRL_S_PHERO_C1 = 18,    /*                     ... 1 (a view of ONE Pheromone object reads one channel,    */
    ANTSRL_S_PHERO_C2 = 19,    /*                     not all of them)                                            */
    ANTSRL_S_PHERO_C3 = 20,
    ANTSRL_S_COUNT_
};

/* Library / ABI version (ANTSRL_ABI_VERSION of the build). */
int antsrl_abi_version(void);

/* sizeof(AntsCfg) as the library was compiled: lets a binding verify its struct mirror. */
size_t antsrl_cfg_size(void);

/* Message for the last error returned on this thread ("" if none). */
const char *antsrl_last_error(void);

/* Bytes of device workspace a batch with this configuration needs.
 * No reference counterpart (the reference allocates numpy arrays per object). */
int antsrl_workspace_bytes(const AntsCfg *cfg, size_t *bytes);

/* Creates a handle over a caller-owned device workspace (>= workspace_bytes,
 * 256-byte aligned).  Replaces RLApi.__init__ (environment/RL_api.py:23) +
 * RLApi.setup_perception (RL_api.py:80-93).  Host-only: touches no device state. */
int antsrl_create(const AntsCfg *cfg, void *workspace, size_t workspace_bytes, AntsHandle **out);

void antsrl_destroy(AntsHandle *h);

/* Loads the initial state of all E environments ("reset"): replaces
 * EnvironmentGenerator.generate (generator/environment_generator.py:52-106)
 * object construction, Anthill.__init__ area rasterisation (anthill.py:28-33),
 * RLApi.register_ants (RL_api.py:57-66) and Reward.setup (rewards/reward.py:12-19,
 * reward_custom.py:13-15,33-35,65-77).  timestep := 1 (environment.py:27). */
int antsrl_reset(AntsHandle *h, const AntsInit *init, void *stream);

/* Episode "reset" on the device: draws what EnvironmentGenerator.generate
 * draws (generator/environment_generator.py:52-106) — anthill in the central half, walls cleared on
 * the anthill, food discs zeroed on walls, rocks in the generator's band, ants in a disc of 0.8 r
 * around the anthill — and loads it exactly like antsrl_reset.  gen->rng_kind picks the random source:
 *   ANTSRL_RNG_COUNTER    a counter-based generator keyed on (episode_seed, global env id, item): the reference's
 *                         distributions, NOT its maps (the oracle's oracle_generate restates this generator);
 *   ANTSRL_RNG_REFERENCE  the reference's own MT19937 streams (Python's `random` and `np.random`, seeded like
 *                         environment_generator.py:53-55): env e equals EnvironmentGenerator(seed = episode_seed +
 *                         env_id_base + e)
 *                         — anthill, food discs, rocks, ants and per-ant seeds bit for bit (see ANTSRL_RNG_* above).
 * With gen->auto_reset the handle keeps `gen` and re-runs it after every finished episode: with episode_seed + 1,
 * + 2, ... (ANTSRL_RNG_COUNTER) or AntsCfg.n_envs_total seeds further each time (ANTSRL_RNG_REFERENCE: global env g of
 * episode k takes seed episode_seed + k * n_envs_total + g, whatever the sharding).  Two limits of that mode:
 * np.random.seed takes 32 bits, so (seed + env_id_base + E) * 5 must stay
 * below 2^32 — checked again at every auto-reset, which returns ANTSRL_E_INVALID instead of wrapping; and with
 * ANTSRL_WALLS_INPUT every episode re-uses the bitmaps of the first call (the reference calls walls_generator.generate
 * per episode: pass fresh bitmaps through antsrl_generate between episodes if the walls are to change). */
int antsrl_generate(AntsHandle *h, const AntsGen *gen, uint64_t episode_seed, void *stream);

/* RLApi.step (environment/RL_api.py:168-204): mandibles/food exchange, pheromone
 * activation, rotate, forward move, observation, reward, done.
 *   rotation  int8 [E][N] in {-1,0,1} or NULL (= leave theta unchanged, RL_api.py:190)
 *   phero     int8 [E][N] in {0,1,2}  or NULL (= leave activation, RL_api.py:187)
 *   obs         float [E][N][P][P][K]   perception  (may be NULL: skip the write — rewards, agent_state and, with
 *                                       antsrl_set_inloop_policy, the next actions are produced all the same: the
 *                                       act-only rollout of collect_agent_memory.py:189-199 with training=False)
 *   agent_state float [E][N][2]         [holding, seed], RL_api.py:160-162
 *   reward      float [E][N]            Reward.step, rewards/reward.py:38
 *   done        uint8 [E]               RL_api.py:200                         */
int antsrl_step(AntsHandle *h, const int8_t *rotation, const int8_t *phero, float *obs,
                float *agent_state, float *reward, uint8_t *done, void *stream);

/* RLApi.observation (environment/RL_api.py:96-165) on the current state, including
 * its reward side effects (reward.observation, RL_api.py:164).  reward may be NULL. */
int antsrl_observe(AntsHandle *h, float *obs, float *agent_state, float *reward, void *stream);

/* Environment.update (environment/environment.py:42-47): timestep += 1, then
 * Walls (walls.py:22-30), CircleObstacles (circle_obstacles.py:32-58), Pheromone
 * (pheromone.py:43-45), Ants (ants.py:123-130), Anthill (anthill.py:41-46).
 *   wall_jitter  double [E][N] or NULL.  When given, entry k of env e is the k-th
 *   value np.random.random(k) would have returned in Walls.update (walls.py:28):
 *   the k-th colliding ant, in ant-index order, consumes it.  NULL = built-in
 *   counter-based generator keyed on (AntsCfg.rng_seed, AntsCfg.env_id_base + e, timestep, ant).
 * DEFERRED UPDATE.  With wall_jitter == NULL on the cell-meta path (<= 1024 ants, ANTSRL_Q_DEFERRED_UPDATE; scaled
 * pheromone units or an explicit sweep alike) the call does the update's bookkeeping (and enqueues the pheromone sweep, if
 * there is one) and returns without enqueuing the update's kernel: the
 * next antsrl_step / antsrl_step_update runs it in the same launch as its move (k_update_move: the move re-reads
 * what the update has just written — one launch and most of the second kernel's HBM fetches saved).  Every other
 * entry point that reads or replaces the state (antsrl_observe, antsrl_read_state, antsrl_set_activation, a second
 * antsrl_update, antsrl_reset / antsrl_generate) enqueues or drops it first, on ITS stream argument: results are
 * the same as with an immediate launch, bit for bit; only the moment the kernel is enqueued moves.  Callers that
 * alternate streams between calls must order them as they already have to for the state itself.
 * WORKSPACE CONSISTENCY: while an update is deferred, the workspace still holds the pre-update state.  A caller that
 * copies or checkpoints the workspace bytes, records an event "after the update" or times the update on its own calls
 * antsrl_flush first. */
int antsrl_update(AntsHandle *h, const double *wall_jitter, void *stream);

/* Environment.update (environment/environment.py:42-47) ONE REFERENCE STEP AT A TIME, for callers whose own EnvObjects must
 * run BETWEEN the world's objects: the reference sorts every object of the environment by update_step() (stable) and calls
 * them in turn — Walls (-1), Food / CircleObstacles / Pheromone / RLApi (0), Ants (999), Anthill (1000); an object the
 * caller added with, say, update_step() == 500 runs after the rocks and the pheromone update and before the ants'.
 * antsrl_update_phase runs one of the four device steps and returns; the four calls in order are one antsrl_update, bit for
 * bit (same device functions, cut at launch boundaries; never deferred).  antsrl_read_state between two phases sees the
 * state as the reference's object would (positions already reverted off walls after WALLS, the decayed / diffused grid
 * after ROCKS_PHEROMONE, ...).  `wall_jitter` as for antsrl_update; only WALLS reads it.  Environment.timestep advances
 * with ANTHILL (the reference increments it before the first object: a binding that exposes `timestep` adds one while an
 * update is in progress).  While phases are outstanding every entry point that changes the state returns ANTSRL_E_INVALID. */
enum { ANTSRL_PHASE_WALLS = 0, ANTSRL_PHASE_ROCKS_PHEROMONE = 1, ANTSRL_PHASE_ANTS = 2, ANTSRL_PHASE_ANTHILL = 3 };
int antsrl_update_phase(AntsHandle *h, int phase, const double *wall_jitter, void *stream);

/* Enqueues a deferred update's kernel on `stream` now (no-op when none is pending): afterwards every kernel the
 * handle owes has been enqueued and, once `stream` has drained, the workspace holds the complete state.  No reference
 * counterpart (the reference updates eagerly; this is the price of k_update_move).  antsrl_destroy drops a pending
 * update with the handle: the workspace is the caller's, flush first if its content is still wanted. */
int antsrl_flush(AntsHandle *h, void *stream);

/* main.py:98 followed by main.py:131 — one full simulation step. */
int antsrl_step_update(AntsHandle *h, const int8_t *rotation, const int8_t *phero,
                       const double *wall_jitter, float *obs, float *agent_state, float *reward,
                       uint8_t *done, void *stream);

/* Measurement hook (no reference counterpart; the reference only keeps a wall-clock EMA,
 * main.py:93,132-136).  events = ANTSRL_TIMING_EVENTS caller-created hipEvent_t, or NULL to disable.
 * While set, the NEXT antsrl_step_update records them on its stream: [0] before the pheromone sweep,
 * [1] after it, [2] after the per-ant action kernel (k_move; equal to [1] where one kernel does both),
 * [3] after the perception kernel (k_perceive / k_act), [4] after the update kernel; the hook then
 * clears itself.  With a deferred update (antsrl_update) [1]..[2] brackets k_update_move — the PREVIOUS step's
 * update and this step's move — and [3]..[4] is empty under scaled units; under an explicit sweep the step's sweep
 * follows its kernels then ([0]..[1] empty, [3]..[4] brackets the sweep: the deferred deposits land in its input). */
#define ANTSRL_TIMING_EVENTS 5
int antsrl_set_timing_events(AntsHandle *h, void *const *events);

/* What the handle resolved its configuration to (no reference counterpart; bench.py and the tests name the
 * kernels and byte models from it). */
enum {
    ANTSRL_Q_CELL_META = 0,        /* 1: k_move + k_perceive on the cell-meta layout, 0: k_act           */
    ANTSRL_Q_SCALED_UNITS = 1,     /* 1: pheromone held in units of f0^S (no per-step sweep)              */
    ANTSRL_Q_INTERLEAVED = 2,      /* 1: {p0, p1, food, meta} 16-byte cell records                        */
    ANTSRL_Q_FILTER_SEPARABLE = 3, /* 1: DIFFUSE_FILTER detected as rank-1 (separable stencil march)      */
    ANTSRL_Q_PERCEIVE_RUN = 4,     /* ants per wave of k_perceive (0 without the cell-meta path)          */
    ANTSRL_Q_TIMESTEP = 5,         /* Environment.timestep (environment.py:27,45) as the host mirrors it: every env of a
                                      handle steps in lockstep, so no device read is needed                */
    ANTSRL_Q_DEFERRED_UPDATE = 6,  /* 1: antsrl_update(NULL jitter) is deferred into the next step (k_update_move) */
    ANTSRL_Q_COUNT_
};
int antsrl_query(const AntsHandle *h, int what, long long *value);

/* Measurement helper (no reference counterpart): a plain device-to-device copy of `bytes` bytes (multiple of
 * 16, both pointers 16-byte aligned) with 16 bytes per lane, enqueued on `stream` — bench.py times it for the
 * box's achievable read + write bandwidth next to the 8 TB/s specification. */
int antsrl_bench_copy(void *dst, const void *src, size_t bytes, void *stream);

/* Device memory for the step's big buffers — the workspace and the observation tensor — (no reference counterpart; the
 * step entry points never allocate: this is an allocator the CALLER may use for the buffers it owns).  The memory is one
 * virtual range backed by physical pieces of at most ANTSRL_MEM_PIECE_BYTES (hipMemCreate / hipMemMap).  On MI355X the
 * physical layout of these two buffers is worth 15 % of the observation kernel: when both lie in physically contiguous
 * ranges of 128 MiB or more (what hipMalloc hands a fresh process) the observation write stream and the cell-record
 * gathers alias on the memory channels; with either buffer in pieces of at most 32 MiB they do not — k_perceive 0.167 ms
 * against 0.197 ms at 1024 envs x 512 ants, on every allocation (profiles/history/r04/placement_probe4*.txt).  The pointer is
 * aligned to the device's allocation granularity (2 MiB); contents are undefined; free with antsrl_mem_free (never hipFree).
 * antsrl_mem_free waits for the block's device and PARKS the block, still mapped, in a per-device pool; antsrl_mem_alloc hands
 * a parked block of the same device and (piece-rounded) size back before it maps anything new.  Nothing is unmapped while
 * the program runs, so no address is ever translated to other memory than it was first mapped to (a range that is unmapped,
 * freed and reserved again can meet stale GPU translations on ROCm 7.2: antsrl_mem.hip) and the reserved address space is
 * bounded by the blocks that were alive at once.  antsrl_mem_trim returns the parked blocks' physical memory to the device
 * (their ranges are retired, never reused); antsrl_mem_stats reports bytes handed out / parked / reserved / retired (any
 * pointer may be NULL).  Returns ANTSRL_E_NOMEM when the device cannot supply the pieces, ANTSRL_E_DEVICE when the runtime
 * lacks the virtual-memory API. */
#define ANTSRL_MEM_PIECE_BYTES ((size_t)16 << 20)
int antsrl_mem_alloc(size_t bytes, int device, void **ptr);
int antsrl_mem_free(void *ptr);
int antsrl_mem_trim(void);
int antsrl_mem_stats(size_t *live_bytes, size_t *pooled_bytes, size_t *reserved_va_bytes, size_t *retired_va_bytes);

/* Observation tensor format (no reference counterpart: the reference's perception is float64 numpy,
 * cast to float32 by torch.Tensor(state) in the agents, collect_agent_memory.py:194).
 * ANTSRL_OBS_F32 (default): `obs` arguments are float [E][N][P][P][K].
 * ANTSRL_OBS_BF16: `obs` arguments point to bfloat16 (uint16_t) buffers of the same shape holding the
 * same values rounded to nearest even — half the bytes written per step, and what the bf16 policy
 * (antsrl_policy_mlp*) rounds its input to anyway.  Supported for 2 pheromone channels, the
 * generator's channel order and perceptions of at most 64 cells; otherwise the step returns
 * ANTSRL_E_UNSUPPORTED. */
#define ANTSRL_OBS_F32 0
#define ANTSRL_OBS_BF16 1
int antsrl_set_obs_format(AntsHandle *h, int format);

/* Observation row stride (no reference counterpart; opt-in, the default is the dense [E][N][P][P][K] tensor of
 * RL_api.py:122).  A row of P*P*K values is 1 372 bytes at the reference's 7x7x7 float32 perception: no row starts or ends
 * on a 128-byte line.  With stride_elems = the row rounded up to whole lines (float32 7x7x7: 352 elements = 1 408 bytes;
 * bfloat16: 384) the `obs` arguments point to [E][N][stride_elems] buffers (128-byte aligned): element k of ant a's
 * perception lies at a * stride_elems + k, the padding elements behind it are written as zeros, and every copy-out of
 * the observation kernel is whole lines of one wave's own.  A torch caller sees the reference's shape through a view:
 * buf.view(E, N, stride)[..., :P*P*K].view(E, N, P, P, K).  0 (or P*P*K) = dense.  Cell-meta path only
 * (ANTSRL_Q_CELL_META); not together with antsrl_set_inloop_policy; antsrl_set_obs_format resets it. */
int antsrl_set_obs_row_stride(AntsHandle *h, int32_t stride_elems);

/* Ants.activate_all_pheromones (environment/ants.py:86-87).  act: float [E][N][C].
 * new_deposit_strength > 0 also changes AntsCfg.deposit_strength (the dtype switch
 * of SURVEY.md §8(a) A4); pass 0 to keep it. */
int antsrl_set_activation(AntsHandle *h, const float *act, double new_deposit_strength,
                          void *stream);

/* In-loop policy inference (BASELINE config 5): the reference's linear DQN nets
 * (agents/explore_agent_pytorch.py:24-45, agents/collect_agent.py:24-51) evaluated on the
 * observation tensor without leaving the device, bf16 operands / fp32 accumulation (MFMA):
 *     out = layer1(cat[obs.view(M, F), agent_state.view(M, 2)])   layer1: w1 float [32][F+2], b1 [32]
 *     rotation  = argmax(layer2(out)) - 1                         layer2: w2 float [3][32],  b2 [3]
 *     pheromone = argmax(layer3(out))      (skipped if w3 NULL)   layer3: w3 float [3][32],  b3 [3]
 * Weights are PyTorch nn.Linear layouts, device pointers.  M = number of ants (E*N), F = P*P*K.
 * rotation/pheromone: int8 [M], directly usable as antsrl_step's actions.  logits (float [M][6],
 * nullable) receives the six head outputs.  `h` may be NULL; a handle set to ANTSRL_OBS_BF16
 * (antsrl_set_obs_format) makes `obs` a bfloat16 buffer of the same shape. */
int antsrl_policy_mlp(AntsHandle *h, const float *obs, const float *agent_state, int64_t n_ants, int32_t n_features,
                      const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                      const float *b3, int8_t *rotation, int8_t *pheromone, float *logits, void *stream);

/* The same network evaluated INSIDE the observation kernel (no reference counterpart; BASELINE config 5's loop:
 * action = agent.get_action(obs) of main.py:96 for the NEXT step, computed where the rows are produced).  While set,
 * every antsrl_step / antsrl_step_update / antsrl_observe — with an observation buffer or with obs == NULL — also stores
 *   rotation_next int8 [E][N] = argmax(layer2(out)) - 1     pheromone_next int8 [E][N] = argmax(layer3(out)) (or NULL)
 * for the rows it has just produced — the values antsrl_policy_mlp returns for that observation tensor, bit for bit
 * (same bf16 fragments, same order of MFMAs and additions) without re-reading it from HBM; with obs == NULL the rows
 * never leave the workgroup's LDS (no observation traffic at all).  Needs the cell-meta
 * path with bfloat16 observations (ANTSRL_Q_CELL_META, antsrl_set_obs_format); ANTSRL_E_UNSUPPORTED otherwise.
 * The weights (device pointers, float32, row-major like nn.Linear: w1 [32][n_features + 2], w2 / w3 [3][32]) are
 * copied into the handle's workspace at the call; w1 == NULL switches the in-loop policy off. */
int antsrl_set_inloop_policy(AntsHandle *h, int32_t n_features, const float *w1, const float *b1, const float *w2,
                             const float *b2, const float *w3, const float *b3, int8_t *rotation_next,
                             int8_t *pheromone_next, void *stream);

/* Copies one piece of state into a caller device buffer in the canonical
 * reference-shaped layout (ANTSRL_S_*).  Replaces attribute reads such as
 * api.ants.ants, pheromone.phero, food.qte, anthill.food. */
int antsrl_read_state(AntsHandle *h, int which, void *dst, void *stream);

/* RLApi.perceptive_field (environment/RL_api.py:144-153; main.py:51 sets save_perceptive_field for the viewer): dst
 * uint8 [E][W][H] = 1 where the perception of some ant of the environment reaches — masked cells not counted — computed from
 * the ants' positions as they stand: call it right behind antsrl_step / antsrl_observe (the reference computes it inside
 * RLApi.observation).  A deferred update (antsrl_update with the library's jitter) has not moved anything yet and is NOT
 * flushed by this call. */
int antsrl_perceptive_field(AntsHandle *h, uint8_t *dst, void *stream);

/* Size in bytes of what antsrl_read_state(which) writes. */
int antsrl_state_bytes(const AntsHandle *h, int which, size_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* ANTSRL_H */
