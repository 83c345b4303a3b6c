/*
 * antsrl_oracle.c — CPU restatement of the AntsRL environment step loop.
 *
 * TEST INFRASTRUCTURE ONLY (see antsrl_oracle.h).  Plain C, float64 throughout like
 * the reference, explicit per-ant / per-cell loops, one environment at a time
 * (environments are independent; the batch wrappers at the bottom optionally spread
 * them over OpenMP threads for the timed cpu_baseline leg of bench.py).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * golden vectors recorded from the real reference (tests/golden/make_golden.py):
 * ant x/y/theta, holding, mandibles, food, anthill.food, rock centres, explored map,
 * perception, agent_state, reward, done — bit-exact in float64; pheromone bit-exact
 * for the shipped centre-only filter and within 1e-12 relative when a diffusion
 * filter is patched in (scipy's summation order is not restated).
 *
 * Build with -ffp-contract=off: numpy rounds every product before adding, so fused
 * multiply-add must not be formed.
 *
 * All file:line citations are relative to the reference checkout.
 */
#include "antsrl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.141592653589793 /* np.pi */

/* np.mod on float64 (numpy npy_divmod semantics): result takes the sign of b. */
static double np_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0) {
        if ((b < 0) != (r < 0)) r += b;
    } else {
        r = copysign(0.0, b);
    }
    return r;
}

/* Ants.warp_xy, environment/ants.py:69-71.  np.mod(-1e-17, W) rounds to exactly W,
 * where the reference would raise IndexError on its next grid access (SURVEY §8(a)
 * A2); this restatement maps that single value to 0 instead. */
static double warp_coord(double v, double size)
{
    double r = np_mod(v, size);
    if (r >= size) r = 0.0;
    return r;
}

/* python int % int for the wrapped perception coordinates, RL_api.py:118-119 */
static long imod(long a, long b)
{
    long r = a % b;
    if (r < 0) r += b;
    return r;
}

/* ---- counter-based jitter generator (specification shared with the device) ---- */
static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

double oracle_jitter_u01(uint64_t seed, uint32_t env, uint32_t timestep, uint32_t ant)
{
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)env + 1));
    k = mix64(k ^ (0xD1B54A32D192ED03ULL * ((uint64_t)timestep + 1)));
    k = mix64(k + 0x9E3779B97F4A7C15ULL * ((uint64_t)ant + 1));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0); /* 53-bit mantissa, [0,1) */
}

/* ---- device-side episode generator restated (antsrl_generate) ----
 * Same draws as the HIP kernels k_gen_env / k_gen_cells / k_gen_ants, which mirror what
 * EnvironmentGenerator.generate builds (generator/environment_generator.py:52-106) but from a
 * counter-based generator instead of Python's MT19937 streams. */
#define GEN_SALT 0x6A09E667F3BCC909ULL
enum { GEN_ANTHILL = 0, GEN_WALLS = 1, GEN_FOOD = 2, GEN_ROCKS = 3, GEN_ANT_ANGLE = 4, GEN_ANT_DIST = 5,
       GEN_ANT_THETA = 6, GEN_ANT_SEED = 7, GEN_WALL_OFFSET = 8 };
static double gen_u01(uint64_t seed, uint32_t env, uint32_t tag, uint32_t idx)
{
    return oracle_jitter_u01(seed ^ GEN_SALT, env, tag, idx);
}

/* ---- Perlin walls (generator/map_generators.py:9-25 -> utils.py:7-17 -> noise.pnoise2) ----
 * The `noise` package is a third-party dependency that is neither installed nor under /root/reference
 * (no pinned version: the reference has no requirements file).  Restated from the published algorithm
 * (K. Perlin, "Improving Noise", 2002: reference permutation, fade 6t^5 - 15t^4 + 10t^3, gradient from
 * the low 4 bits of the hash over 16 directions; fractal sum returned as total / sum of amplitudes), in
 * float32.  PARITY UNPINNED against the package; pinned against antsrl_amd/generator.py::perlin_noise
 * (numpy, tests/test_generator.py) and the device generator (tests/test_gpu_generate.py). */
static const unsigned char PERLIN_P[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240,
    21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88,
    237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83,
    111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216,
    80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186,
    3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17,
    182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129,
    22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238,
    210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184,
    84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195,
    78, 66, 215, 61, 156, 180};
static int perlin_p(long k) { return PERLIN_P[k & 255]; } /* the doubled table of the classic code */

static float perlin_dot(int corner_hash, float dx, float dy)
{
    static const float dir[16][2] = {{1, 1}, {-1, 1}, {1, -1}, {-1, -1}, {1, 0}, {-1, 0}, {1, 0}, {-1, 0},
                                     {0, 1}, {0, -1}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {0, -1}, {0, 1}};
    const float *d = dir[perlin_p(corner_hash) & 15];
    return dx * d[0] + dy * d[1];
}

static float perlin_octave(float x, float y, float period_x, float period_y)
{
    long x0 = (long)floorf(fmodf(x, period_x)), y0 = (long)floorf(fmodf(y, period_y));
    long x1 = (long)fmodf((float)(x0 + 1), period_x), y1 = (long)fmodf((float)(y0 + 1), period_y);
    x0 &= 255; y0 &= 255; x1 &= 255; y1 &= 255;
    const float tx = x - floorf(x), ty = y - floorf(y);
    const float sx = tx * tx * tx * (tx * (tx * 6.0f - 15.0f) + 10.0f);
    const float sy = ty * ty * ty * (ty * (ty * 6.0f - 15.0f) + 10.0f);
    const int h00 = perlin_p(perlin_p(x0) + y0), h01 = perlin_p(perlin_p(x0) + y1);
    const int h10 = perlin_p(perlin_p(x1) + y0), h11 = perlin_p(perlin_p(x1) + y1);
    const float n00 = perlin_dot(h00, tx, ty), n10 = perlin_dot(h10, tx - 1.0f, ty);
    const float n01 = perlin_dot(h01, tx, ty - 1.0f), n11 = perlin_dot(h11, tx - 1.0f, ty - 1.0f);
    const float bottom = n00 + sx * (n10 - n00), top = n01 + sx * (n11 - n01);
    return bottom + sy * (top - bottom);
}

static double perlin_fractal(float x, float y, int octaves, float persistence, float lacunarity)
{
    if (octaves == 1) return (double)perlin_octave(x, y, 1024.0f, 1024.0f);
    float freq = 1.0f, amp = 1.0f, norm = 0.0f, sum = 0.0f;
    for (int o = 0; o < octaves; ++o) {
        const float period = (float)(1024.0 * (double)freq);
        sum = sum + perlin_octave(x * freq, y * freq, period, period) * amp;
        norm = norm + amp;
        freq = freq * lacunarity;
        amp = amp * persistence;
    }
    return (double)(sum / norm);
}

/* utils.py:7-17: gen[i][j] = pnoise2((i + offset_x) / scale, (j + offset_y) / scale, ...) */
void oracle_perlin_noise(int w, int h, long offset_x, long offset_y, double scale, int octaves, double persistence,
                         double lacunarity, double *out)
{
    for (long i = 0; i < w; ++i)
        for (long j = 0; j < h; ++j)
            out[i * h + j] = perlin_fractal((float)((double)(i + offset_x) / scale), (float)((double)(j + offset_y) / scale),
                                            octaves, (float)persistence, (float)lacunarity);
}

void oracle_generate_init(const AntsCfg *c, const AntsGen *g, uint64_t seed, double *ants_xyt, double *seed_out,
                          uint8_t *walls, float *food, int32_t *xyr, double *rocks)
{
    const int W = c->w, H = c->h, N = c->n_ants, R = c->n_rocks, m = W < H ? W : H;
    const size_t G = (size_t)W * H;
    for (int e = 0; e < c->n_envs; ++e) {
        const uint32_t ge = (uint32_t)c->env_id_base + (uint32_t)e; /* the env's GLOBAL id keys its streams (AntsCfg.env_id_base) */
        const long ax = (int)(gen_u01(seed, ge, GEN_ANTHILL, 0) * W * 0.5 + W * 0.25);   /* :61 */
        const long ay = (int)(gen_u01(seed, ge, GEN_ANTHILL, 1) * H * 0.5 + H * 0.25);   /* :62 */
        const long ar = (int)(gen_u01(seed, ge, GEN_ANTHILL, 2) * m * 0.05 + m * 0.05);  /* :63 */
        xyr[3 * e] = (int32_t)ax; xyr[3 * e + 1] = (int32_t)ay; xyr[3 * e + 2] = (int32_t)ar;
        for (int q = 0; q < R; ++q) {                                                   /* :77-85 */
            double *rk = rocks + ((size_t)e * R + q) * 4;
            rk[0] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 0) * (W * 0.75) + W * 0.25;
            rk[1] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 1) * (H * 0.25) + H * 0.25;
            rk[2] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 2) * 5 + 5;
            rk[3] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 3) * 50 + 50;
        }
        long discs[ANTSRL_MAX_FOOD_DISCS][3];
        for (int d = 0; d < g->n_food_discs; ++d) {                                     /* map_generators.py:37-40 */
            long rad = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 0) * (g->food_rmax - g->food_rmin) + g->food_rmin);
            const long cap = (m - 1) / 2;
            if (rad > cap) rad = cap;
            discs[d][0] = rad;
            discs[d][1] = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 1) * (W - 2 * rad) + rad);
            discs[d][2] = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 2) * (H - 2 * rad) + rad);
        }
        /* PerlinGenerator.generate draws random.randint(-10000, 10000) twice, map_generators.py:19-20 */
        const long pox = (long)(gen_u01(seed, ge, GEN_WALL_OFFSET, 0) * 20001.0) - 10000;
        const long poy = (long)(gen_u01(seed, ge, GEN_WALL_OFFSET, 1) * 20001.0) - 10000;
        for (long x = 0; x < W; ++x)
            for (long y = 0; y < H; ++y) {
                const size_t cell = (size_t)x * H + y;
                const int area = ar >= 0 && (ax - x) * (ax - x) + (ay - y) * (ay - y) <= ar * ar;
                int wall;                                                                 /* :66-67 */
                if (g->wall_kind == ANTSRL_WALLS_PERLIN)
                    wall = !area && perlin_fractal((float)((double)(x + pox) / g->perlin_scale),
                                                   (float)((double)(y + poy) / g->perlin_scale), g->perlin_octaves,
                                                   (float)g->perlin_persistence, (float)g->perlin_lacunarity) > g->wall_density;
                else
                    wall = !area && gen_u01(seed, ge, GEN_WALLS, (uint32_t)cell) < g->wall_density;
                int fd = 0;
                for (int d = 0; d < g->n_food_discs; ++d) {
                    const long dx = discs[d][1] - x, dy = discs[d][2] - y;
                    fd |= dx * dx + dy * dy <= discs[d][0] * discs[d][0];
                }
                walls[e * G + cell] = (uint8_t)wall;
                food[e * G + cell] = (fd && !wall) ? 1.0f : 0.0f;                        /* :71-72 */
            }
        for (int a = 0; a < N; ++a) {                                                    /* :87-93 */
            const double ang = gen_u01(seed, ge, GEN_ANT_ANGLE, a) * 2 * PI_D;
            const double dist = gen_u01(seed, ge, GEN_ANT_DIST, a) * (double)ar * 0.8;
            double *o = ants_xyt + ((size_t)e * N + a) * 3;
            o[0] = cos(ang) * dist + (double)ax;
            o[1] = sin(ang) * dist + (double)ay;
            o[2] = gen_u01(seed, ge, GEN_ANT_THETA, a) * 2 * PI_D;
            seed_out[(size_t)e * N + a] = (double)(float)gen_u01(seed, ge, GEN_ANT_SEED, a);
        }
    }
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* per-env views ------------------------------------------------------------- */
typedef struct EnvView {
    int N, W, H, C, R;
    double *x, *y, *theta, *prev_x, *prev_y, *holding, *seed, *activation;
    uint8_t *mandibles, *reward_state;
    double *phero, *food;
    uint8_t *walls, *area, *explored;
    int32_t *xyr;
    double *anthill_food;
    double *rcx, *rcy, *rrad, *rwt;
    int32_t *timestep;
    double *prev_holding, *prev_dist;
    uint8_t *primed;
} EnvView;

static EnvView view(const AntsCfg *c, OracleState *s, int e)
{
    EnvView v;
    size_t N = c->n_ants, G = (size_t)c->w * c->h, C = c->n_phero, R = c->n_rocks;
    v.N = c->n_ants; v.W = c->w; v.H = c->h; v.C = c->n_phero; v.R = c->n_rocks;
    v.x = s->x + e * N; v.y = s->y + e * N; v.theta = s->theta + e * N;
    v.prev_x = s->prev_x + e * N; v.prev_y = s->prev_y + e * N;
    v.holding = s->holding + e * N; v.seed = s->seed + e * N;
    v.activation = s->activation + e * N * C;
    v.mandibles = s->mandibles + e * N; v.reward_state = s->reward_state + e * N;
    v.phero = s->phero + e * C * G; v.food = s->food + e * G;
    v.walls = s->walls + e * G; v.area = s->anthill_area + e * G; v.explored = s->explored + e * G;
    v.xyr = s->anthill_xyr + 3 * e; v.anthill_food = s->anthill_food + e;
    v.rcx = s->rock_cx ? s->rock_cx + e * R : NULL; v.rcy = s->rock_cy ? s->rock_cy + e * R : NULL;
    v.rrad = s->rock_radius ? s->rock_radius + e * R : NULL;
    v.rwt = s->rock_weight ? s->rock_weight + e * R : NULL;
    v.timestep = s->timestep + e;
    v.prev_holding = s->prev_holding + e * N; v.prev_dist = s->prev_dist + e * N;
    v.primed = s->reward_primed + e;
    return v;
}

/* All_Rewards.compute_distance, rewards/reward_custom.py:59-60 */
static double anthill_dist(const EnvView *v, double x, double y)
{
    double dx = x - (double)v->xyr[0], dy = y - (double)v->xyr[1];
    return sqrt(dx * dx + dy * dy);
}

/* ---------------------------------------------------------------------------
 * reset: Anthill.__init__ (environment/anthill.py:17-33), Ants.__init__
 * (ants.py:18-41), RLApi.register_ants (RL_api.py:57-66), Reward.setup
 * (rewards/reward.py:12-19; reward_custom.py:13-15, 33-35, 65-77),
 * Environment.__init__ timestep = 1 (environment.py:27).
 * ------------------------------------------------------------------------- */
static void env_reset(const AntsCfg *c, EnvView *v)
{
    int N = v->N, W = v->W, H = v->H;
    /* anthill.py:28-33: area[x,y] = sqrt((ax-x)^2+(ay-y)^2) <= radius, integer ax,ay,r */
    long ax = v->xyr[0], ay = v->xyr[1], ar = v->xyr[2];
    for (long x = 0; x < W; ++x)
        for (long y = 0; y < H; ++y) {
            double d = sqrt((double)((ax - x) * (ax - x) + (ay - y) * (ay - y)));
            v->area[x * H + y] = d <= (double)ar;
        }
    *v->anthill_food = 0.0;
    for (int i = 0; i < N; ++i) {
        /* ants.py:27-30: copy, warp_xy, prev_ants = copy.  theta is NOT wrapped. */
        v->x[i] = warp_coord(v->x[i], (double)W);
        v->y[i] = warp_coord(v->y[i], (double)H);
        v->prev_x[i] = v->x[i]; v->prev_y[i] = v->y[i];
        v->mandibles[i] = 0; v->holding[i] = 0.0; v->reward_state[i] = 0;
        for (int k = 0; k < v->C; ++k) v->activation[i * v->C + k] = 0.0; /* ants.py:32,83 */
        v->prev_holding[i] = 0.0;
        v->prev_dist[i] = anthill_dist(v, v->x[i], v->y[i]); /* reward_custom.py:77 */
    }
    memset(v->explored, 0, (size_t)W * H); /* reward_custom.py:15,68 */
    *v->timestep = 1;
    *v->primed = 0;
    (void)c;
}

/* ---------------------------------------------------------------------------
 * RLApi.observation, environment/RL_api.py:96-165  (+ setup_perception :80-93,
 * reward.observation hooks rewards/reward_custom.py:17-22, 37-40, 79-106)
 * ------------------------------------------------------------------------- */
static void env_observe(const AntsCfg *c, EnvView *v, double *obs, double *agent_state,
                        double *reward, long *scratch_cells, uint8_t *ants_map)
{
    const int N = v->N, W = v->W, H = v->H, K = c->n_channels;
    const int r = c->perception_radius, P = 2 * r + 1, PP = P * P;

    /* RL_api.py:136-142: presence map from floor(xy); `+= 1` with repeated indices
     * does not accumulate, so the map is 0/1. */
    int need_ants = 0;
    for (int k = 0; k < K; ++k) need_ants |= c->channel_kind[k] == ANTSRL_CH_ANTS;
    if (need_ants) {
        memset(ants_map, 0, (size_t)W * H);
        for (int i = 0; i < N; ++i) {
            long ix = imod((long)v->x[i], W), iy = imod((long)v->y[i], H);
            ants_map[ix * H + iy] = 1;
        }
    }

    for (int i = 0; i < N; ++i) {
        /* RL_api.py:100-104 */
        double xf = v->x[i], yf = v->y[i];
        double tf = v->theta[i] + PI_D * 0.5;
        if (c->fwd_delta != 0.0) {
            xf += cos(v->theta[i]) * c->fwd_delta;
            yf += sin(v->theta[i]) * c->fwd_delta;
        }
        double ct = cos(tf), st = sin(tf); /* :107-108 */
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) {
                /* :92-93  coords[a][b] = (arange[b], arange[a]) * DELTA */
                double px = (double)(b - r) * c->delta, py = (double)(a - r) * c->delta;
                /* :110-111 rotation, :114 translation */
                double rx = ct * px - st * py;
                double ry = st * px + ct * py;
                double fx = rx + xf, fy = ry + yf;
                /* :117-119 np.round (half to even) -> int -> mod */
                long ix = imod((long)rint(fx), W), iy = imod((long)rint(fy), H);
                long cell = ix * H + iy;
                scratch_cells[(size_t)i * PP + a * P + b] = cell;
                if (!obs) continue;
                double m = c->has_mask ? (double)c->mask[a * P + b] : 1.0;
                double *o = obs + (((size_t)i * P + a) * P + b) * K;
                for (int k = 0; k < K; ++k) {
                    double p = 0.0;
                    switch (c->channel_kind[k]) {
                    case ANTSRL_CH_PHERO: /* :124-125 */
                        p = v->phero[(size_t)c->channel_arg[k] * W * H + cell] / c->phero_max_val;
                        break;
                    case ANTSRL_CH_FOOD: p = v->food[cell]; break;             /* :126-127 */
                    case ANTSRL_CH_WALLS: p = (double)v->walls[cell]; break;    /* :128-129 */
                    case ANTSRL_CH_ANTHILL: p = (double)v->area[cell]; break;   /* :130-131 */
                    case ANTSRL_CH_ROCKS: {                                     /* :132-135 */
                        int any = 0;
                        for (int q = 0; q < v->R; ++q) {
                            double vx = (double)ix - v->rcx[q], vy = (double)iy - v->rcy[q];
                            double d = sqrt(vx * vx + vy * vy);
                            any |= d < v->rrad[q];
                        }
                        p = (double)any;
                    } break;
                    case ANTSRL_CH_ANTS: p = (double)ants_map[cell]; break;     /* :142 */
                    default: break;
                    }
                    /* :147-148 perception = mask * (perception + 1) - 1 */
                    o[k] = c->has_mask ? m * (p + 1.0) - 1.0 : p;
                }
            }
        /* :160-162 */
        agent_state[2 * i + 0] = v->holding[i];
        agent_state[2 * i + 1] = v->seed[i];
    }

    /* ---- reward.observation(abs_coords, perception, agent_state), RL_api.py:164 ---- */
    const int kind = c->reward_kind;
    if (kind == ANTSRL_REWARD_NONE) { /* rewards/reward.py:19,27: zeros, no-op */
        for (int i = 0; i < N; ++i) reward[i] = 0.0;
        return;
    }
    /* Alias quirk: Food_Reward/All_Rewards.setup bind ants_holding to the LIVE
     * Ants.holding array (reward_custom.py:35,68), which Ants.update_mandibles mutates
     * in place (ants.py:117), so the first observation after setup always sees
     * delta-holding == 0; afterwards ants_holding is the previous agent_state column. */
    if (!*v->primed) {
        for (int i = 0; i < N; ++i) v->prev_holding[i] = v->holding[i];
        *v->primed = 1;
    }
    if (kind == ANTSRL_REWARD_EXPLORATION) {
        /* reward_custom.py:17-22: count unexplored among ALL P*P cells (mask ignored)
         * against the pre-observation map, /10, then mark. */
        for (int i = 0; i < N; ++i) {
            long cnt = 0;
            for (int q = 0; q < PP; ++q) cnt += 1 - v->explored[scratch_cells[(size_t)i * PP + q]];
            reward[i] = (double)cnt / 10.0;
        }
        for (size_t q = 0; q < (size_t)N * PP; ++q) v->explored[scratch_cells[q]] = 1;
    } else if (kind == ANTSRL_REWARD_FOOD) {
        /* reward_custom.py:37-40 */
        for (int i = 0; i < N; ++i) {
            double d = v->holding[i] - v->prev_holding[i];
            if (d < 0) d = 10.0;
            reward[i] = d;
            v->prev_holding[i] = v->holding[i];
        }
    } else { /* ANTSRL_REWARD_ALL, reward_custom.py:79-106 */
        int explore = c->fct_explore != 0.0 || c->fct_explore_holding != 0.0;
        for (int i = 0; i < N; ++i) {
            double rw = 0.0;                                   /* :80 */
            double dh = v->holding[i] - v->prev_holding[i];    /* :82 */
            double r_food = dh < 0 ? 0.0 : dh;                 /* :84 */
            double r_anthill = dh < 0 ? 1.0 : 0.0;             /* :97-98 (dh>0 -> 0, dh==0 stays 0) */
            v->prev_holding[i] = v->holding[i];                /* :85 */
            if (explore) {                                     /* :87-94 */
                long cnt = 0;
                for (int q = 0; q < PP; ++q) cnt += 1 - v->explored[scratch_cells[(size_t)i * PP + q]];
                double re = (double)cnt / 10.0;
                re = (v->holding[i] == 0.0) ? re * c->fct_explore : re * c->fct_explore_holding;
                rw += re;
            }
            double nd = anthill_dist(v, v->x[i], v->y[i]);     /* :102 */
            double heading = (double)((v->prev_dist[i] > nd) * (v->holding[i] > 0)) * 0.1; /* :103 */
            v->prev_dist[i] = nd;                              /* :104 */
            rw += r_food * c->fct_food + r_anthill * c->fct_anthill + heading * c->fct_headinganthill; /* :106 */
            reward[i] = rw;
        }
        if (explore)
            for (size_t q = 0; q < (size_t)N * PP; ++q) v->explored[scratch_cells[q]] = 1; /* :93 */
    }
}

/* ---------------------------------------------------------------------------
 * RLApi.step, environment/RL_api.py:168-204
 * ------------------------------------------------------------------------- */
static void env_step(const AntsCfg *c, EnvView *v, const int8_t *rot, const int8_t *ph, double *obs,
                     double *agent_state, double *reward, uint8_t *done, long *scratch_cells,
                     uint8_t *ants_map, double *tmp /* 3N */, uint8_t *newm /* N */)
{
    const int N = v->N, W = v->W, H = v->H, K = c->n_channels;
    (void)W;
    /* :178-185 mandible target, iterating perceived_objects in order */
    for (int i = 0; i < N; ++i) {
        long cprev = (long)v->prev_x[i] * H + (long)v->prev_y[i];
        int m = v->mandibles[i];
        for (int k = 0; k < K; ++k) {
            if (c->channel_kind[k] == ANTSRL_CH_FOOD)
                m = (v->food[cprev] > 0) | m;                                    /* :182 */
            else if (c->channel_kind[k] == ANTSRL_CH_ANTHILL)
                m = (1 - v->area[(long)v->x[i] * H + (long)v->y[i]]) & m;        /* :184 */
        }
        newm[i] = (uint8_t)m;
    }
    /* Ants.update_mandibles, environment/ants.py:102-117 */
    double *q_old = tmp, *taken = tmp + N, *dropped = tmp + 2 * N;
    for (int i = 0; i < N; ++i) {
        int closing = newm[i] & (1 - v->mandibles[i]);  /* :103 */
        int opening = (1 - newm[i]) & v->mandibles[i];  /* :104 */
        long cprev = (long)v->prev_x[i] * H + (long)v->prev_y[i];
        q_old[i] = v->food[cprev];
        taken[i] = fmin(c->max_hold, fmax(0.0, q_old[i])) * (double)closing; /* :111 */
        dropped[i] = v->holding[i] * (double)opening;                         /* :114 */
    }
    for (int i = 0; i < N; ++i) {
        long cprev = (long)v->prev_x[i] * H + (long)v->prev_y[i];
        /* :116 fancy-index `+=`: every ant adds to the PRE-update value and the last
         * ant (in index order) on a cell wins. */
        v->food[cprev] = q_old[i] + (dropped[i] - taken[i]);
        v->holding[i] += taken[i] - dropped[i];               /* :117 */
        v->mandibles[i] = newm[i];                            /* :107 */
    }
    /* :187-188 Ants.activate_pheromone, ants.py:89-96 (hard-codes two channels) */
    if (ph) {
        for (int i = 0; i < N; ++i) {
            double a0 = 0.0, a1 = 0.0;
            if (ph[i] == 1) a0 = c->deposit_strength;
            else if (ph[i] != 0) a1 = c->deposit_strength;
            v->activation[i * v->C + 0] = a0;
            if (v->C > 1) v->activation[i * v->C + 1] = a1;
        }
    }
    /* :190-191 Ants.rotate_ants, ants.py:65-67 (+ warp_theta :62-63) */
    if (rot)
        for (int i = 0; i < N; ++i)
            v->theta[i] = np_mod(v->theta[i] + (double)rot[i] * c->max_rot_speed, 2 * PI_D);
    /* :194-196 forward move, Ants.forward_ants ants.py:77-80, translate/warp :69-75 */
    for (int i = 0; i < N; ++i) {
        double fwd = 1.0 * c->max_speed * (1 - v->holding[i] * c->carry_speed_reduction);
        if (fwd < 0) fwd *= c->backward_speed_reduction;
        double ax = cos(v->theta[i]) * fwd, ay = sin(v->theta[i]) * fwd;
        v->x[i] = warp_coord(v->x[i] + ax, (double)v->W);
        v->y[i] = warp_coord(v->y[i] + ay, (double)v->H);
    }
    /* :198 */
    env_observe(c, v, obs, agent_state, reward, scratch_cells, ants_map);
    /* :200 */
    *done = (uint8_t)(c->max_time == *v->timestep);
    /* :203 Ants.give_reward, ants.py:119-121 */
    for (int i = 0; i < N; ++i)
        if (reward[i] - c->reward_threshold > 0) v->reward_state[i] = 255;
}

/* ---------------------------------------------------------------------------
 * Environment.update, environment/environment.py:42-47
 * ------------------------------------------------------------------------- */
static int env_update(const AntsCfg *c, EnvView *v, int env_index, const double *jitter,
                      double *grid_tmp /* W*H */)
{
    const int N = v->N, W = v->W, H = v->H, C = v->C, R = v->R;
    const size_t G = (size_t)W * H;
    *v->timestep += 1; /* :45 */

    /* --- Walls.update (step -1), environment/walls.py:22-30 --- */
    int hits = 0;
    for (int i = 0; i < N; ++i) {
        long cell = (long)v->x[i] * H + (long)v->y[i];   /* :25 */
        if (v->walls[cell]) {                             /* :26 */
            v->x[i] = v->prev_x[i]; v->y[i] = v->prev_y[i]; /* :27 */
            double u = jitter ? jitter[hits]
                              : oracle_jitter_u01(c->rng_seed, (uint32_t)c->env_id_base + (uint32_t)env_index, /* the env's GLOBAL id */
                                                  (uint32_t)*v->timestep, (uint32_t)i);
            v->theta[i] += u - 0.5;                       /* :28 (theta not re-wrapped) */
            ++hits;
        }
    }
    for (int k = 0; k < C; ++k)                           /* :30 */
        for (size_t g = 0; g < G; ++g)
            if (v->walls[g]) v->phero[k * G + g] = 0.0;

    /* --- CircleObstacles.update (step 0), environment/circle_obstacles.py:32-58 --- */
    if (R > 0) {
        for (int q = 0; q < R; ++q) { /* pass 1 :35-40, sum over ants in index order */
            double sx = 0.0, sy = 0.0;
            for (int i = 0; i < N; ++i) {
                double vx = v->rcx[q] - v->x[i], vy = v->rcy[q] - v->y[i];
                double d = sqrt(vx * vx + vy * vy);
                double f = 1 - v->rrad[q] / (d + 0.001);
                double px = vx * f, py = vy * f;
                if (d > v->rrad[q]) { px = 0.0; py = 0.0; }
                sx += px; sy += py;
            }
            v->rcx[q] -= sx / v->rwt[q];
            v->rcy[q] -= sy / v->rwt[q];
        }
        for (int i = 0; i < N; ++i) { /* pass 2 :53-58 with the UPDATED centres */
            double sx = 0.0, sy = 0.0;
            for (int q = 0; q < R; ++q) {
                double vx = v->rcx[q] - v->x[i], vy = v->rcy[q] - v->y[i];
                double d = sqrt(vx * vx + vy * vy);
                double f = 1 - v->rrad[q] / (d + 0.001);
                double px = vx * f, py = vy * f;
                if (d > v->rrad[q]) { px = 0.0; py = 0.0; }
                sx += px; sy += py;
            }
            v->x[i] = warp_coord(v->x[i] + sx, (double)W); /* translate_ants, ants.py:73-75 */
            v->y[i] = warp_coord(v->y[i] + sy, (double)H);
        }
    }

    /* --- Pheromone.update (step 0), environment/pheromone.py:43-45 --- */
    {
        const int fr = c->filter_radius, fs = 2 * fr + 1;
        for (int k = 0; k < C; ++k) {
            double *p = v->phero + k * G;
            if (fr == 0) {
                for (size_t g = 0; g < G; ++g) {
                    double o = p[g] * c->filter[0];
                    p[g] = o < c->phero_threshold ? 0.0 : o;
                }
            } else {
                /* convolve2d(phero, F, 'same', 'fill', 0):
                 * out[x,y] = sum_{a,b} F[a,b] * in[x - a + fr, y - b + fr], zero outside */
                for (int x = 0; x < W; ++x)
                    for (int y = 0; y < H; ++y) {
                        double acc = 0.0;
                        for (int a = 0; a < fs; ++a) {
                            int sx = x - a + fr;
                            if (sx < 0 || sx >= W) continue;
                            for (int b = 0; b < fs; ++b) {
                                int sy = y - b + fr;
                                if (sy < 0 || sy >= H) continue;
                                acc += c->filter[a * fs + b] * p[(size_t)sx * H + sy];
                            }
                        }
                        grid_tmp[(size_t)x * H + y] = acc < c->phero_threshold ? 0.0 : acc;
                    }
                memcpy(p, grid_tmp, G * sizeof(double));
            }
        }
    }

    /* --- Ants.update (step 999), environment/ants.py:123-130 --- */
    for (int i = 0; i < N; ++i) { v->prev_x[i] = v->x[i]; v->prev_y[i] = v->y[i]; } /* :124 */
    for (int k = 0; k < C; ++k) {
        /* Pheromone.add_pheromones, pheromone.py:36-41: last writer wins, then
         * whole-grid minimum with max_val. */
        double *p = v->phero + k * G;
        for (int i = 0; i < N; ++i) grid_tmp[i] = p[(long)v->x[i] * H + (long)v->y[i]];
        for (int i = 0; i < N; ++i)
            p[(long)v->x[i] * H + (long)v->y[i]] = grid_tmp[i] + v->activation[i * C + k];
        if (c->has_max_val && N > 0)
            for (size_t g = 0; g < G; ++g) p[g] = fmin(p[g], c->phero_max_val);
    }
    for (int i = 0; i < N; ++i) /* :130 */
        v->reward_state[i] = (uint8_t)((double)v->reward_state[i] * 0.9);

    /* --- Anthill.update (step 1000), environment/anthill.py:41-46 --- */
    {
        double gain_sum = 0.0;
        for (size_t g = 0; g < G; ++g) {
            double gain = v->food[g] * (double)v->area[g];
            v->food[g] -= gain;
            gain_sum += gain;
        }
        *v->anthill_food += gain_sum;
    }
    return hits;
}

/* ---- batch wrappers -------------------------------------------------------- */
void oracle_reset(const AntsCfg *c, OracleState *s)
{
    for (int e = 0; e < c->n_envs; ++e) {
        EnvView v = view(c, s, e);
        env_reset(c, &v);
    }
}

void oracle_observe(const AntsCfg *c, OracleState *s, double *obs, double *agent_state,
                    double *reward, int n_threads)
{
    const size_t N = c->n_ants, P = 2 * c->perception_radius + 1, K = c->n_channels;
    const size_t G = (size_t)c->w * c->h;
    (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        long *cells = (long *)malloc(sizeof(long) * N * P * P);
        uint8_t *amap = (uint8_t *)malloc(G);
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < c->n_envs; ++e) {
            EnvView v = view(c, s, e);
            env_observe(c, &v, obs ? obs + e * N * P * P * K : NULL, agent_state + e * N * 2,
                        reward + e * N, cells, amap);
        }
        free(cells); free(amap);
    }
}

void oracle_step(const AntsCfg *c, OracleState *s, const int8_t *rotation, const int8_t *phero,
                 double *obs, double *agent_state, double *reward, uint8_t *done, int n_threads)
{
    const size_t N = c->n_ants, P = 2 * c->perception_radius + 1, K = c->n_channels;
    const size_t G = (size_t)c->w * c->h;
    (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        long *cells = (long *)malloc(sizeof(long) * N * P * P);
        uint8_t *amap = (uint8_t *)malloc(G);
        double *tmp = (double *)malloc(sizeof(double) * 3 * N);
        uint8_t *newm = (uint8_t *)malloc(N);
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < c->n_envs; ++e) {
            EnvView v = view(c, s, e);
            env_step(c, &v, rotation ? rotation + e * N : NULL, phero ? phero + e * N : NULL,
                     obs ? obs + e * N * P * P * K : NULL, agent_state + e * N * 2, reward + e * N,
                     done + e, cells, amap, tmp, newm);
        }
        free(cells); free(amap); free(tmp); free(newm);
    }
}

void oracle_update(const AntsCfg *c, OracleState *s, const double *wall_jitter, int32_t *hit_count,
                   int n_threads)
{
    const size_t N = c->n_ants, G = (size_t)c->w * c->h;
    (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        double *gt = (double *)malloc(sizeof(double) * (G > N ? G : N));
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < c->n_envs; ++e) {
            EnvView v = view(c, s, e);
            int hits = env_update(c, &v, e, wall_jitter ? wall_jitter + e * N : NULL, gt);
            if (hit_count) hit_count[e] = hits;
        }
        free(gt);
    }
}
