// antsrl_state.hip — reset, device-side episode generator, activation, state read-out (not on the hot
// path), launchers.
#include "antsrl_util.h"

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
        const size_t ri = e * G + prec_cell(p, (uint32_t)g), fi = e * G + frec_cell(p, (uint32_t)g); // the cell's records (canonical inputs are row-major)
        p.s.food[fi * p.fs] = food[i];
        if (p.meta) // wall / anthill bits (k_reset_bits ran before on this stream), no ant, never explored
            reinterpret_cast<uint32_t *>(p.s.food)[fi * p.fs + 1] =
                (test_bit(p.s.walls_bits + e * p.words, (uint32_t)g) ? META_WALL : 0u) |
                (test_bit(p.s.area_bits + e * p.words, (uint32_t)g) ? META_AREA : 0u) | (META_NEVER << META_STAMP_SHIFT);
        for (int c = 0; c < p.C; ++c) {
            const float v = phero ? phero[(e * p.C + c) * G + g] : 0.0f;
            p.s.phero[0][ri * p.ps + c] = v;
            p.s.phero[1][ri * p.ps + c] = v;
        }
    }
}

// ---- device-side episode generator (antsrl_generate) ----------------------
// Draw `idx` of stream `tag` of environment `env`; the oracle (oracle_gen_u01) is bit-identical.
#define GEN_SALT 0x6A09E667F3BCC909ULL
enum { GEN_ANTHILL = 0, GEN_WALLS = 1, GEN_FOOD = 2, GEN_ROCKS = 3, GEN_ANT_ANGLE = 4, GEN_ANT_DIST = 5,
       GEN_ANT_THETA = 6, GEN_ANT_SEED = 7, GEN_WALL_OFFSET = 8 };
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
    const uint32_t ge = p.env_id_base + (uint32_t)e; // the env's GLOBAL id keys its streams (AntsCfg.env_id_base)
    p.s.anthill_xyr[3 * e + 0] = (int)(gen_u01(seed, ge, GEN_ANTHILL, 0) * W * 0.5 + W * 0.25);
    p.s.anthill_xyr[3 * e + 1] = (int)(gen_u01(seed, ge, GEN_ANTHILL, 1) * H * 0.5 + H * 0.25);
    p.s.anthill_xyr[3 * e + 2] = (int)(gen_u01(seed, ge, GEN_ANTHILL, 2) * m * 0.05 + m * 0.05);
    p.s.anthill_food[e] = 0.0;
    p.s.timestep[e] = 1; // environment.py:27
    p.s.reward_primed[e] = 0;
    for (int q = 0; q < p.R; ++q) {
        p.s.rock_cx[(size_t)e * p.R + q] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 0) * (W * 0.75) + W * 0.25;
        p.s.rock_cy[(size_t)e * p.R + q] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 1) * (H * 0.25) + H * 0.25;
        p.s.rock_r[(size_t)e * p.R + q] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 2) * 5 + 5;
        p.s.rock_w[(size_t)e * p.R + q] = gen_u01(seed, ge, GEN_ROCKS, 4 * q + 3) * 50 + 50;
    }
    for (int d = 0; d < g.n_food_discs; ++d) {
        int rad = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 0) * (g.food_rmax - g.food_rmin) + g.food_rmin);
        const int cap = (m - 1) / 2;
        rad = rad > cap ? cap : rad;
        int32_t *dd = p.s.gen_discs + ((size_t)e * ANTSRL_MAX_FOOD_DISCS + d) * 3;
        dd[0] = rad;
        dd[1] = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 1) * (W - 2 * rad) + rad);
        dd[2] = (int)(gen_u01(seed, ge, GEN_FOOD, 3 * d + 2) * (H - 2 * rad) + rad);
    }
}

// ---- PerlinGenerator (generator/map_generators.py:9-25 over utils.py:7-17): 2-D improved Perlin noise
// (Perlin 2002: permutation table, quintic fade, 16 gradient directions; octaves summed as total / max),
// float32 like the `noise` package the reference calls.  antsrl_amd/generator.py (perlin_noise) and the
// oracle hold the same arithmetic; all three agree bit for bit (-ffp-contract=off).
__device__ const uint8_t PERLIN_PERM[256] = {
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
__device__ const int8_t PERLIN_GRAD[16][2] = {{1, 1}, {-1, 1}, {1, -1}, {-1, -1}, {1, 0}, {-1, 0}, {1, 0}, {-1, 0},
                                              {0, 1}, {0, -1}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {0, -1}, {0, 1}};

__device__ __forceinline__ float perlin_grad(int hash, float gx, float gy)
{
    const int h = PERLIN_PERM[hash & 255] & 15;
    return gx * (float)PERLIN_GRAD[h][0] + gy * (float)PERLIN_GRAD[h][1];
}
__device__ __forceinline__ float perlin_lerp(float t, float a, float b) { return a + t * (b - a); }

__device__ float perlin_noise2(float x, float y, float rx, float ry)
{
    long i = (long)floorf(fmodf(x, rx)), j = (long)floorf(fmodf(y, ry));
    long ii = (long)fmodf((float)(i + 1), rx), jj = (long)fmodf((float)(j + 1), ry);
    i &= 255; j &= 255; ii &= 255; jj &= 255;
    x = x - floorf(x);
    y = y - floorf(y);
    const float fx = x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f);
    const float fy = y * y * y * (y * (y * 6.0f - 15.0f) + 10.0f);
    const int A = PERLIN_PERM[i], B = PERLIN_PERM[ii];
    const int AA = PERLIN_PERM[(A + j) & 255], AB = PERLIN_PERM[(A + jj) & 255];
    const int BA = PERLIN_PERM[(B + j) & 255], BB = PERLIN_PERM[(B + jj) & 255];
    return perlin_lerp(fy, perlin_lerp(fx, perlin_grad(AA, x, y), perlin_grad(BA, x - 1.0f, y)),
                       perlin_lerp(fx, perlin_grad(AB, x, y - 1.0f), perlin_grad(BB, x - 1.0f, y - 1.0f)));
}

// pnoise2((cx + ox) / scale, (cy + oy) / scale, octaves, persistence, lacunarity), utils.py:11-16
__device__ double perlin_at(long cx, long cy, long ox, long oy, const AntsGen &g)
{
    const float x = (float)((double)(cx + ox) / g.perlin_scale), y = (float)((double)(cy + oy) / g.perlin_scale);
    if (g.perlin_octaves == 1) return (double)perlin_noise2(x, y, 1024.0f, 1024.0f);
    float freq = 1.0f, amp = 1.0f, mx = 0.0f, total = 0.0f;
    for (int o = 0; o < g.perlin_octaves; ++o) {
        const float rep = (float)(1024.0 * (double)freq);
        total = total + perlin_noise2(x * freq, y * freq, rep, rep) * amp;
        mx = mx + amp;
        freq = freq * (float)g.perlin_lacunarity;
        amp = amp * (float)g.perlin_persistence;
    }
    return (double)(total / mx);
}


// ---- ANTSRL_RNG_REFERENCE: the reference's own random streams -----------------------------------------------
// EnvironmentGenerator.generate seeds Python's `random` with `seed` and numpy's legacy generator with `seed * 5`
// (environment_generator.py:53-55): two MT19937 generators (Matsumoto & Nishimura 1998, as published: state of
// 624 words, twist with 0x9908b0df, tempering 11 / 7 & 0x9d2c5680 / 15 & 0xefc60000 / 18) that differ in their
// seeding — Python: init_by_array over the 32-bit limbs of the integer, numpy: init_genrand(seed) — and share the
// 53-bit double (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53.  One wave per environment; lane 0 advances the
// generators (the recurrence is sequential), all lanes turn a block's outputs into doubles and scatter them.
#define MT_N 624
#define MT_M 397
__device__ __forceinline__ void mt_init_genrand(uint32_t *mt, uint32_t s)
{
    mt[0] = s;
    for (int i = 1; i < MT_N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}
__device__ __forceinline__ void mt_init_by_array(uint32_t *mt, const uint32_t *key, int len)
{
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    for (int k = MT_N > len ? MT_N : len; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (int k = MT_N - 1; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
}
// next block of 624 outputs: twist in place, then tempered values into out[]
__device__ __forceinline__ void mt_next_block(uint32_t *mt, uint32_t *out)
{
    for (int k = 0; k < MT_N; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % MT_N] & 0x7fffffffu);
        mt[k] = mt[(k + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (int k = 0; k < MT_N; ++k) {
        uint32_t y = mt[k];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[k] = y;
    }
}
__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// per environment: everything EnvironmentGenerator.generate draws (environment_generator.py:52-94)
__global__ void __launch_bounds__(64) k_gen_mt(const KP p, const AntsGen g, const uint64_t seed0)
{
    __shared__ uint32_t mt[MT_N], out[MT_N];
    __shared__ int s_ar;
    const int e = blockIdx.x, lane = threadIdx.x;
    const int W = p.W, H = p.H, N = p.N, R = p.R, m = W < H ? W : H;
    const uint64_t seed = seed0 + (uint64_t)p.env_id_base + (uint64_t)e; // EnvironmentGenerator(seed = episode_seed + GLOBAL env id)
    const size_t eN = (size_t)e * N;

    // ---- Python stream: anthill (:60-63), PerlinGenerator's offsets (map_generators.py:19-20), food circles (:37-39)
    if (lane == 0) {
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        mt_init_by_array(mt, key, key[1] ? 2 : 1); // random.seed(int): the limbs of abs(seed), least significant first
        mt_next_block(mt, out);
        int pos = 0;
        auto next32 = [&]() -> uint32_t {
            if (pos == MT_N) { mt_next_block(mt, out); pos = 0; }
            return out[pos++];
        };
        auto rnd = [&]() -> double { const uint32_t a = next32(), b = next32(); return mt_double(a, b); };
        p.s.anthill_xyr[3 * e + 0] = (int)(rnd() * W * 0.5 + W * 0.25);
        p.s.anthill_xyr[3 * e + 1] = (int)(rnd() * H * 0.5 + H * 0.25);
        const int ar = (int)(rnd() * m * 0.05 + m * 0.05);
        p.s.anthill_xyr[3 * e + 2] = ar;
        s_ar = ar;
        if (g.wall_kind == ANTSRL_WALLS_PERLIN)
            for (int k = 0; k < 2; ++k) { // random.randint(-10000, 10000) = -10000 + _randbelow(20001): 15-bit draws, rejection
                uint32_t r;
                do r = next32() >> 17; while (r >= 20001u);
                p.s.gen_perlin[2 * e + k] = (int)r - 10000;
            }
        for (int d = 0; d < g.n_food_discs; ++d) {
            const int rad = (int)(rnd() * (g.food_rmax - g.food_rmin) + g.food_rmin);
            int32_t *dd = p.s.gen_discs + ((size_t)e * ANTSRL_MAX_FOOD_DISCS + d) * 3;
            dd[0] = rad;
            dd[1] = (int)(rnd() * (W - 2 * rad) + rad);
            dd[2] = (int)(rnd() * (H - 2 * rad) + rad);
        }
        p.s.anthill_food[e] = 0.0;
        p.s.timestep[e] = 1; // environment.py:27
        p.s.reward_primed[e] = 0;
        mt_init_genrand(mt, (uint32_t)(seed * 5u)); // np.random.seed(seed * 5)
    }
    __syncthreads();
    // ---- numpy stream, doubles in draw order: rock centres [R][2], radiuses [R], weights [R] (:77-85, the evident
    // intent of the broken branch), then ants_angle [N], ants_dist [N], ants_t [N] (:87-91), Ants.seed [N] (ants.py:41)
    const int n_dbl = 4 * R + 4 * N;
    for (int base = 0; base < n_dbl; base += MT_N / 2) {
        if (lane == 0) mt_next_block(mt, out);
        __syncthreads();
        for (int j = lane; j < MT_N / 2 && base + j < n_dbl; j += 64) {
            const double u = mt_double(out[2 * j], out[2 * j + 1]);
            int d = base + j;
            if (d < 2 * R) {
                const int r = d >> 1;
                if (d & 1) p.s.rock_cy[(size_t)e * R + r] = u * (H * 0.25) + H * 0.25;
                else p.s.rock_cx[(size_t)e * R + r] = u * (W * 0.75) + W * 0.25;
                continue;
            }
            d -= 2 * R;
            if (d < R) { p.s.rock_r[(size_t)e * R + d] = u * 5 + 5; continue; }
            d -= R;
            if (d < R) { p.s.rock_w[(size_t)e * R + d] = u * 50 + 50; continue; }
            d -= R;
            const int arr = d / N, a = d - arr * N;
            if (arr == 0) p.s.prev_x[eN + a] = u * 2 * PI_D;                  // ants_angle (scratch until the ants are placed)
            else if (arr == 1) p.s.prev_y[eN + a] = u * (double)s_ar * 0.8;   // ants_dist
            else if (arr == 2) p.s.theta[eN + a] = u * 2 * PI_D;              // ants_t
            else p.s.seed[eN + a] = (float)u;                                 // Ants.seed
        }
        __syncthreads();
    }
    // ---- ants around the anthill (:89-90), then Ants.__init__ (ants.py:18-41)
    const double ax = (double)p.s.anthill_xyr[3 * e + 0], ay = (double)p.s.anthill_xyr[3 * e + 1];
    for (int a = lane; a < N; a += 64) {
        const size_t i = eN + a;
        const double ang = p.s.prev_x[i], dist = p.s.prev_y[i];
        const double x = warp_coord(cos(ang) * dist + ax, (double)W);
        const double y = warp_coord(sin(ang) * dist + ay, (double)H);
        p.s.x[i] = x; p.s.y[i] = y;
        p.s.prev_x[i] = x; p.s.prev_y[i] = y;
        p.s.holding[i] = 0.0f; p.s.prev_holding[i] = 0.0f;
        p.s.mandibles[i] = 0; p.s.reward_state[i] = 0;
        p.s.dirty_cell[i] = -1;
        p.s.walldep_cell[i] = -1;
        for (int c = 0; c < p.C; ++c) p.s.activation[i * p.C + c] = 0.0f;
        const double dx = x - ax, dy = y - ay;
        p.s.prev_dist[i] = sqrt(dx * dx + dy * dy); // reward_custom.py:77
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
        // PerlinGenerator.generate: random.randint(-10000, 10000) twice (map_generators.py:19-20); with the
        // reference's streams k_gen_mt drew them from the environment's Python generator
        const bool ref_rng = g.rng_kind == ANTSRL_RNG_REFERENCE;
        const long pox = ref_rng ? (long)p.s.gen_perlin[2 * e + 0] : (long)(gen_u01(seed, p.env_id_base + (uint32_t)e, GEN_WALL_OFFSET, 0) * 20001.0) - 10000;
        const long poy = ref_rng ? (long)p.s.gen_perlin[2 * e + 1] : (long)(gen_u01(seed, p.env_id_base + (uint32_t)e, GEN_WALL_OFFSET, 1) * 20001.0) - 10000;
        uint32_t wb = 0, ab = 0;
        for (int b = 0; b < 32; ++b) {
            const size_t cell = w * 32 + b;
            if (cell >= G) break;
            const long x = (long)(cell / p.H), y = (long)(cell % p.H);
            const bool area = ar >= 0 && (ax - x) * (ax - x) + (ay - y) * (ay - y) <= ar * ar;
            const bool wall = !area && (g.wall_kind == ANTSRL_WALLS_PERLIN  ? perlin_at(x, y, pox, poy, g) > g.wall_density
                                        : g.wall_kind == ANTSRL_WALLS_INPUT ? g.walls_input[e * G + cell] != 0
                                        : g.wall_density > 0.0 && gen_u01(seed, p.env_id_base + (uint32_t)e, GEN_WALLS, (uint32_t)cell) < g.wall_density);
            bool fd = false;
            for (int d = 0; d < g.n_food_discs; ++d) {
                const long rad = discs[3 * d], dx = discs[3 * d + 1] - x, dy = discs[3 * d + 2] - y;
                fd |= dx * dx + dy * dy <= rad * rad;
            }
            if (area) ab |= 1u << b;
            if (wall) wb |= 1u << b;
            const size_t ri = e * G + prec_xy(p, (int)x, (int)y), fi = e * G + frec_xy(p, (int)x, (int)y);
            p.s.food[fi * p.fs] = (fd && !wall) ? 1.0f : 0.0f;
            if (p.meta)
                reinterpret_cast<uint32_t *>(p.s.food)[fi * p.fs + 1] =
                    (wall ? META_WALL : 0u) | (area ? META_AREA : 0u) | (META_NEVER << META_STAMP_SHIFT);
            for (int c = 0; c < p.C; ++c) {
                p.s.phero[0][ri * p.ps + c] = 0.0f;
                p.s.phero[1][ri * p.ps + c] = 0.0f;
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
        const uint32_t ge = p.env_id_base + e;
        const double ang = gen_u01(seed, ge, GEN_ANT_ANGLE, a) * 2 * PI_D;
        const double dist = gen_u01(seed, ge, GEN_ANT_DIST, a) * ar * 0.8;
        const double x = warp_coord(cos(ang) * dist + ax, (double)p.W);
        const double y = warp_coord(sin(ang) * dist + ay, (double)p.H);
        p.s.x[i] = x; p.s.y[i] = y;
        p.s.theta[i] = gen_u01(seed, ge, GEN_ANT_THETA, a) * 2 * PI_D;
        p.s.prev_x[i] = x; p.s.prev_y[i] = y;
        p.s.holding[i] = 0.0f; p.s.prev_holding[i] = 0.0f;
        p.s.mandibles[i] = 0; p.s.reward_state[i] = 0;
        p.s.seed[i] = (float)gen_u01(seed, ge, GEN_ANT_SEED, a);
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
            float v = p.s.phero[cur][(e * G + prec_cell(p, (uint32_t)g)) * p.ps + c];
            if (p.scaled) {
                v *= (float)p.g_now;
                if (v < (float)p.threshold) v = 0.0f;
            }
            ((float *)dstv)[i] = v;
        }
        break;
    case ANTSRL_S_PHERO_C0: case ANTSRL_S_PHERO_C1: case ANTSRL_S_PHERO_C2: case ANTSRL_S_PHERO_C3: {
        const int c = which - ANTSRL_S_PHERO_C0; // one channel [E][W][H]
        for (size_t i = t0; i < EG; i += stride) {
            const size_t e = i / G;
            float v = p.s.phero[cur][(e * G + prec_cell(p, (uint32_t)(i - e * G))) * p.ps + c];
            if (p.scaled) {
                v *= (float)p.g_now;
                if (v < (float)p.threshold) v = 0.0f;
            }
            ((float *)dstv)[i] = v;
        }
    } break;
    case ANTSRL_S_FOOD:
        for (size_t i = t0; i < EG; i += stride) {
            const size_t e = i / G;
            ((float *)dstv)[i] = p.s.food[(e * G + frec_cell(p, (uint32_t)(i - e * G))) * p.fs];
        }
        break;
    case ANTSRL_S_EXPLORED:
    case ANTSRL_S_WALLS:
    case ANTSRL_S_ANTHILL_AREA: {
        const uint32_t *bits = which == ANTSRL_S_EXPLORED ? p.s.explored_bits
                               : which == ANTSRL_S_WALLS  ? p.s.walls_bits : p.s.area_bits;
        if (which == ANTSRL_S_EXPLORED && p.meta) { // cell-meta layout: explored <=> the cell carries a stamp
            const uint32_t *m = reinterpret_cast<const uint32_t *>(p.s.food) + 1;
            for (size_t i = t0; i < EG; i += stride) {
                const size_t e = i / G;
                ((uint8_t *)dstv)[i] = (uint8_t)((m[(e * G + frec_cell(p, (uint32_t)(i - e * G))) * p.fs] >> META_STAMP_SHIFT) != META_NEVER);
            }
            break;
        }
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

// Measurement helper (antsrl_bench_copy): 16 bytes per lane.  Every 256-thread workgroup copies ONE contiguous
// 16 KiB chunk (four independent 16-byte loads per lane in flight, then four stores) and workgroups take the
// chunks in address order: the chip then reads and writes a compact, advancing window, which is what the memory
// side rewards (profiles/history/fill_clone_probe.hip: 6.1-6.8 TB/s of writes with 4-16 KiB per workgroup against 4.9-5.3
// with 64 KiB and more).
__global__ void __launch_bounds__(256) k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, const size_t n)
{
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(base + (size_t)u * 256, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + (size_t)u * 256 < n) dst[base + (size_t)u * 256] = v[u];
}

hipError_t antsrl_launch_copy16(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    const size_t n = bytes / 16;
    if (n == 0) return hipSuccess;
    const size_t blocks = (n + 1023) / 1024;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)blocks), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)src, n);
    return hipGetLastError();
}

// host-side launchers (called from antsrl_capi.hip)
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
    if (g.rng_kind == ANTSRL_RNG_REFERENCE) { // the reference's MT19937 streams: one wave per environment
        hipLaunchKernelGGL(k_gen_mt, dim3(p.E), dim3(64), 0, st, p, g, seed);
        hipLaunchKernelGGL(k_gen_cells, dim3(grid_for((size_t)p.E * p.words)), dim3(256), 0, st, p, g, seed);
        return hipGetLastError();
    }
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

// RLApi.perceptive_field (RL_api.py:144-153, save_perceptive_field — the viewer's overlay): dst[e][x][y] = 1 where some ant's
// perception reaches (masked cells not counted), from the ants' positions as they stand — the caller asks right behind the
// observation.  One thread per (ant, perceived cell); the geometry is k_perceive's (centre shifted by fwd_delta along the
// heading, offsets rotated by theta + pi / 2, half-to-even rounding, toroidal wrap: RL_api.py:100-119).  Not on the hot path.
__global__ void __launch_bounds__(256) k_perceptive_field(const KP p, uint8_t *__restrict__ dst)
{
    const size_t n = (size_t)p.E * p.N * p.PP;
    const size_t G = (size_t)p.W * p.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t a = i / p.PP;
        const int q = (int)(i - a * p.PP);
        if (p.has_mask && !p.mask[q]) continue;
        const size_t e = a / p.N;
        const double th = p.s.theta[a];
        double sn, cs, sn2, cs2;
        sincos(th + PI_D * 0.5, &sn2, &cs2);
        double cx = p.s.x[a], cy = p.s.y[a];
        if (p.fwd_delta != 0.0) {
            sincos(th, &sn, &cs);
            cx += cs * p.fwd_delta;
            cy += sn * p.fwd_delta;
        }
        const double ox = (double)(q % p.P - p.r) * p.delta, oy = (double)(q / p.P - p.r) * p.delta;
        const int ix = wrap_index((int)rint((cs2 * ox - sn2 * oy) + cx), p.W);
        const int iy = wrap_index((int)rint((sn2 * ox + cs2 * oy) + cy), p.H);
        dst[e * G + (size_t)ix * p.H + iy] = 1;
    }
}

hipError_t antsrl_launch_perceptive_field(const KP &p, uint8_t *dst, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)p.E * p.W * p.H, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_perceptive_field, dim3(grid_for((size_t)p.E * p.N * p.PP)), dim3(256), 0, st, p, dst);
    return hipGetLastError();
}

hipError_t antsrl_launch_read_state(const KP &p, int which, int cur, void *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_read_state, dim3(grid_for((size_t)p.E * p.W * p.H)), dim3(256), 0, st, p, which, cur, dst);
    return hipGetLastError();
}
