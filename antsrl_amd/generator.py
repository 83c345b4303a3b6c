"""Episode "reset": host-side counterpart of generator/environment_generator.py and
generator/map_generators.py.

EnvironmentGenerator.generate(rl_api) draws the initial state with the SAME calls, in the same
order, from the same global `random` / `np.random` streams as the reference
(environment_generator.py:52-106), so an equal seed gives an identical initial state
(tests/test_generator.py pins this against the golden fixtures).  It then uploads the state
once (antsrl_reset) and wires the RLApi / Environment views.

Extensions over the reference (none changes its behaviour for the reference's arguments):
  * n_envs > 1 builds a batch; env e is drawn with seed + env_id_base + e (seed=None: fresh randomness each);
  * n_rocks > 0 works.  The reference's rock branch raises NameError (`n_rocks` undefined at
    environment_generator.py:83-84); this follows its evident intent (self.n_rocks).
PerlinGenerator (the walls of main.py:75) thresholds 2-D improved Perlin noise.  The reference gets the
noise from the third-party `noise` package (utils.py:7-17, `noise.pnoise2`), which is absent here and
not under /root/reference: `perlin_noise` below restates the published algorithm (Perlin 2002:
permutation table, quintic fade, 16-entry gradient table, octaves summed as total/max) in float32.
It could not be checked against the package, so equal seeds give the same KIND of cave map, not a
pinned bit-identical one ("parity unpinned" for this generator; walls are an input bitmap to the
path, any object with .generate(w, h) -> bool[w, h] works).
"""
from __future__ import annotations

import random

import numpy as np

from . import config as cm
from .rl_api import (Ants, Anthill, CircleObstacles, Environment, Food, Pheromone, RLApi, Walls)

PHERO_COLORS = [(255, 64, 0), (64, 64, 255), (100, 255, 100)]  # environment_generator.py:12-16


class CirclesGenerator:  # generator/map_generators.py:28-46
    def __init__(self, n_circles, min_radius, max_radius):
        self.n_circles, self.min_radius, self.max_radius = n_circles, min_radius, max_radius

    def generate(self, w, h):
        gen = np.zeros((w, h), dtype=bool)
        for _ in range(self.n_circles):
            radius = int(random.random() * (self.max_radius - self.min_radius) + self.min_radius)
            xc = int(random.random() * (w - 2 * radius) + radius)
            yc = int(random.random() * (h - 2 * radius) + radius)
            xs = np.arange(xc - radius, xc + radius + 1)
            ys = np.arange(yc - radius, yc + radius + 1)
            # ((xc-x)**2 + (yc-y)**2) ** 0.5 <= radius with integer operands
            inside = (xc - xs)[:, None] ** 2 + (yc - ys)[None, :] ** 2 <= radius * radius
            gx, gy = np.nonzero(inside)
            gen[xs[gx], ys[gy]] = True  # negative indices wrap exactly like the reference's gen[x, y]
        return gen


class BernoulliGenerator:
    """Independent wall cells with the given density, drawn from np.random (a stand-in for
    PerlinGenerator)."""

    def __init__(self, density=0.05):
        self.density = density

    def generate(self, w, h):
        return np.random.random((w, h)) < self.density


class EmptyGenerator:
    def generate(self, w, h):
        return np.zeros((w, h), dtype=bool)


# Ken Perlin's reference permutation (Improved Noise, 2002)
_PERM = np.array([
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
    78, 66, 215, 61, 156, 180], dtype=np.int64)
_PERM2 = np.concatenate([_PERM, _PERM])
# gradient directions (x, y components of the 16-entry table of the improved noise)
_GRAD = np.array([[1, 1], [-1, 1], [1, -1], [-1, -1], [1, 0], [-1, 0], [1, 0], [-1, 0],
                  [0, 1], [0, -1], [0, 1], [0, -1], [1, 0], [-1, 0], [0, -1], [0, 1]], dtype=np.float32)


def _noise2(x, y, repeatx, repeaty, base=0):
    """One octave of 2-D improved Perlin noise on float32 arrays (tiles with period repeatx / repeaty)."""
    f32 = np.float32
    x, y = x.astype(f32), y.astype(f32)
    i = np.floor(np.fmod(x, f32(repeatx))).astype(np.int64)
    j = np.floor(np.fmod(y, f32(repeaty))).astype(np.int64)
    ii = np.fmod((i + 1).astype(f32), f32(repeatx)).astype(np.int64)
    jj = np.fmod((j + 1).astype(f32), f32(repeaty)).astype(np.int64)
    i, j, ii, jj = (i & 255) + base, (j & 255) + base, (ii & 255) + base, (jj & 255) + base
    x = x - np.floor(x)
    y = y - np.floor(y)
    fx = x * x * x * (x * (x * f32(6) - f32(15)) + f32(10))
    fy = y * y * y * (y * (y * f32(6) - f32(15)) + f32(10))
    A, B = _PERM2[i], _PERM2[ii]
    AA, AB, BA, BB = _PERM2[A + j], _PERM2[A + jj], _PERM2[B + j], _PERM2[B + jj]

    def grad(h, gx, gy):
        g = _GRAD[_PERM2[h] & 15]
        return gx * g[..., 0] + gy * g[..., 1]

    def lerp(t, a, b):
        return a + t * (b - a)
    one = f32(1)
    return lerp(fy, lerp(fx, grad(AA, x, y), grad(BA, x - one, y)),
                lerp(fx, grad(AB, x, y - one), grad(BB, x - one, y - one)))


def perlin_noise(w, h, offset_x, offset_y, scale=22.0, octaves=2, persistence=0.5, lacunarity=2.0):
    """utils.py:7-17 (`perlin_noise_generator`): gen[i, j] = pnoise2((i + offset_x) / scale,
    (j + offset_y) / scale, octaves, persistence, lacunarity, base=0), values in [-1, 1]."""
    f32 = np.float32
    x = ((np.arange(w, dtype=np.float64) + offset_x) / scale).astype(f32)[:, None] * np.ones((1, h), f32)
    y = ((np.arange(h, dtype=np.float64) + offset_y) / scale).astype(f32)[None, :] * np.ones((w, 1), f32)
    if octaves == 1:
        return _noise2(x, y, 1024.0, 1024.0).astype(np.float64)
    freq, amp, mx = f32(1), f32(1), f32(0)
    total = np.zeros((w, h), f32)
    for _ in range(octaves):
        total = total + _noise2(x * freq, y * freq, 1024.0 * float(freq), 1024.0 * float(freq)) * amp
        mx = f32(mx + amp)
        freq = f32(freq * f32(lacunarity))
        amp = f32(amp * f32(persistence))
    return (total / mx).astype(np.float64)


class PerlinGenerator:  # generator/map_generators.py:9-25
    def __init__(self, scale=22.0, density=0.05, octaves=2, persistence=0.5, lacunarity=2.0):
        self.scale, self.density, self.octaves = scale, density, octaves
        self.persistence, self.lacunarity = persistence, lacunarity

    def generate(self, w, h):
        # two draws from the global `random` stream, like the reference (:19-20)
        ox = random.randint(-10000, 10000)
        oy = random.randint(-10000, 10000)
        return perlin_noise(w, h, ox, oy, self.scale, self.octaves, self.persistence, self.lacunarity) > self.density


class EnvironmentGenerator:  # generator/environment_generator.py:19-106
    def __init__(self, w, h, n_ants, n_pheromones, n_rocks, food_generator, walls_generator, max_steps,
                 seed=None, n_envs=1, env_id_base=0, n_envs_total=0):
        # env_id_base / n_envs_total (extension, AntsCfg.env_id_base): this generator builds envs [env_id_base,
        # env_id_base + n_envs) of a sharded batch of n_envs_total — global env g is drawn with seed + g on any shard
        self.env_id_base, self.n_envs_total = int(env_id_base), int(n_envs_total)
        self.w, self.h, self.n_ants = w, h, n_ants
        self.n_pheromones, self.n_rocks = n_pheromones, n_rocks
        self.food_generator, self.walls_generator = food_generator, walls_generator
        self.perception_mask = cm.DEFAULT_MASK.astype(bool)  # :35-41
        self.perception_shift = 4                            # :43
        self.max_steps, self.seed, self.n_envs = max_steps, seed, n_envs

    def setup_perception(self, new_mask, new_shift):  # :48-50
        self.perception_mask, self.perception_shift = new_mask, new_shift

    def _draw_one(self, seed):
        """One environment's initial arrays; RNG call order of environment_generator.py:52-94."""
        w, h, n = self.w, self.h, self.n_ants
        if seed is not None:
            random.seed(seed)          # :54
            np.random.seed(seed * 5)   # :55
        ax = int(random.random() * w * 0.5 + w * 0.25)                     # :61
        ay = int(random.random() * h * 0.5 + h * 0.25)                     # :62
        ar = int(random.random() * min(w, h) * 0.05 + min(w, h) * 0.05)    # :63
        xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
        area = (ax - xs) ** 2 + (ay - ys) ** 2 <= ar * ar                  # anthill.py:28-33
        walls = np.array(self.walls_generator.generate(w, h)).astype(bool) # :66
        walls[area] = False                                                # :67
        food = np.array(self.food_generator.generate(w, h)).astype(float)  # :71, food.py:15
        food *= (1 - walls)                                                # :72
        rocks = np.zeros((0, 4))
        if self.n_rocks > 0:                                               # :76-85
            c = np.random.random((self.n_rocks, 2))
            c[:, 0] *= w * 0.75
            c[:, 1] *= h * 0.25
            c[:, 0] += w * 0.25
            c[:, 1] += h * 0.25
            rad = np.random.random(self.n_rocks) * 5 + 5
            wgt = np.random.random(self.n_rocks) * 50 + 50
            rocks = np.concatenate([c, rad[:, None], wgt[:, None]], axis=1)
        ang = np.random.random(n) * 2 * np.pi                              # :87
        dist = np.random.random(n) * ar * 0.8                              # :88
        x = np.cos(ang) * dist + ax                                        # :89
        y = np.sin(ang) * dist + ay                                        # :90
        t = np.random.random(n) * 2 * np.pi                                # :91
        seed_arr = np.random.random(n)                                     # Ants.__init__, ants.py:41
        return dict(ants_xyt=np.array([x, y, t]).T, seed=seed_arr, walls=walls.astype(np.uint8),
                    food=food.astype(np.float32), anthill_xyr=np.array([ax, ay, ar], np.int32), rocks=rocks)

    def draw(self):
        """Initial state of the whole batch as env-major numpy arrays (AntsInit layout)."""
        per = [self._draw_one(None if self.seed is None else self.seed + self.env_id_base + e) for e in range(self.n_envs)]
        init = {k: np.stack([p[k] for p in per]) for k in per[0]}
        if self.n_rocks == 0:
            init.pop("rocks")
        return init

    def generate(self, rl_api: RLApi) -> Environment:  # :52-106
        init = self.draw()
        env = Environment(self.w, self.h, self.max_steps)
        perceived = []
        anthill = Anthill(env, init["anthill_xyr"])
        perceived.append(anthill)
        walls = Walls(env)
        perceived.append(walls)
        food = Food(env)
        perceived.append(food)
        if self.n_rocks > 0:
            r = init["rocks"]
            rocks = CircleObstacles(env, r[..., 2][0] if self.n_envs == 1 else r[..., 2],
                                    r[..., 3][0] if self.n_envs == 1 else r[..., 3])
            perceived.append(rocks)
        ants = Ants(env, self.n_ants, self.n_envs, 5)                      # :93
        perceived.insert(0, ants)                                          # :94
        for p in range(self.n_pheromones):                                 # :96-99
            phero = Pheromone(env, p, color=PHERO_COLORS[p % len(PHERO_COLORS)], max_val=255)
            ants.register_pheromone(phero)
            perceived.insert(p + 1, phero)
        rl_api.register_ants(ants)                                         # :101
        rl_api._pending = (dict(n_envs=self.n_envs, n_ants=self.n_ants, w=self.w, h=self.h,
                                n_phero=self.n_pheromones, n_rocks=self.n_rocks, max_time=self.max_steps,
                                max_hold=5.0, phero_max_val=255.0, deposit_strength=1.0,
                                env_id_base=self.env_id_base, n_envs_total=self.n_envs_total), init)
        mask = None if self.perception_mask is None else np.asarray(self.perception_mask)
        radius = mask.shape[0] // 2 if mask is not None else 3
        rl_api.setup_perception(radius, perceived, mask, self.perception_shift)  # :102-105
        return env


class DeviceEnvironmentGenerator(EnvironmentGenerator):
    """EnvironmentGenerator whose draws happen ON THE GPU (antsrl_generate): no
    O(W*H) Python loops (anthill.py:29-33, map_generators.py:42-46), no host arrays, no upload.
    Walls are independent cells of the given density, or — walls_generator=PerlinGenerator(...), as in
    main.py:75 — that generator's Perlin caves drawn on the device (its scale / density / octaves /
    persistence / lacunarity; the same noise function as PerlinGenerator.generate on the host); food is
    `n_food_discs` discs of radius food_rmin..food_rmax (CirclesGenerator's family, main.py:74).
    reference_streams=False: counter-based random streams — the reference's distributions, different maps.
    reference_streams=True: the reference's own MT19937 streams on the device (ANTSRL_RNG_REFERENCE): env e is
    EnvironmentGenerator(seed=seed + e).generate of the reference — same anthill, food discs, rocks, ants and
    per-ant seeds for equal seeds; walls_generator is then a PerlinGenerator (drawn on the device), None (no walls)
    or any object with .generate(w, h) -> bool[w, h] that does not use the global random streams (called on the
    host once per env, uploaded as bitmaps).
    auto_reset=True regenerates every env (the next seeds) right after the update of the step that reported done.
    Two limits of auto_reset with reference_streams=True (include/antsrl.h, antsrl_generate): (1) episode k draws env e
    from seed + k * n_envs_total + env_id_base + e, and np.random.seed takes 32 bits — once (seed + env_id_base + n_envs) * 5 would pass 2^32 the step that
    triggers the reset raises (ANTSRL_E_INVALID) instead of wrapping onto an earlier episode's seed; (2) bitmaps of a
    custom walls_generator are drawn ONCE here and re-used by every later episode (the reference calls
    walls_generator.generate per episode, environment_generator.py:66): call generate() again between episodes if the
    walls must change — a PerlinGenerator has no such limit (its offsets are drawn per episode on the device)."""

    def __init__(self, w, h, n_ants, n_pheromones, n_rocks, max_steps, seed=0, n_envs=1, wall_density=0.05,
                 n_food_discs=20, food_rmin=5, food_rmax=10, auto_reset=False, walls_generator=None,
                 reference_streams=False, env_id_base=0, n_envs_total=0):
        super().__init__(w, h, n_ants, n_pheromones, n_rocks, None, walls_generator, max_steps, seed=seed, n_envs=n_envs,
                         env_id_base=env_id_base, n_envs_total=n_envs_total)
        rng = "reference" if reference_streams else "counter"
        self.walls_bitmaps = None
        if walls_generator is None:
            self.gen = cm.make_gen(0.0 if reference_streams else wall_density, n_food_discs, food_rmin, food_rmax,
                                   auto_reset, rng=rng)
        elif isinstance(walls_generator, PerlinGenerator):
            g = walls_generator
            self.gen = cm.make_gen(g.density, n_food_discs, food_rmin, food_rmax, auto_reset, walls="perlin",
                                   perlin_scale=g.scale, perlin_octaves=g.octaves, perlin_persistence=g.persistence,
                                   perlin_lacunarity=g.lacunarity, rng=rng)
        elif reference_streams:
            self.walls_bitmaps = np.stack([np.asarray(walls_generator.generate(w, h)).astype(np.uint8) for _ in range(n_envs)])
            self.gen = cm.make_gen(0.0, n_food_discs, food_rmin, food_rmax, auto_reset, walls="input", rng=rng)
        else:
            raise TypeError("the device generator draws Bernoulli walls (walls_generator=None) or a PerlinGenerator's")

    def generate(self, rl_api: RLApi) -> Environment:
        env = Environment(self.w, self.h, self.max_steps)
        # anthill / rock parameters are only known on the device: the views read them back lazily
        anthill = _DeviceAnthill(env)
        perceived = [anthill, Walls(env), Food(env)]
        if self.n_rocks > 0:
            perceived.append(_DeviceRocks(env, self.n_rocks))
        ants = Ants(env, self.n_ants, self.n_envs, 5)
        perceived.insert(0, ants)
        for p in range(self.n_pheromones):
            phero = Pheromone(env, p, color=PHERO_COLORS[p % len(PHERO_COLORS)], max_val=255)
            ants.register_pheromone(phero)
            perceived.insert(p + 1, phero)
        rl_api.register_ants(ants)
        rl_api._pending = (dict(n_envs=self.n_envs, n_ants=self.n_ants, w=self.w, h=self.h,
                                n_phero=self.n_pheromones, n_rocks=self.n_rocks, max_time=self.max_steps,
                                max_hold=5.0, phero_max_val=255.0, deposit_strength=1.0,
                                env_id_base=self.env_id_base, n_envs_total=self.n_envs_total),
                           ("device", self.gen, int(self.seed or 0), self.walls_bitmaps))
        mask = None if self.perception_mask is None else np.asarray(self.perception_mask)
        rl_api.setup_perception(mask.shape[0] // 2 if mask is not None else 3, perceived, mask, self.perception_shift)
        return env


class _DeviceAnthill(Anthill):
    """Anthill view whose x / y / radius come from the device (they change at every auto-reset)."""

    def __init__(self, environment):
        super().__init__(environment, np.zeros((1, 3), np.int32))

    @property
    def _xyr_now(self):
        return self._read(cm.S_ANTHILL_XYR).astype(np.int64)  # [E, 3]: x, y, radius as the device generator drew them

    x = property(lambda self: int(self._xyr_now[0, 0]) if self._xyr_now.shape[0] == 1 else self._xyr_now[:, 0])
    y = property(lambda self: int(self._xyr_now[0, 1]) if self._xyr_now.shape[0] == 1 else self._xyr_now[:, 1])
    radius = property(lambda self: int(self._xyr_now[0, 2]) if self._xyr_now.shape[0] == 1 else self._xyr_now[:, 2])

    @property
    def area(self):
        a = self._read(cm.S_ANTHILL_AREA).astype(bool)
        return a[0] if a.shape[0] == 1 else a


class _DeviceRocks(CircleObstacles):
    def __init__(self, environment, n_rocks):
        super().__init__(environment, np.zeros(n_rocks), np.zeros(n_rocks))
