"""ctypes mirror of ``AntsCfg`` / ``AntsInit`` (include/antsrl.h) and the defaults the
reference hard-codes.

Defaults follow the reference's own constants:
  * perception mask / shift     generator/environment_generator.py:35-43
  * DELTA = 1.1                 environment/RL_api.py:15
  * speeds                      main.py:45-50
  * max_hold = 5, max_val = 255 generator/environment_generator.py:93,97
  * EVAP/DIFFUSE filter         environment/pheromone.py:5-10
  * channel order               generator/environment_generator.py:60-105
                                ([Ants, Phero.., Anthill, Walls, Food, (Rocks)])
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import numpy as np

ABI_VERSION = 5
MAX_CHANNELS = 16
MAX_PSIDE = 15
MAX_PCELLS = MAX_PSIDE * MAX_PSIDE
MAX_FILTER_RADIUS = 3
MAX_FILTER_TAPS = 49
MAX_PHERO = 4

CH_ANTS, CH_PHERO, CH_ANTHILL, CH_WALLS, CH_FOOD, CH_ROCKS = range(6)
REWARD_NONE, REWARD_EXPLORATION, REWARD_FOOD, REWARD_ALL = range(4)
PHERO_AUTO, PHERO_EXPLICIT_SWEEP = 0, 1
ACT_AUTO, ACT_CELL_META, ACT_SINGLE_KERNEL = 0, 1, 2
TIMING_EVENTS = 5  # antsrl_set_timing_events
Q_CELL_META, Q_SCALED_UNITS, Q_INTERLEAVED, Q_FILTER_SEPARABLE, Q_PERCEIVE_RUN, Q_TIMESTEP, Q_DEFERRED_UPDATE = range(7)  # antsrl_query

(S_ANTS_XYT, S_PREV_XY, S_HOLDING, S_MANDIBLES, S_ACTIVATION, S_PHERO, S_FOOD, S_EXPLORED,
 S_ANTHILL_FOOD, S_ROCK_CENTERS, S_TIMESTEP, S_REWARD_STATE, S_WALLS, S_ANTHILL_AREA,
 S_SEED, S_ANTHILL_XYR, S_ROCK_RW, S_PHERO_C0, S_PHERO_C1, S_PHERO_C2, S_PHERO_C3) = range(21)


class AntsCfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_envs", C.c_int32),
        ("n_ants", C.c_int32),
        ("w", C.c_int32),
        ("h", C.c_int32),
        ("n_phero", C.c_int32),
        ("n_rocks", C.c_int32),
        ("max_time", C.c_int32),
        ("perception_radius", C.c_int32),
        ("n_channels", C.c_int32),
        ("channel_kind", C.c_int32 * MAX_CHANNELS),
        ("channel_arg", C.c_int32 * MAX_CHANNELS),
        ("has_mask", C.c_int32),
        ("mask", C.c_uint8 * MAX_PCELLS),
        ("_pad0", C.c_uint8 * 7),
        ("delta", C.c_double),
        ("fwd_delta", C.c_double),
        ("max_speed", C.c_double),
        ("max_rot_speed", C.c_double),
        ("carry_speed_reduction", C.c_double),
        ("backward_speed_reduction", C.c_double),
        ("max_hold", C.c_double),
        ("has_max_val", C.c_int32),
        ("filter_radius", C.c_int32),
        ("phero_max_val", C.c_double),
        ("deposit_strength", C.c_double),
        ("phero_threshold", C.c_double),
        ("filter", C.c_double * MAX_FILTER_TAPS),
        ("reward_kind", C.c_int32),
        ("phero_mode", C.c_int32),
        ("reward_threshold", C.c_double),
        ("fct_explore", C.c_double),
        ("fct_food", C.c_double),
        ("fct_anthill", C.c_double),
        ("fct_explore_holding", C.c_double),
        ("fct_headinganthill", C.c_double),
        ("rng_seed", C.c_uint64),
        ("act_path", C.c_int32),
        ("env_id_base", C.c_int32),
        ("n_envs_total", C.c_int32),
        ("_pad1", C.c_int32),
    ]

    # convenience -----------------------------------------------------------
    @property
    def pside(self) -> int:
        return 2 * self.perception_radius + 1

    def copy(self) -> "AntsCfg":
        out = AntsCfg()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(AntsCfg))
        return out

    def obs_shape(self):
        p = self.pside
        return (self.n_envs, self.n_ants, p, p, self.n_channels)


class AntsGen(C.Structure):
    """Device-side episode generator parameters (include/antsrl.h: AntsGen)."""
    _fields_ = [("wall_density", C.c_double), ("n_food_discs", C.c_int32), ("food_rmin", C.c_int32),
                ("food_rmax", C.c_int32), ("auto_reset", C.c_int32), ("wall_kind", C.c_int32),
                ("perlin_octaves", C.c_int32), ("perlin_scale", C.c_double), ("perlin_persistence", C.c_double),
                ("perlin_lacunarity", C.c_double), ("rng_kind", C.c_int32), ("_pad", C.c_int32),
                ("walls_input", C.c_void_p)]


WALLS_BERNOULLI, WALLS_PERLIN, WALLS_INPUT = 0, 1, 2
#: antsrl_update_phase (include/antsrl.h): the reference's update steps -1 / 0 / 999 / 1000
PHASE_WALLS, PHASE_ROCKS_PHEROMONE, PHASE_ANTS, PHASE_ANTHILL = 0, 1, 2, 3
RNG_COUNTER, RNG_REFERENCE = 0, 1


def make_gen(wall_density=0.05, n_food_discs=20, food_rmin=5, food_rmax=10, auto_reset=False, walls="bernoulli",
             perlin_scale=22.0, perlin_octaves=2, perlin_persistence=0.5, perlin_lacunarity=2.0, rng="counter",
             walls_input_ptr=None) -> AntsGen:
    """Defaults: main.py:74 (CirclesGenerator(20, 5, 10)); walls 5 % independent cells.
    walls="perlin": PerlinGenerator(scale, density=wall_density, octaves, persistence, lacunarity), main.py:75;
    walls="input": the caller's bitmap (walls_input_ptr: device pointer to uint8 [E][W][H]).
    rng="reference": the reference's own MT19937 streams (ANTSRL_RNG_REFERENCE)."""
    kind = {"bernoulli": WALLS_BERNOULLI, "perlin": WALLS_PERLIN, "input": WALLS_INPUT}[walls]
    return AntsGen(wall_density, n_food_discs, food_rmin, food_rmax, 1 if auto_reset else 0, kind,
                   int(perlin_octaves), float(perlin_scale), float(perlin_persistence), float(perlin_lacunarity),
                   {"counter": RNG_COUNTER, "reference": RNG_REFERENCE}[rng], 0, walls_input_ptr)


class AntsInit(C.Structure):
    _fields_ = [
        ("ants_xyt", C.c_void_p),
        ("seed", C.c_void_p),
        ("walls", C.c_void_p),
        ("food", C.c_void_p),
        ("anthill_xyr", C.c_void_p),
        ("rocks", C.c_void_p),
        ("phero", C.c_void_p),
    ]


#: generator/environment_generator.py:35-41
DEFAULT_MASK = np.array([[0, 0, 1, 1, 1, 0, 0],
                         [0, 1, 1, 1, 1, 1, 0],
                         [1, 1, 1, 1, 1, 1, 1],
                         [1, 1, 1, 1, 1, 1, 1],
                         [1, 1, 1, 1, 1, 1, 1],
                         [0, 1, 1, 1, 1, 1, 0],
                         [0, 0, 1, 1, 1, 0, 0]], dtype=np.uint8)


def diffuse_filter(diffuse_factor: float = 0.0, evap_factor: float = 0.001) -> np.ndarray:
    """DIFFUSE_FILTER exactly as environment/pheromone.py:5-10 builds it (3x3)."""
    f = np.ones((3, 3)) * diffuse_factor
    f[1, 1] = 1 - 8 * diffuse_factor
    f *= 1 - evap_factor
    return f


def default_channels(n_phero: int, n_rocks: int):
    kinds = [CH_ANTS] + [CH_PHERO] * n_phero + [CH_ANTHILL, CH_WALLS, CH_FOOD]
    args = [0] + list(range(n_phero)) + [0, 0, 0]
    if n_rocks > 0:
        kinds.append(CH_ROCKS)
        args.append(0)
    return kinds, args


def make_cfg(n_envs: int, n_ants: int, w: int, h: int, *, n_phero: int = 2, n_rocks: int = 0,
             max_time: int = 2000, mask: Optional[np.ndarray] = DEFAULT_MASK,
             perception_radius: Optional[int] = None, fwd_delta: float = 4.0, delta: float = 1.1,
             channels: Optional[Sequence] = None,
             max_speed: float = 1.0, max_rot_speed: float = 40 / 180 * np.pi,
             carry_speed_reduction: float = 0.05, backward_speed_reduction: float = 0.5,
             max_hold: float = 5.0, phero_max_val: Optional[float] = 255.0,
             deposit_strength: float = 1.0, phero_threshold: float = 0.01,
             filt: Optional[np.ndarray] = None, reward_kind: int = REWARD_EXPLORATION,
             reward_threshold: float = 1.0, fct_explore: float = 1.0, fct_food: float = 1.0,
             fct_anthill: float = 5.0, fct_explore_holding: float = 0.0,
             fct_headinganthill: float = 1.0, rng_seed: int = 0x5EED,
             phero_mode: int = PHERO_AUTO, act_path: int = ACT_AUTO, env_id_base: int = 0,
             n_envs_total: int = 0) -> AntsCfg:
    """Build an AntsCfg with the reference's defaults (see module docstring).
    env_id_base: the GLOBAL id of this handle's environment 0 (a shard of envs [lo, hi) passes lo): every random stream
    the library keys on an environment takes env_id_base + e, so sharding cannot change a result (include/antsrl.h).
    n_envs_total: environments of the whole sharded batch (0 = env_id_base + n_envs)."""
    c = AntsCfg()
    c.abi_version = ABI_VERSION
    c.n_envs, c.n_ants, c.w, c.h = n_envs, n_ants, w, h
    c.n_phero, c.n_rocks, c.max_time = n_phero, n_rocks, max_time
    if mask is not None:
        mask = np.asarray(mask)
        assert mask.ndim == 2 and mask.shape[0] == mask.shape[1] and mask.shape[0] % 2 == 1
        r = mask.shape[0] // 2 if perception_radius is None else perception_radius
        assert mask.shape[0] == 2 * r + 1
        c.has_mask = 1
        flat = mask.astype(np.uint8).reshape(-1)
        for i, v in enumerate(flat):
            c.mask[i] = int(v)
    else:
        r = 3 if perception_radius is None else perception_radius
        c.has_mask = 0
    assert 2 * r + 1 <= MAX_PSIDE
    c.perception_radius = r
    if channels is None:
        kinds, args = default_channels(n_phero, n_rocks)
    else:
        kinds = [k if isinstance(k, int) else k[0] for k in channels]
        args = [0 if isinstance(k, int) else k[1] for k in channels]
    assert len(kinds) <= MAX_CHANNELS
    c.n_channels = len(kinds)
    for i, (k, a) in enumerate(zip(kinds, args)):
        c.channel_kind[i] = k
        c.channel_arg[i] = a
    c.delta, c.fwd_delta = delta, fwd_delta
    c.max_speed, c.max_rot_speed = max_speed, max_rot_speed
    c.carry_speed_reduction, c.backward_speed_reduction = carry_speed_reduction, backward_speed_reduction
    c.max_hold = max_hold
    c.has_max_val = 0 if phero_max_val is None else 1
    c.phero_max_val = 0.0 if phero_max_val is None else phero_max_val
    c.deposit_strength = deposit_strength
    c.phero_threshold = phero_threshold
    if filt is None:
        filt = diffuse_filter()
    filt = np.asarray(filt, dtype=np.float64)
    assert filt.ndim == 2 and filt.shape[0] == filt.shape[1] and filt.shape[0] % 2 == 1
    fr = filt.shape[0] // 2
    # A filter whose off-centre taps are all zero is a pure per-cell scale
    # (the shipped DIFFUSE_FACTOR = 0 case, pheromone.py:5): store it as radius 0.
    off = filt.copy()
    off[fr, fr] = 0.0
    if not off.any():
        filt = filt[fr:fr + 1, fr:fr + 1]
        fr = 0
    assert fr <= MAX_FILTER_RADIUS
    c.filter_radius = fr
    for i, v in enumerate(filt.reshape(-1)):
        c.filter[i] = float(v)
    c.reward_kind = reward_kind
    c.reward_threshold = reward_threshold
    c.fct_explore, c.fct_food, c.fct_anthill = fct_explore, fct_food, fct_anthill
    c.fct_explore_holding, c.fct_headinganthill = fct_explore_holding, fct_headinganthill
    c.rng_seed = rng_seed
    c.phero_mode = phero_mode
    c.act_path = act_path
    c.env_id_base = env_id_base
    c.n_envs_total = n_envs_total
    return c


def mask_array(cfg: AntsCfg) -> Optional[np.ndarray]:
    if not cfg.has_mask:
        return None
    p = cfg.pside
    return np.array(list(cfg.mask[: p * p]), dtype=bool).reshape(p, p)


__all__ = [n for n in dir() if n.isupper() or n in (
    "AntsCfg", "AntsInit", "AntsGen", "make_gen", "make_cfg", "diffuse_filter", "default_channels", "mask_array")]
_ = math


def uses_scaled_units(cfg: AntsCfg) -> bool:
    """Mirror of use_scaled() in csrc/antsrl_capi.hip: a centre-only decay filter under
    PHERO_AUTO is held in units of f0^S and needs no per-step sweep."""
    return (cfg.phero_mode == PHERO_AUTO and cfg.filter_radius == 0 and 0.5 < cfg.filter[0] <= 1.0
            and cfg.phero_threshold >= 0.0)
