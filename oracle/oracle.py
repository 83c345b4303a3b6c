"""ctypes wrapper of the CPU oracle (oracle/antsrl_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg import this
module.  The product path (``antsrl_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from antsrl_amd.config import AntsCfg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "antsrl_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH)
        for p in (src, os.path.join(_HERE, "antsrl_oracle.h"),
                  os.path.join(_HERE, "..", "include", "antsrl.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


class _State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "x", "y", "theta", "prev_x", "prev_y", "holding", "mandibles", "seed", "reward_state",
        "activation", "phero", "food", "walls", "anthill_area", "explored", "anthill_xyr",
        "anthill_food", "rock_cx", "rock_cy", "rock_radius", "rock_weight", "timestep",
        "prev_holding", "prev_dist", "reward_primed")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_jitter_u01.restype = C.c_double
        _lib.oracle_jitter_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.oracle_perlin_noise.restype = None
        _lib.oracle_perlin_noise.argtypes = [C.c_int, C.c_int, C.c_long, C.c_long, C.c_double, C.c_int, C.c_double,
                                             C.c_double, C.c_void_p]
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """A batch of E reference-semantics environments held in float64 numpy arrays."""

    def __init__(self, cfg: AntsCfg, init: Dict[str, np.ndarray], n_threads: int = 1):
        self.cfg = cfg.copy()
        self.n_threads = n_threads
        E, N, W, H, Cn, R = cfg.n_envs, cfg.n_ants, cfg.w, cfg.h, cfg.n_phero, cfg.n_rocks
        xyt = np.asarray(init["ants_xyt"], dtype=np.float64).reshape(E, N, 3)
        self.x = np.ascontiguousarray(xyt[..., 0])
        self.y = np.ascontiguousarray(xyt[..., 1])
        self.theta = np.ascontiguousarray(xyt[..., 2])
        self.prev_x = np.zeros((E, N))
        self.prev_y = np.zeros((E, N))
        self.holding = np.zeros((E, N))
        self.mandibles = np.zeros((E, N), np.uint8)
        self.seed = np.ascontiguousarray(np.asarray(init["seed"], dtype=np.float64).reshape(E, N))
        self.reward_state = np.zeros((E, N), np.uint8)
        self.activation = np.zeros((E, N, Cn))
        ph = init.get("phero")
        self.phero = (np.zeros((E, Cn, W, H)) if ph is None
                      else np.ascontiguousarray(np.asarray(ph, dtype=np.float64).reshape(E, Cn, W, H)))
        self.food = np.ascontiguousarray(np.asarray(init["food"], dtype=np.float64).reshape(E, W, H))
        self.walls = np.ascontiguousarray(np.asarray(init["walls"]).astype(np.uint8).reshape(E, W, H))
        self.anthill_area = np.zeros((E, W, H), np.uint8)
        self.explored = np.zeros((E, W, H), np.uint8)
        self.anthill_xyr = np.ascontiguousarray(np.asarray(init["anthill_xyr"], dtype=np.int32).reshape(E, 3))
        self.anthill_food = np.zeros(E)
        if R > 0:
            rocks = np.asarray(init["rocks"], dtype=np.float64).reshape(E, R, 4)
            self.rock_cx = np.ascontiguousarray(rocks[..., 0])
            self.rock_cy = np.ascontiguousarray(rocks[..., 1])
            self.rock_radius = np.ascontiguousarray(rocks[..., 2])
            self.rock_weight = np.ascontiguousarray(rocks[..., 3])
        else:
            self.rock_cx = self.rock_cy = self.rock_radius = self.rock_weight = None
        self.timestep = np.ones(E, np.int32)
        self.prev_holding = np.zeros((E, N))
        self.prev_dist = np.zeros((E, N))
        self.reward_primed = np.zeros(E, np.uint8)
        self._st = _State(*[_p(getattr(self, n)) for n, _ in _State._fields_])
        lib().oracle_reset(C.byref(self.cfg), C.byref(self._st))

    # ------------------------------------------------------------------ API
    def _outs(self, want_obs=True):
        c = self.cfg
        E, N, P, K = c.n_envs, c.n_ants, c.pside, c.n_channels
        obs = np.empty((E, N, P, P, K)) if want_obs else None
        return obs, np.empty((E, N, 2)), np.empty((E, N))

    def step(self, rotation, phero, want_obs=True):
        """RLApi.step for every env -> (obs, agent_state, reward, done[E])."""
        c = self.cfg
        obs, ast, rew = self._outs(want_obs)
        done = np.zeros(c.n_envs, np.uint8)
        rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.int8)
        ph = None if phero is None else np.ascontiguousarray(phero, dtype=np.int8)
        lib().oracle_step(C.byref(c), C.byref(self._st), _p(rot), _p(ph), _p(obs), _p(ast), _p(rew),
                          _p(done), C.c_int(self.n_threads))
        return obs, ast, rew, done

    def observe(self, want_obs=True):
        obs, ast, rew = self._outs(want_obs)
        lib().oracle_observe(C.byref(self.cfg), C.byref(self._st), _p(obs), _p(ast), _p(rew),
                             C.c_int(self.n_threads))
        return obs, ast, rew

    def update(self, wall_jitter=None):
        """Environment.update for every env; returns the per-env number of jitter draws used."""
        c = self.cfg
        hits = np.zeros(c.n_envs, np.int32)
        j = None if wall_jitter is None else np.ascontiguousarray(wall_jitter, dtype=np.float64)
        if j is not None:
            assert j.shape == (c.n_envs, c.n_ants)
        lib().oracle_update(C.byref(c), C.byref(self._st), _p(j), _p(hits), C.c_int(self.n_threads))
        return hits

    def set_activation(self, act, new_deposit_strength=0.0):
        self.activation[...] = np.asarray(act, dtype=np.float64).reshape(self.activation.shape)
        if new_deposit_strength > 0:
            self.cfg.deposit_strength = new_deposit_strength

    @property
    def ants_xyt(self):
        return np.stack([self.x, self.y, self.theta], axis=-1)

    @property
    def rock_centers(self):
        return np.stack([self.rock_cx, self.rock_cy], axis=-1)


def generate_init(cfg: AntsCfg, gen, seed: int) -> Dict[str, np.ndarray]:
    """oracle_generate_init: the device-side episode generator restated on the host."""
    E, N, W, H, R = cfg.n_envs, cfg.n_ants, cfg.w, cfg.h, cfg.n_rocks
    out = dict(ants_xyt=np.zeros((E, N, 3)), seed=np.zeros((E, N)), walls=np.zeros((E, W, H), np.uint8),
               food=np.zeros((E, W, H), np.float32), anthill_xyr=np.zeros((E, 3), np.int32),
               rocks=np.zeros((E, max(R, 1), 4)))
    lib().oracle_generate_init(C.byref(cfg), C.byref(gen), C.c_uint64(seed), _p(out["ants_xyt"]), _p(out["seed"]),
                               _p(out["walls"]), _p(out["food"]), _p(out["anthill_xyr"]), _p(out["rocks"]))
    if R == 0:
        out.pop("rocks")
    else:
        out["rocks"] = out["rocks"][:, :R]
    return out


def perlin_noise(w, h, offset_x, offset_y, scale=22.0, octaves=2, persistence=0.5, lacunarity=2.0) -> np.ndarray:
    """oracle_perlin_noise: utils.py:7-17 on the oracle's restatement of the noise function."""
    out = np.zeros((w, h))
    lib().oracle_perlin_noise(w, h, int(offset_x), int(offset_y), float(scale), int(octaves), float(persistence),
                              float(lacunarity), _p(out))
    return out


def jitter_u01(seed: int, env: int, timestep: int, ant: int) -> float:
    return lib().oracle_jitter_u01(seed, env, timestep, ant)


def max_threads() -> int:
    return lib().oracle_max_threads()
