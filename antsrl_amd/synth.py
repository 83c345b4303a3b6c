"""Synthetic initial states for benchmarks and large parity tests.

Seeded per environment with numpy Generator(PCG64(seed + env_id)) on the host; mirrors what
EnvironmentGenerator.generate builds (generator/environment_generator.py:52-106):
  * anthill centre uniform in the central half, radius int(u*0.05*min + 0.05*min)   (:60-63)
  * walls Bernoulli(p) per cell, cleared on the anthill area (:66-68); any bitmap is a valid input to
    the path (antsrl_amd.generator.PerlinGenerator draws the reference's cave-like ones).
  * food = n discs of radius 5..10 (main.py:74, generator/map_generators.py:34-46), zeroed on
    walls (:72)
  * ants uniform in a disc of 0.8*radius around the anthill, theta uniform [0, 2pi)   (:87-91)
  * rocks: centres uniform in the generator's band, radius U[5,10), weight U[50,100)   (:77-85)
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .config import AntsCfg


def synth_init(cfg: AntsCfg, seed: int = 1234, wall_density: float = 0.05, n_food_discs: int = 20,
               food_rmin: int = 5, food_rmax: int = 10, env_offset: int = 0) -> Dict[str, np.ndarray]:
    E, N, W, H, R = cfg.n_envs, cfg.n_ants, cfg.w, cfg.h, cfg.n_rocks
    ants = np.empty((E, N, 3))
    seeds = np.empty((E, N))
    walls = np.zeros((E, W, H), np.uint8)
    food = np.zeros((E, W, H), np.float32)
    xyr = np.empty((E, 3), np.int32)
    rocks = np.empty((E, R, 4)) if R > 0 else None
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
    m = min(W, H)
    for e in range(E):
        rng = np.random.Generator(np.random.PCG64(seed + env_offset + e))
        ax = int(rng.random() * W * 0.5 + W * 0.25)
        ay = int(rng.random() * H * 0.5 + H * 0.25)
        ar = int(rng.random() * m * 0.05 + m * 0.05)
        xyr[e] = (ax, ay, ar)
        area = (ax - xs) ** 2 + (ay - ys) ** 2 <= ar * ar
        wl = rng.random((W, H)) < wall_density
        wl[area] = False
        walls[e] = wl
        fd = np.zeros((W, H), bool)
        for _ in range(n_food_discs):
            rad = int(rng.random() * (food_rmax - food_rmin) + food_rmin)
            rad = max(0, min(rad, (m - 1) // 2))
            xc = int(rng.random() * (W - 2 * rad) + rad)
            yc = int(rng.random() * (H - 2 * rad) + rad)
            x0, x1, y0, y1 = max(xc - rad, 0), min(xc + rad + 1, W), max(yc - rad, 0), min(yc + rad + 1, H)
            sub = (xs[x0:x1, y0:y1] - xc) ** 2 + (ys[x0:x1, y0:y1] - yc) ** 2 <= rad * rad
            fd[x0:x1, y0:y1] |= sub
        food[e] = fd & ~wl
        ang = rng.random(N) * 2 * np.pi
        dist = rng.random(N) * ar * 0.8
        ants[e, :, 0] = np.cos(ang) * dist + ax
        ants[e, :, 1] = np.sin(ang) * dist + ay
        ants[e, :, 2] = rng.random(N) * 2 * np.pi
        seeds[e] = rng.random(N)
        if R > 0:
            c = rng.random((R, 2))
            rocks[e, :, 0] = c[:, 0] * W * 0.75 + W * 0.25
            rocks[e, :, 1] = c[:, 1] * H * 0.25 + H * 0.25
            rocks[e, :, 2] = rng.random(R) * 5 + 5
            rocks[e, :, 3] = rng.random(R) * 50 + 50
    # keep ants inside the grid (warp happens at reset anyway, ants.py:28)
    ants[..., 0] %= W
    ants[..., 1] %= H
    out = dict(ants_xyt=ants, seed=seeds, walls=walls, food=food, anthill_xyr=xyr)
    if R > 0:
        out["rocks"] = rocks
    return out


def random_actions(cfg: AntsCfg, steps: int, seed: int = 99):
    """Uniform random policy (collect_agent_memory.py:201-203): rotation in {-1,0,1}, pheromone in {0,1,2}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rot = rng.integers(-1, 2, (steps, cfg.n_envs, cfg.n_ants), dtype=np.int8)
    ph = rng.integers(0, 3, (steps, cfg.n_envs, cfg.n_ants), dtype=np.int8)
    return rot, ph
