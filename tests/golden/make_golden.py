#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (the reference lives at /root/reference there and never
travels):   python tests/golden/make_golden.py

The reference is imported unmodified with the two in-process shims of SURVEY.md §8(c):
a dummy ``noise`` module (imported but unused on this path) and ``np.bool`` (alias removed
from numpy >= 1.24, used at environment/ants.py:83).  Nothing of the reference's source is
stored: a fixture holds only inputs (initial state, actions, the random draws
Walls.update consumed) and the outputs / state the reference produced.

Fixture layout (np.savez_compressed):
  meta_json        configuration (sizes, reward kind and weights, filter, max_time, ...)
  init_*           ants_xyt[N,3], seed[N], walls[W,H], food[W,H], anthill_xyr[3], rocks[R,4],
                   activation[N,C] (after the optional activate_all_pheromones call)
  ops[T]           0 = api.step, 1 = env.update, 2 = api.observation
  rot[T,N], ph[T,N], has_rot[T], has_ph[T]      actions of step ops
  jitter[T,N], hits[T]                          np.random.random draws of update ops
  after every op t:
    ants[T,N,3] prev[T,N,2] holding[T,N] mandibles[T,N] activation[T,N,C] reward_state[T,N]
    food[T,W,H] phero[T,C,W,H] explored[T,W,H] anthill_food[T] rock_centers[T,R,2] timestep[T]
  after step / observe ops:  obs[T,N,P,P,K] agent_state[T,N,2] reward[T,N] done[T]
"""
import json
import os
import random
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.modules.setdefault("noise", types.ModuleType("noise"))  # food.py:2, anthill.py:2, utils.py:2
if not hasattr(np, "bool"):
    np.bool = bool  # ants.py:83
sys.path.insert(0, "/root/reference")

import environment.pheromone as ref_pheromone  # noqa: E402
from environment.RL_api import RLApi  # noqa: E402
from environment.circle_obstacles import CircleObstacles  # noqa: E402
from environment.rewards.reward import Reward  # noqa: E402
from environment.rewards.reward_custom import ExplorationReward, All_Rewards, Food_Reward  # noqa: E402
from environment.pheromone import Pheromone  # noqa: E402
from environment.food import Food  # noqa: E402
from environment.walls import Walls  # noqa: E402
from environment.anthill import Anthill  # noqa: E402
from generator.environment_generator import EnvironmentGenerator  # noqa: E402
from generator.map_generators import CirclesGenerator  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
OP_STEP, OP_UPDATE, OP_OBSERVE = 0, 1, 2


class BernoulliWalls:
    """Stand-in for PerlinGenerator (needs the absent `noise` package): any bool[w,h] bitmap
    is a valid input to the path."""

    def __init__(self, density, rng):
        self.density, self.rng = density, rng

    def generate(self, w, h):
        return self.rng.random((w, h)) < self.density


class FoodNearAnthill:
    """CirclesGenerator plus one disc placed by the caller (to force pickups early and food
    lying on the anthill area at reset)."""

    def __init__(self, n, rmin, rmax, extra=None, rich=None):
        self.base = CirclesGenerator(n, rmin, rmax)
        self.extra = extra
        self.rich = rich  # optional rng: integer quantities 0..8 per cell instead of 0/1

    def generate(self, w, h):
        g = self.base.generate(w, h)
        if self.extra is not None:
            cx, cy, r = self.extra
            xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
            g |= ((xs - cx) ** 2 + (ys - cy) ** 2) <= r * r
        if self.rich is not None:
            return g.astype(int) * self.rich.integers(1, 9, size=g.shape)
        return g


def make_reward(kind, weights):
    if kind == "exploration":
        return ExplorationReward()
    if kind == "food":
        return Food_Reward()
    if kind == "all":
        return All_Rewards(**weights)
    return Reward()


def run_scenario(name, *, w=64, h=64, n_ants=32, n_phero=2, steps=60, seed=3, wall_density=0.0,
                 n_rocks=0, reward="exploration", weights=None, float_activation=False,
                 filt=None, max_time=2000, script=None, food_extra=True, none_actions=False,
                 store_every=1, rich_food=False):
    rng = np.random.default_rng(1000 + seed)
    weights = weights or {}
    old_filter = ref_pheromone.DIFFUSE_FILTER
    if filt is not None:
        ref_pheromone.DIFFUSE_FILTER = np.asarray(filt, dtype=float)  # read at call time, pheromone.py:44
    try:
        api = RLApi(make_reward(reward, weights), reward_threshold=1, max_speed=1,
                    max_rot_speed=40 / 180 * np.pi, carry_speed_reduction=0.05,
                    backward_speed_reduction=0.5)
        # the generator draws the anthill from `random` first; peek so food can be put near it
        random.seed(seed)
        ax = int(random.random() * w * 0.5 + w * 0.25)
        ay = int(random.random() * h * 0.5 + h * 0.25)
        extra = (ax + 2, ay + 1, 6) if food_extra else None
        gen = EnvironmentGenerator(w, h, n_ants, n_phero, 0,
                                   FoodNearAnthill(6, 3, 6, extra, rng if rich_food else None),
                                   BernoulliWalls(wall_density, rng), max_time, seed=seed)
        env = gen.generate(api)
        ants = api.ants
        rocks = None
        if n_rocks > 0:
            # generator's rock branch raises NameError (environment_generator.py:83-84):
            # build the object directly, near the anthill so ants collide with it.
            centers = np.stack([ax + rng.uniform(-12, 12, n_rocks), ay + rng.uniform(-12, 12, n_rocks)], 1)
            rocks = CircleObstacles(env, centers=centers.copy(), radiuses=rng.random(n_rocks) * 5 + 5,
                                    weights=rng.random(n_rocks) * 50 + 50)
            api.perceived_objects.append(rocks)
        if float_activation:
            ants.activate_all_pheromones(np.ones((n_ants, n_phero)) * 10)  # collect_agent_memory.py:129-131

        objs = {type(o).__name__: o for o in env.objects}
        pheros = [o for o in env.objects if isinstance(o, Pheromone)]
        food, walls, anthill = objs["Food"], objs["Walls"], objs["Anthill"]
        kinds = []
        for o in api.perceived_objects:
            kinds.append(type(o).__name__)

        meta = dict(name=name, w=w, h=h, n_ants=n_ants, n_phero=n_phero, n_rocks=n_rocks,
                    reward=reward, weights=weights, max_time=max_time,
                    deposit_strength=256.0 if float_activation else 1.0,
                    filter=np.asarray(ref_pheromone.DIFFUSE_FILTER).tolist(),
                    perceived=kinds, fwd_delta=float(api.perception_fwd_delta),
                    max_speed=api.max_speed, max_rot_speed=api.max_rot_speed,
                    carry=api.carry_speed_reduction, backward=api.backward_speed_reduction,
                    max_hold=float(ants.max_hold), max_val=float(pheros[0].max_val),
                    reward_threshold=api.reward_threshold, seed=seed)
        rec = dict(
            init_ants_xyt=ants.ants.copy(), init_seed=ants.seed.copy(), init_walls=walls.map.copy(),
            init_food=food.qte.copy(), init_anthill_xyr=np.array([anthill.x, anthill.y, anthill.radius]),
            init_rocks=(np.concatenate([rocks.centers, rocks.radiuses[:, None], rocks.weights[:, None]], 1)
                        if rocks is not None else np.zeros((0, 4))),
            init_activation=ants.phero_activation.astype(float).copy(),
            init_anthill_area=anthill.area.copy(), mask=api.perception_mask.copy())

        if script is None:
            script = []
            for _ in range(steps):
                script += [OP_STEP, OP_UPDATE]
        T = len(script)
        P = api.perception_coords.shape[0]
        K = len(api.perceived_objects)
        R = n_rocks
        S = dict(ops=np.array(script, np.int8), rot=np.zeros((T, n_ants), np.int8),
                 ph=np.zeros((T, n_ants), np.int8), has_rot=np.zeros(T, np.uint8), has_ph=np.zeros(T, np.uint8),
                 jitter=np.zeros((T, n_ants)), hits=np.zeros(T, np.int32),
                 ants=np.zeros((T, n_ants, 3)), prev=np.zeros((T, n_ants, 2)), holding=np.zeros((T, n_ants)),
                 mandibles=np.zeros((T, n_ants), np.uint8), activation=np.zeros((T, n_ants, n_phero)),
                 reward_state=np.zeros((T, n_ants), np.uint8),
                 food=np.zeros((T, w, h), np.float32), phero=np.zeros((T, n_phero, w, h)),
                 explored=np.zeros((T, w, h), np.uint8), anthill_food=np.zeros(T),
                 rock_centers=np.zeros((T, R, 2)), timestep=np.zeros(T, np.int32),
                 obs=np.zeros((T, n_ants, P, P, K)), agent_state=np.zeros((T, n_ants, 2)),
                 reward=np.zeros((T, n_ants)), done=np.zeros(T, np.uint8), stored=np.zeros(T, np.uint8))
        real_random = np.random.random
        for t, op in enumerate(script):
            if op == OP_STEP:
                rot = rng.integers(-1, 2, n_ants)
                ph = rng.integers(0, 3, n_ants)
                use_rot, use_ph = True, True
                if none_actions:
                    use_rot, use_ph = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                S["rot"][t], S["ph"][t] = rot, ph
                S["has_rot"][t], S["has_ph"][t] = use_rot, use_ph
                o, a, r, d = api.step(rot if use_rot else None, ph if use_ph else None)
                S["obs"][t], S["agent_state"][t], S["reward"][t], S["done"][t] = o, a, r, d
            elif op == OP_OBSERVE:
                o, a, _ = api.observation()
                S["obs"][t], S["agent_state"][t] = o, a
                S["reward"][t] = api.reward.rewards
            else:
                draws = []

                def recording_random(n=None):
                    v = real_random(n)
                    draws.append(np.atleast_1d(v).copy())
                    return v

                np.random.random = recording_random
                try:
                    env.update()
                finally:
                    np.random.random = real_random
                d = np.concatenate(draws) if draws else np.zeros(0)
                S["jitter"][t, : len(d)] = d
                S["hits"][t] = len(d)
            S["ants"][t] = ants.ants
            S["prev"][t] = ants.prev_ants[:, :2]
            S["holding"][t] = ants.holding
            S["mandibles"][t] = np.asarray(ants.mandibles).astype(np.uint8)
            S["activation"][t] = ants.phero_activation.astype(float)
            S["reward_state"][t] = np.asarray(ants.reward_state).astype(np.uint8)
            S["food"][t] = food.qte
            assert np.array_equal(S["food"][t].astype(np.float64), food.qte)
            store = (t % (2 * store_every) in (0, 1)) or t >= T - 2
            S["stored"][t] = store
            if store:
                for c, p in enumerate(pheros):
                    S["phero"][t, c] = p.phero
            em = getattr(api.reward, "explored_map", None)
            if em is not None:
                S["explored"][t] = em
            S["anthill_food"][t] = anthill.food
            if rocks is not None:
                S["rock_centers"][t] = rocks.centers
            S["timestep"][t] = env.timestep
        S["explored"] = np.packbits(S["explored"], axis=-1)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, meta_json=np.array(json.dumps(meta)), **rec, **S)
        print("%-28s T=%3d  pickups(max holding)=%g  hits=%d  anthill_food=%g  %.0f KiB" % (
            name, T, S["holding"].max(), S["hits"].sum(), S["anthill_food"][-1], os.path.getsize(path) / 1024))
    finally:
        ref_pheromone.DIFFUSE_FILTER = old_filter


def diffuse3(d, e=0.001):
    f = np.ones((3, 3)) * d
    f[1, 1] = 1 - 8 * d
    return f * (1 - e)


def radius3_filter(e=0.001):
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2 * 1.5 ** 2))
    g[0, 1] *= 1.3  # deliberately asymmetric: pins convolution (flip) vs correlation
    g[5, 2] *= 0.7
    return g / g.sum() * (1 - e)


MAIN_WEIGHTS = dict(fct_explore=1, fct_food=2, fct_anthill=10, fct_explore_holding=1, fct_headinganthill=3)  # main.py:42

if __name__ == "__main__":
    run_scenario("s01_plain", steps=60, seed=3)
    run_scenario("s02_walls", steps=60, seed=4, wall_density=0.05)
    run_scenario("s03_walls_rocks", steps=60, seed=5, wall_density=0.05, n_rocks=3)
    run_scenario("s04_float_activation", steps=60, seed=6, wall_density=0.05, float_activation=True)
    run_scenario("s05_all_rewards", steps=60, seed=7, wall_density=0.05, reward="all", weights=MAIN_WEIGHTS,
                 float_activation=True)
    run_scenario("s06_food_reward", steps=40, seed=8, wall_density=0.03, reward="food")
    run_scenario("s07_diffuse3x3", steps=40, seed=9, wall_density=0.05, filt=diffuse3(0.05), float_activation=True)
    run_scenario("s08_radius3", steps=30, seed=10, wall_density=0.05, filt=radius3_filter(), float_activation=True)
    # odd call patterns: observation() before the first step (main.py:88), two steps without an
    # update, None actions (RL_api.py:187,190), `done` at timestep == max_time (RL_api.py:200)
    scr = [OP_OBSERVE] + [OP_STEP, OP_UPDATE] * 4 + [OP_STEP, OP_STEP, OP_UPDATE, OP_OBSERVE, OP_UPDATE] + \
          [OP_STEP, OP_UPDATE] * 10
    run_scenario("s09_script_none_done", seed=11, wall_density=0.05, n_rocks=2, reward="all",
                 weights=MAIN_WEIGHTS, max_time=9, script=scr, none_actions=True)
    run_scenario("s10_rect_96x48", w=96, h=48, n_ants=40, steps=40, seed=12, wall_density=0.05, n_rocks=2)
    run_scenario("s11_base_reward", steps=12, seed=13, wall_density=0.05, reward="none")
    run_scenario("s13_rich_food", steps=50, seed=15, wall_density=0.03, reward="all", weights=MAIN_WEIGHTS,
                 float_activation=True, rich_food=True)
    run_scenario("s12_256_n256", w=256, h=256, n_ants=256, steps=20, seed=14, wall_density=0.05,
                 n_rocks=4, store_every=5)
