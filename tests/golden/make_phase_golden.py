#!/usr/bin/env python3
"""Generates tests/golden/contract/update_phases_ref.npz by RUNNING THE REFERENCE — build container only (the reference
lives at /root/reference there and never travels):

    python tests/golden/make_phase_golden.py

What it pins: Environment.update calls every object's update() in stable update_step() order (environment.py:42-47), so
an EnvObject the CALLER adds runs BETWEEN the world's objects — after Walls (-1), after the step-0 objects (Food,
CircleObstacles, Pheromone, RLApi), after Ants (999), after Anthill (1000).  The scenario adds six recording probes
(update_step -2, -1, 0, 500, 999, 1000; appended after the generator's objects) to a 64 x 64 world with walls, three
circle obstacles, float activation, 32 ants, and runs 10 x (api.step, env.update) with the shipped centre-only filter
("scaled") and with a 3 x 3 diffusion filter ("diffuse"); every probe records, at its own update(), what the world looks
like at that moment: ants (x, y, theta), previous positions, rock centres, both pheromone grids, food, anthill.food,
env.timestep.  Also recorded: RLApi.perceptive_field after every step (save_perceptive_field, main.py:51; RL_api.py:144-153).  Same import shims as make_golden.py; only inputs and recorded outputs are stored."""
import os
import random
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.modules.setdefault("noise", types.ModuleType("noise"))
if not hasattr(np, "bool"):
    np.bool = bool
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import environment.pheromone as ref_pheromone  # noqa: E402
from environment.RL_api import RLApi  # noqa: E402
from environment.circle_obstacles import CircleObstacles  # noqa: E402
from environment.environment import EnvObject  # noqa: E402
from environment.pheromone import Pheromone  # noqa: E402
from environment.rewards.reward_custom import ExplorationReward  # noqa: E402
from generator.environment_generator import EnvironmentGenerator  # noqa: E402
from make_golden import BernoulliWalls, FoodNearAnthill, diffuse3  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "contract")
PROBE_STEPS = (-2, -1, 0, 500, 999, 1000)


def run(tag, filt, seed=21, w=64, h=64, n_ants=32, n_rocks=3, steps=10):
    rng = np.random.default_rng(1000 + seed)
    old = ref_pheromone.DIFFUSE_FILTER
    if filt is not None:
        ref_pheromone.DIFFUSE_FILTER = np.asarray(filt, dtype=float)
    try:
        api = RLApi(ExplorationReward(), reward_threshold=1, max_speed=1, max_rot_speed=40 / 180 * np.pi,
                    carry_speed_reduction=0.05, backward_speed_reduction=0.5)
        random.seed(seed)
        ax = int(random.random() * w * 0.5 + w * 0.25)
        ay = int(random.random() * h * 0.5 + h * 0.25)
        gen = EnvironmentGenerator(w, h, n_ants, 2, 0, FoodNearAnthill(6, 3, 6, (ax + 2, ay + 1, 6), None),
                                   BernoulliWalls(0.06, rng), 2000, seed=seed)
        env = gen.generate(api)
        ants = api.ants
        centers = np.stack([ax + rng.uniform(-10, 10, n_rocks), ay + rng.uniform(-10, 10, n_rocks)], 1)
        rocks = CircleObstacles(env, centers=centers.copy(), radiuses=rng.random(n_rocks) * 4 + 4, weights=rng.random(n_rocks) * 50 + 50)
        api.perceived_objects.append(rocks)
        ants.activate_all_pheromones(np.ones((n_ants, 2)) * 10)
        objs = {type(o).__name__: o for o in env.objects}
        pheros = [o for o in env.objects if isinstance(o, Pheromone)]
        food, walls, anthill = objs["Food"], objs["Walls"], objs["Anthill"]
        rec = {"%s_init_ants_xyt" % tag: ants.ants.copy(), "%s_init_seed" % tag: ants.seed.copy(), "%s_init_walls" % tag: walls.map.copy(),
               "%s_init_food" % tag: food.qte.copy(), "%s_init_anthill_xyr" % tag: np.array([anthill.x, anthill.y, anthill.radius]),
               "%s_init_rocks" % tag: np.concatenate([rocks.centers, rocks.radiuses[:, None], rocks.weights[:, None]], 1),
               "%s_mask" % tag: api.perception_mask.copy(), "%s_filter" % tag: np.asarray(ref_pheromone.DIFFUSE_FILTER, dtype=float)}
        seen = {s: [] for s in PROBE_STEPS}

        class Probe(EnvObject):
            def __init__(self, environment, step):
                self.step = step
                super().__init__(environment)

            def update_step(self):
                return self.step

            def update(self):
                seen[self.step].append(dict(ants=ants.ants.copy(), prev=ants.prev_ants[:, :2].copy(), rocks=rocks.centers.copy(),
                                            phero=np.stack([p.phero.copy() for p in pheros]), food=food.qte.copy(),
                                            anthill_food=float(anthill.food), timestep=int(env.timestep)))

        for s in PROBE_STEPS[::-1]:  # (added in reverse: the sort, not the insertion order, decides)
            Probe(env, s)
        rot = rng.integers(-1, 2, (steps, n_ants))
        ph = rng.integers(0, 3, (steps, n_ants))
        jit = np.zeros((steps, n_ants))
        fields = []
        api.save_perceptive_field = True  # main.py:51: RLApi.observation then leaves the cells the ants perceive (RL_api.py:144-153)
        real_random = np.random.random
        for t in range(steps):
            api.step(rot[t], ph[t])
            fields.append(api.perceptive_field.copy())
            draws = []

            def recording(n=None):
                v = real_random(n)
                draws.append(np.atleast_1d(v).copy())
                return v
            np.random.random = recording
            try:
                env.update()
            finally:
                np.random.random = real_random
            d = np.concatenate(draws) if draws else np.zeros(0)
            jit[t, :len(d)] = d
        rec["%s_rot" % tag], rec["%s_ph" % tag], rec["%s_jitter" % tag] = rot.astype(np.int8), ph.astype(np.int8), jit
        rec["%s_perceptive_field" % tag] = np.packbits(np.stack(fields), axis=-1)  # [steps, w, h / 8]
        for s in PROBE_STEPS:
            for k in ("ants", "prev", "rocks", "phero", "food"):
                rec["%s_probe%d_%s" % (tag, s, k)] = np.stack([x[k] for x in seen[s]])
            rec["%s_probe%d_anthill_food" % (tag, s)] = np.array([x["anthill_food"] for x in seen[s]])
            rec["%s_probe%d_timestep" % (tag, s)] = np.array([x["timestep"] for x in seen[s]], np.int32)
        hits = int((jit != 0).sum())
        moved = float(np.abs(rec["%s_probe500_rocks" % tag][-1] - rec["%s_init_rocks" % tag][:, :2]).max())
        print("%-8s %d updates, wall hits %d, rocks moved by up to %.3f, differs between probes -1 / 0 / 500 / 999: ants %s, phero %s" % (
            tag, steps, hits, moved,
            [bool((rec["%s_probe%d_ants" % (tag, a)] != rec["%s_probe%d_ants" % (tag, b)]).any()) for a, b in ((-2, -1), (-1, 0), (500, 999))],
            [bool((rec["%s_probe%d_phero" % (tag, a)] != rec["%s_probe%d_phero" % (tag, b)]).any()) for a, b in ((-1, 0), (500, 999), (999, 1000))]))
        return rec
    finally:
        ref_pheromone.DIFFUSE_FILTER = old


if __name__ == "__main__":
    out = {}
    out.update(run("scaled", None))
    out.update(run("diffuse", diffuse3(0.05)))
    out["probe_steps"] = np.array(PROBE_STEPS)
    path = os.path.join(OUT, "update_phases_ref.npz")
    np.savez_compressed(path, **out)
    print("%s  %.0f KiB" % (path, os.path.getsize(path) / 1024))
