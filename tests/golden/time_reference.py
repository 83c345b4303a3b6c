#!/usr/bin/env python3
"""Times the UNMODIFIED reference CPU path.

Build-container only: the reference lives at /root/reference here and never travels, so its timing
cannot be taken on the GPU box.  Same import shims as make_golden.py.  One process per core
(`multiprocessing`, environments are independent); reports per-core and all-core ant-steps/s.

    python tests/golden/time_reference.py            # writes profiles/r01/reference_cpu_timing.json
"""
import json
import multiprocessing as mp
import os
import sys
import time
import types


def run_case(args):
    w, h, n_ants, n_rocks, steps, seed = args
    import numpy as np
    sys.dont_write_bytecode = True
    sys.modules.setdefault("noise", types.ModuleType("noise"))
    if not hasattr(np, "bool"):
        np.bool = bool
    sys.path.insert(0, "/root/reference")
    from environment.RL_api import RLApi
    from environment.circle_obstacles import CircleObstacles
    from environment.rewards.reward_custom import ExplorationReward
    from generator.environment_generator import EnvironmentGenerator
    from generator.map_generators import CirclesGenerator

    class Walls5:
        def generate(self, ww, hh):
            return np.random.random((ww, hh)) < 0.05

    api = RLApi(ExplorationReward(), 1, 1, 40 / 180 * np.pi, 0.05, 0.5)
    env = EnvironmentGenerator(w, h, n_ants, 2, 0, CirclesGenerator(20 if w >= 128 else 5, 5 if w >= 128 else 3,
                                                                     10 if w >= 128 else 6), Walls5(), 10 ** 9,
                               seed=seed).generate(api)
    if n_rocks:
        rng = np.random.default_rng(seed)
        c = np.stack([rng.random(n_rocks) * w * 0.75 + w * 0.25, rng.random(n_rocks) * h * 0.25 + h * 0.25], 1)
        rocks = CircleObstacles(env, centers=c, radiuses=rng.random(n_rocks) * 5 + 5, weights=rng.random(n_rocks) * 50 + 50)
        api.perceived_objects.append(rocks)
    api.ants.activate_all_pheromones(np.ones((n_ants, 2)) * 10)
    rng = np.random.default_rng(seed + 1)
    for _ in range(5):
        api.step(rng.integers(-1, 2, n_ants), rng.integers(0, 3, n_ants))
        env.update()
    t0 = time.perf_counter()
    for _ in range(steps):
        api.step(rng.integers(-1, 2, n_ants), rng.integers(0, 3, n_ants))
        env.update()
    return time.perf_counter() - t0


CASES = {  # name: (w, h, ants, rocks, timed steps)
    "c1 (1 env, 64x64, 32 ants)": (64, 64, 32, 0, 400),
    "c2 shape (256x256, 256 ants)": (256, 256, 256, 0, 60),
    "c3 shape (256x256, 512 ants, 8 rocks)": (256, 256, 512, 8, 40),
    "c4 shape (512x512, 1024 ants)": (512, 512, 1024, 0, 15),
}

if __name__ == "__main__":
    cores = len(os.sched_getaffinity(0))
    out = {"nproc": cores, "python": sys.version.split()[0], "cases": {}}
    for name, (w, h, n, r, steps) in CASES.items():
        one = run_case((w, h, n, r, steps, 3))
        with mp.get_context("spawn").Pool(cores) as pool:
            t0 = time.perf_counter()
            ts = pool.map(run_case, [(w, h, n, r, steps, 10 + i) for i in range(cores)])
            wall = time.perf_counter() - t0
        out["cases"][name] = {
            "one_process_ant_steps_per_s": n * steps / one,
            "one_process_ms_per_step": one / steps * 1e3,
            "all_core_ant_steps_per_s": cores * n * steps / max(ts),
            "all_core_processes": cores,
        }
        print(name, json.dumps(out["cases"][name]))
        del wall
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", "r01",
                       "reference_cpu_timing.json")
    json.dump(out, open(dst, "w"), indent=1)
