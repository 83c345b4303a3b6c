"""Wall time per step of the reference-shaped Python surface driven like main.py:98-131 (one env, 32 ants,
64x64: BASELINE configs[0]): api.step(actions) -> numpy observation, then env.update().

    gpurun -- 'python3 profiles/rlapi_latency.py'
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from antsrl_amd.generator import EnvironmentGenerator
from antsrl_amd.rl_api import ExplorationReward, RLApi
from test_generator import BernoulliWalls, FoodNearAnthill

def run(as_numpy, n_ants=32, w=64, h=64, steps=2000):
    rng = np.random.default_rng(7)
    api = RLApi(reward=ExplorationReward(), reward_threshold=1, max_speed=1, max_rot_speed=40 / 180 * np.pi,
                carry_speed_reduction=0.05, backward_speed_reduction=0.5, as_numpy=as_numpy)
    gen = EnvironmentGenerator(w, h, n_ants, 2, 0, FoodNearAnthill(6, 3, 6, (w // 2, h // 2, 6)),
                               BernoulliWalls(0.05, rng), 1 << 30, seed=3)
    env = gen.generate(api)
    api.observation()
    rot = rng.integers(-1, 2, (64, n_ants)).astype(np.int8)
    ph = rng.integers(0, 3, (64, n_ants)).astype(np.int8)
    if not as_numpy:
        dev = torch.device("cuda", 0)
        rot, ph = torch.from_numpy(rot).to(dev), torch.from_numpy(ph).to(dev)
    for t in range(100):
        api.step(rot[t % 64], ph[t % 64]); env.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        obs, ast, rew, done = api.step(rot[t % 64], ph[t % 64])
        env.update()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

for as_numpy in (True, False):
    print("RLApi.step + Environment.update, 1 env x 32 ants, 64x64, %s in/out: %.1f us per step" % (
        "numpy" if as_numpy else "device tensors", run(as_numpy)))
