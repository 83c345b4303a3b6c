"""Replays one n_phero != 2 case of tests/test_gpu_fuzz.py::test_random_configuration_vs_oracle (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_fuzz as f
from test_gpu_parity import _anthill_dist
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from oracle.oracle import Oracle

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
E, N, W, H, kw = f._random_case(rng)
cfg = cm.make_cfg(E, N, W, H, **kw)
init = synth_init(cfg, seed=seed, n_food_discs=4, food_rmin=1, food_rmax=4, wall_density=0.08)
assert cfg.n_phero != 2
env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init)
env.reset(init)
act = rng.choice([0.0, 10.0], size=(E, N, cfg.n_phero)).astype(np.float32)
env.set_activation(act); orc.set_activation(act)
prev = _anthill_dist(init, orc.ants_xyt)
for t in range(5):
    rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
    jit = rng.random((E, N))
    obs, ast, rew, done = env.step(rot, None)
    o_obs, o_ast, o_rew, o_done = orc.step(rot, None)
    d = rew.cpu().numpy(); w = o_rew.astype(np.float32)
    nd = _anthill_dist(init, orc.ants_xyt)
    dx = env.read_state(cm.S_ANTS_XYT).cpu().numpy()
    for e, i in np.argwhere(d != w):
        print("step %d env %d ant %d: dev %.6f oracle %.6f (f64 %.17g) hold %s/%s prev_dist %.17g new %.17g  dev xyt %r orc %r anthill %r" % (
            t, e, i, d[e, i], w[e, i], o_rew[e, i], ast.cpu().numpy()[e, i, 0], o_ast[e, i, 0], prev[e, i], nd[e, i], dx[e, i].tolist(), orc.ants_xyt[e, i].tolist(), init["anthill_xyr"][e].tolist()))
    prev = nd
    env.update(jit); orc.update(jit)
