"""Replays one case of tests/test_gpu_fuzz.py and prints where the device and the oracle part (debug aid).
usage: python profiles/fuzz_case_debug.py SEED [REPEATS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_fuzz as f
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init, random_actions
from oracle.oracle import Oracle

seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for rep in range(reps):
    rng = np.random.default_rng(1000 + seed)
    E, N, W, H, kw = f._random_case(rng)
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=seed, n_food_discs=4, food_rmin=1, food_rmax=4, wall_density=0.08)
    env = BatchedAntsEnv(cfg); env.reset(init)
    orc = Oracle(cfg, init, n_threads=4)
    steps = 5
    rot, ph = random_actions(cfg, steps, seed)
    r2 = np.random.default_rng(seed)
    for t in range(steps):
        obs, ast, rew, done = env.step(rot[t], ph[t])
        o_obs, o_ast, o_rew, o_done = orc.step(rot[t], ph[t])
        d = rew.cpu().numpy(); w = o_rew.astype(np.float32)
        bad = np.argwhere(d != w)
        for e, i in bad:
            xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()[e, i]
            print("rep %d step %d env %d ant %d: dev %.6f oracle %.6f hold dev %s orc %s xyt dev %r orc %r anthill %r" % (
                rep, t, e, i, d[e, i], w[e, i], ast.cpu().numpy()[e, i, 0], o_ast[e, i, 0], xyt.tolist(), orc.ants_xyt[e, i].tolist(),
                init["anthill_xyr"][e].tolist()))
            ax, ay = init["anthill_xyr"][e][:2]
            print("   dist dev %.17g orc %.17g" % (np.hypot(xyt[0] - ax, xyt[1] - ay), np.hypot(orc.ants_xyt[e, i, 0] - ax, orc.ants_xyt[e, i, 1] - ay)))
        if len(sys.argv) > 3:
            a = int(sys.argv[3])
            print("  t=%d after step : dev %r orc %r" % (t, env.read_state(cm.S_ANTS_XYT).cpu().numpy()[0, a].tolist(), orc.ants_xyt[0, a].tolist()))
        jit = r2.random((E, cfg.n_ants))
        env.update(jit); orc.update(jit)
        if len(sys.argv) > 3:
            print("  t=%d after update: dev %r orc %r  rot %d" % (t, env.read_state(cm.S_ANTS_XYT).cpu().numpy()[0, a].tolist(), orc.ants_xyt[0, a].tolist(), rot[t][0, a]))
    print("rep %d done, mismatches at last step: %d" % (rep, len(bad)))
