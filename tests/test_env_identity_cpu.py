"""The oracle's environment-keyed random streams take the GLOBAL environment id (AntsCfg.env_id_base, ABI 5): an oracle
over the environments [lo, hi) of a batch with env_id_base = lo restates rows lo:hi of the whole-batch oracle — built-in
wall jitter and the counter-based episode generator alike.  (The device side of the same statement: -m gpu,
tests/test_gpu_shard_identity.py.)"""
import numpy as np

from antsrl_amd import config as cm
from antsrl_amd.synth import random_actions, synth_init
from oracle.oracle import Oracle, generate_init, jitter_u01


def test_oracle_shard_equals_rows_of_the_whole_batch():
    E, N, W, H, steps = 7, 48, 32, 40, 8
    kw = dict(n_rocks=2, deposit_strength=256.0, rng_seed=99)
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=2, wall_density=0.15, n_food_discs=3, food_rmin=2, food_rmax=4)
    rot, ph = random_actions(cfg, steps, seed=4)
    whole = Oracle(cfg, init)
    parts = []
    for lo, hi in ((0, 3), (3, 4), (4, 7)):
        sub = {k: np.ascontiguousarray(v[lo:hi]) for k, v in init.items()}
        parts.append((lo, hi, Oracle(cm.make_cfg(hi - lo, N, W, H, env_id_base=lo, n_envs_total=E, **kw), sub)))
    lo_f, hi_f = 4, 7
    forgot = Oracle(cm.make_cfg(hi_f - lo_f, N, W, H, **kw), {k: np.ascontiguousarray(v[lo_f:hi_f]) for k, v in init.items()})
    for t in range(steps):
        ow = whole.step(rot[t], ph[t])
        hw = whole.update(None)
        for lo, hi, orc in parts:
            op = orc.step(rot[t][lo:hi], ph[t][lo:hi])
            hp = orc.update(None)
            for a, b in zip(ow, op):
                np.testing.assert_array_equal(a[lo:hi], b)
            np.testing.assert_array_equal(hw[lo:hi], hp)
        forgot.step(rot[t][lo_f:hi_f], ph[t][lo_f:hi_f])
        forgot.update(None)
    for lo, hi, orc in parts:
        for name in ("x", "y", "theta", "holding", "phero", "food", "explored", "rock_cx", "rock_cy"):
            np.testing.assert_array_equal(getattr(whole, name)[lo:hi], getattr(orc, name), err_msg=name)
    assert not np.array_equal(whole.theta[lo_f:hi_f], forgot.theta), "no wall hit: the jitter's env key was not exercised"


def test_oracle_generator_takes_the_global_id():
    E, N, W, H = 6, 20, 32, 32
    cfg = cm.make_cfg(E, N, W, H, n_rocks=1)
    gen = cm.make_gen(0.1, 3, 2, 4)
    whole = generate_init(cfg, gen, 11)
    for lo, hi in ((0, 2), (2, 6)):
        part = generate_init(cm.make_cfg(hi - lo, N, W, H, n_rocks=1, env_id_base=lo), gen, 11)
        for k in whole:
            np.testing.assert_array_equal(whole[k][lo:hi], part[k], err_msg=k)
    assert not np.array_equal(whole["anthill_xyr"][0], whole["anthill_xyr"][1]) or not np.array_equal(whole["walls"][0], whole["walls"][1])


def test_jitter_key_is_the_documented_function_of_the_global_id():
    assert jitter_u01(5, 1000, 3, 7) != jitter_u01(5, 0, 3, 7)
    assert jitter_u01(5, 1000, 3, 7) == jitter_u01(5, 1000, 3, 7)
