"""Randomised configurations against the oracle: perception radius 1..7 (one and several passes), random
masks, channel lists in any order with repeats (generic layout), 1..4 pheromone channels, 0..4 rocks,
every reward kind, diffusion filters of radius 0..2 in both pheromone modes, ragged grids and ant
counts, forward shifts.  Each case is a few steps of step + update with injected wall-jitter draws."""
import numpy as np
import pytest

from test_gpu_parity import _anthill_dist, _check_reward, _compare_with_oracle, _cpu, check_obs, torch_mod  # noqa: F401  (the parity suite's checks and fixture)

pytestmark = pytest.mark.gpu


def _random_case(rng):
    from antsrl_amd import config as cm
    E = int(rng.integers(1, 4))
    N = int(rng.choice([1, 7, 64, 65, 130, 257, 600, 1025, 1100]))
    W, H = int(rng.integers(12, 90)), int(rng.integers(12, 90))
    r = int(rng.integers(1, 8))
    n_phero = int(rng.choice([2, 2, 2, 1, 3, 4]))
    n_rocks = int(rng.integers(0, 5))
    kinds_pool = [cm.CH_ANTS, cm.CH_ANTHILL, cm.CH_WALLS, cm.CH_FOOD] + [cm.CH_PHERO] * 2 + ([cm.CH_ROCKS] if n_rocks else [])
    K = int(rng.integers(1, 10))
    channels = []
    for _ in range(K):
        k = int(rng.choice(kinds_pool))
        channels.append((k, int(rng.integers(0, n_phero)) if k == cm.CH_PHERO else 0))
    if rng.random() < 0.35:
        channels = None  # the generator's default layout
    mask = None if rng.random() < 0.3 else (rng.random((2 * r + 1, 2 * r + 1)) < 0.8)
    fr = int(rng.choice([0, 0, 1, 2]))
    if fr == 0:
        filt = np.array([[float(rng.choice([0.999, 0.9, 1.0]))]])
    else:
        filt = rng.random((2 * fr + 1, 2 * fr + 1))
        if rng.random() < 0.4:  # rank-1: the separable form of the stencil
            filt = np.outer(rng.random(2 * fr + 1), rng.random(2 * fr + 1))
        filt = filt / filt.sum() * float(rng.choice([0.999, 0.95]))
    kw = dict(n_phero=n_phero, n_rocks=n_rocks, mask=mask, perception_radius=r, channels=channels,
              fwd_delta=float(rng.choice([0.0, 4.0, 2.5])), deposit_strength=float(rng.choice([1.0, 256.0])),
              filt=filt, reward_kind=int(rng.integers(0, 4)), fct_explore_holding=float(rng.choice([0.0, 0.5])),
              phero_mode=int(rng.integers(0, 2)), max_time=int(rng.choice([3, 2000])),
              delta=float(rng.choice([1.1, 1.1, 1.0, 1.37])), max_speed=float(rng.choice([1.0, 2.5])),
              max_hold=float(rng.choice([5.0, 2.0])))
    if channels is not None and not any(k == cm.CH_PHERO for k, _ in channels) and rng.random() < 0.5:
        kw["phero_max_val"] = None  # Pheromone(max_val=None): legal as long as no pheromone is perceived
    return E, N, W, H, kw


# a longer one-off campaign:  ANTSRL_FUZZ_BASE=64 ANTSRL_FUZZ_CASES=2000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
import os
_BASE, _CASES = int(os.environ.get("ANTSRL_FUZZ_BASE", "0")), int(os.environ.get("ANTSRL_FUZZ_CASES", "64"))


@pytest.mark.parametrize("seed", range(_BASE, _BASE + _CASES))
def test_random_configuration_vs_oracle(torch_mod, seed):
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    rng = np.random.default_rng(1000 + seed)
    E, N, W, H, kw = _random_case(rng)
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=seed, n_food_discs=4, food_rmin=1, food_rmax=4, wall_density=0.08)
    if cfg.n_phero == 2:
        _compare_with_oracle(torch_mod, cfg, init, steps=5, seed=seed, jitter_mode="injected")
        return
    # pheromone actions need exactly two channels (ants.py:89-96): rotation only, activation set directly
    env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init)
    env.reset(init)
    act = rng.choice([0.0, 10.0], size=(E, N, cfg.n_phero)).astype(np.float32)
    env.set_activation(act)
    orc.set_activation(act)
    prev_dist = _anthill_dist(init, orc.ants_xyt)
    for t in range(5):
        rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
        jit = rng.random((E, N))
        obs, ast, rew, done = env.step(rot, None)
        o_obs, o_ast, o_rew, o_done = orc.step(rot, None)
        for e in range(E):
            check_obs(cfg, _cpu(obs)[e], o_obs[e], "seed %d step %d env %d" % (seed, t, e))
        _check_reward(cfg, init, _cpu(rew), o_rew, orc.ants_xyt, prev_dist, "seed %d step %d" % (seed, t))
        prev_dist = _anthill_dist(init, orc.ants_xyt)
        np.testing.assert_array_equal(_cpu(done), o_done)
        env.update(jit)
        orc.update(jit)
    from helpers import phero_close
    assert phero_close(_cpu(env.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold).all()
    np.testing.assert_allclose(_cpu(env.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed", range(100, 124))
def test_random_configuration_fused_calls_and_standalone_observation(torch_mod, seed):
    """The same random configurations driven the way main.py does: antsrl_step_update (one call per
    step), a standalone observation before the first step and after some updates (main.py:88), and a
    step without pheromone actions."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    rng = np.random.default_rng(1000 + seed)
    E, N, W, H, kw = _random_case(rng)
    kw["n_phero"] = 2
    if kw["channels"] is not None:
        kw["channels"] = [(k, a % 2) for k, a in kw["channels"]]
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=seed, n_food_discs=4, food_rmin=1, food_rmax=4, wall_density=0.08)
    env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init)
    env.reset(init)

    def same_obs(got, want, ctx):
        g = _cpu(got)
        for e in range(E):
            check_obs(cfg, g[e], want[e], "seed %d %s env %d" % (seed, ctx, e))

    prev_dist = [_anthill_dist(init, orc.ants_xyt)]

    def same_reward(got, want, ctx):  # (bit-exact but for heading-term ties, see _check_reward)
        _check_reward(cfg, init, _cpu(got), want, orc.ants_xyt, prev_dist[0], "seed %d %s" % (seed, ctx))
        prev_dist[0] = _anthill_dist(init, orc.ants_xyt)

    obs, ast, rew = env.observe()
    o_obs, o_ast, o_rew = orc.observe()
    same_obs(obs, o_obs, "first observation")
    same_reward(rew, o_rew, "first observation")
    for t in range(6):
        rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
        ph = rng.integers(0, 3, (E, N), dtype=np.int8) if t != 3 else None
        jit = rng.random((E, N))
        obs, ast, rew, done = env.step_update(rot, ph, jit)
        o_obs, o_ast, o_rew, o_done = orc.step(rot, ph)
        same_obs(obs, o_obs, "step %d" % t)
        np.testing.assert_array_equal(_cpu(ast), o_ast.astype(np.float32))
        same_reward(rew, o_rew, "step %d" % t)
        np.testing.assert_array_equal(_cpu(done), o_done)
        orc.update(jit)
        if t in (1, 4):
            obs, ast, rew = env.observe()
            o_obs, o_ast, o_rew = orc.observe()
            same_obs(obs, o_obs, "observation after update %d" % t)
            same_reward(rew, o_rew, "observation after update %d" % t)
    from helpers import phero_close
    assert phero_close(_cpu(env.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold).all()
    np.testing.assert_allclose(_cpu(env.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), orc.food)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_FOOD)), orc.anthill_food)


@pytest.mark.parametrize("seed", range(200, 208))
def test_random_configuration_on_grids_past_the_lds_limit(torch_mod, seed):
    """The random configurations on grids of 0.7–1.2 M cells: presence / explored maps in HBM scratch
    (k_act<..., BIG>), every reward kind, masks, channel lists, rocks, both pheromone modes at radius 0."""
    from antsrl_amd import config as cm
    from antsrl_amd.synth import synth_init
    rng = np.random.default_rng(3000 + seed)
    E, N, W, H, kw = _random_case(rng)
    E, N = 1, int(rng.choice([1, 40, 130, 600]))
    W, H = int(rng.integers(700, 1100)), int(rng.integers(900, 1100))
    kw["n_phero"] = 2
    if kw["channels"] is not None:
        kw["channels"] = [(k, a % 2) for k, a in kw["channels"]]
    if np.asarray(kw["filt"]).shape[0] > 1:
        kw["filt"] = np.array([[0.999]])  # keep the oracle's stencil over a million cells out of the test's time
    kw["perception_radius"] = min(kw["perception_radius"], 5)
    if kw["mask"] is not None:
        p = 2 * kw["perception_radius"] + 1
        kw["mask"] = rng.random((p, p)) < 0.8
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=seed, n_food_discs=6, food_rmin=2, food_rmax=6, wall_density=0.05)
    _compare_with_oracle(torch_mod, cfg, init, steps=4, seed=seed, jitter_mode="injected")


@pytest.mark.parametrize("seed", range(_BASE + 300, _BASE + 300 + max(32, _CASES // 2)))
def test_random_configuration_library_jitter_no_reads_between_steps(torch_mod, seed):
    """The random configurations driven the way bench.py does: one antsrl_step_update per step with the library's own
    wall jitter and NO state read in between, so that every update that can be deferred into the next step is
    (k_update_move, include/antsrl.h "DEFERRED UPDATE"); outputs of every step and the final state against the oracle,
    which implements the same counter-based jitter."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    from helpers import phero_close
    rng = np.random.default_rng(5000 + seed)
    E, N, W, H, kw = _random_case(rng)
    kw["n_phero"] = 2
    if kw["channels"] is not None:
        kw["channels"] = [(k, a % 2) for k, a in kw["channels"]]
    if rng.random() < 0.5:  # the generator's own layout: the cell-meta path where the perception shape allows it
        kw["channels"] = None
        kw["perception_radius"], kw["mask"] = 3, None
        kw.pop("phero_max_val", None)  # (pheromone is perceived again: max_val is needed)
        if seed % 2 == 0:  # ... and every second of these on the product's hot layout: scaled units, interleaved records in
            # blocks of 2 x 4 cells (KP::tiled needs W even, H a multiple of 4), the update deferred into k_update_move
            W, H = W & ~1, H & ~3
            kw["filt"], kw["phero_mode"] = np.array([[float(rng.choice([0.999, 0.9]))]]), cm.PHERO_AUTO
        elif seed % 4 == 1:  # ... a quarter on the explicit-sweep layout with its {food, META} records in blocks of 4 x 4 cells
            # (KP::ftile needs W and H multiples of 4): a second gather per cell, separate pheromone buffers
            W, H = max(W & ~3, 12), max(H & ~3, 12)
            kw["phero_mode"] = cm.PHERO_EXPLICIT_SWEEP
    # round 4: the shard's global env ids key the library's jitter (AntsCfg.env_id_base) — device and oracle alike —, and a
    # third of the cell-meta cases write padded observation rows (antsrl_set_obs_row_stride)
    kw["env_id_base"] = int(rng.integers(0, 1 << 20)) if rng.random() < 0.7 else 0
    padded = rng.random() < 0.33
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=seed, n_food_discs=4, food_rmin=1, food_rmax=4, wall_density=0.08)
    env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init)
    if padded and env.query(cm.Q_CELL_META):
        env = BatchedAntsEnv(cfg, obs_row_stride="line")
    env.reset(init)
    prev_dist = _anthill_dist(init, orc.ants_xyt)
    for t in range(7):
        rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
        ph = rng.integers(0, 3, (E, N), dtype=np.int8) if t != 4 else None
        obs, ast, rew, done = env.step_update(rot, ph, None)
        o_obs, o_ast, o_rew, o_done = orc.step(rot, ph)
        g = _cpu(obs)
        for e in range(E):
            check_obs(cfg, g[e], o_obs[e], "seed %d step %d env %d" % (seed, t, e))
        np.testing.assert_array_equal(_cpu(ast), o_ast.astype(np.float32))
        _check_reward(cfg, init, _cpu(rew), o_rew, orc.ants_xyt, prev_dist, "seed %d step %d" % (seed, t))
        prev_dist = _anthill_dist(init, orc.ants_xyt)
        np.testing.assert_array_equal(_cpu(done), o_done)
        orc.update(None)
    assert phero_close(_cpu(env.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold).all()
    np.testing.assert_allclose(_cpu(env.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), orc.food)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_FOOD)), orc.anthill_food)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_EXPLORED)), orc.explored)
