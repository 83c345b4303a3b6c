"""GPU parity tests: the HIP path (through the C-ABI) against the reference's golden vectors
and against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): ant cell indices, holding / pickup counts, mandibles, food,
anthill.food, explored map, integer perception channels, reward and done BIT-EXACT; ant
coordinates within 1e-9 (device sin/cos may differ from glibc in the last ulp); pheromone grid
and pheromone perception channels within 1e-5 (fp32 grid vs float64 reference, comparator
`helpers.phero_close`).
"""
import numpy as np
import pytest

import helpers
from helpers import OP_OBSERVE, OP_STEP, OP_UPDATE, assert_xy_close, fixture_names, load_fixture, phero_close

pytestmark = pytest.mark.gpu

XY_ATOL = 1e-9


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _cpu(t):
    return t.detach().cpu().numpy()


def check_obs(cfg, got, want, ctx):
    """got f32 [N,P,P,K], want f64 [N,P,P,K]."""
    from antsrl_amd import config as cm
    for k in range(cfg.n_channels):
        g, w = got[..., k].astype(np.float64), want[..., k]
        if cfg.channel_kind[k] == cm.CH_PHERO:
            # masked cells are exactly -1 on both sides
            np.testing.assert_array_equal(g == -1.0, w == -1.0, err_msg=ctx + " mask ch%d" % k)
            # value / max_val (RL_api.py:124-125): the grid's relative error plus one reciprocal multiply — the same relative
            # tolerance as the grid comparator (helpers.PHERO_RTOL, 4 x the largest error measured), no absolute slack to
            # speak of (until round 4: 1e-5 absolute on values of which the smallest non-zero one is 0.01 / 255 = 3.9e-5)
            ok = np.abs(g - w) <= 1e-10 + helpers.PHERO_RTOL * np.abs(w)
            # a cell at the 0.01 cut (pheromone.py:45) may read 0 on one side
            thr = cfg.phero_threshold / cfg.phero_max_val
            ok |= (np.minimum(g, w) == 0) & (np.abs(np.maximum(g, w) - thr) <= helpers.CUT_BAND_RTOL * thr)
            if helpers._ERR_LOG:
                nz = (w > 0) & ok & (np.minimum(g, w) > 0)
                helpers._log_err("obs_phero", max_rel=float((np.abs(g - w)[nz] / w[nz]).max()) if nz.any() else 0.0, ctx=ctx[:80])
            assert ok.all(), "%s phero channel %d: %d cells off, max |d|=%g" % (
                ctx, k, (~ok).sum(), np.abs(g - w)[~ok].max())
        else:
            np.testing.assert_array_equal(g, w, err_msg=ctx + " channel %d" % k)


def check_state(env, cfg, F, t, meta, ctx, envs, with_phero):
    from antsrl_amd import config as cm
    xyt = _cpu(env.read_state(cm.S_ANTS_XYT))
    prev = _cpu(env.read_state(cm.S_PREV_XY))
    hold = _cpu(env.read_state(cm.S_HOLDING))
    mand = _cpu(env.read_state(cm.S_MANDIBLES))
    act = _cpu(env.read_state(cm.S_ACTIVATION))
    rs = _cpu(env.read_state(cm.S_REWARD_STATE))
    food = _cpu(env.read_state(cm.S_FOOD))
    ts = _cpu(env.read_state(cm.S_TIMESTEP))
    af = _cpu(env.read_state(cm.S_ANTHILL_FOOD))
    expl = _cpu(env.read_state(cm.S_EXPLORED))
    rc = _cpu(env.read_state(cm.S_ROCK_CENTERS)) if cfg.n_rocks else None
    ph = _cpu(env.read_state(cm.S_PHERO)) if with_phero else None
    for e in envs:
        assert_xy_close(xyt[e], F["ants"][t], XY_ATOL, ctx)
        np.testing.assert_array_equal(np.floor(xyt[e][:, :2]), np.floor(F["ants"][t][:, :2]), err_msg=ctx + " cells")
        assert_xy_close(prev[e], F["prev"][t], XY_ATOL, ctx)
        np.testing.assert_array_equal(hold[e], F["holding"][t], err_msg=ctx + " holding")
        np.testing.assert_array_equal(mand[e], F["mandibles"][t], err_msg=ctx + " mandibles")
        np.testing.assert_array_equal(act[e], F["activation"][t], err_msg=ctx + " activation")
        np.testing.assert_array_equal(rs[e], F["reward_state"][t], err_msg=ctx + " reward_state")
        np.testing.assert_array_equal(food[e], F["food"][t], err_msg=ctx + " food")
        assert ts[e] == F["timestep"][t], ctx
        assert af[e] == F["anthill_food"][t], ctx + " anthill_food"
        if meta["reward"] in ("exploration", "all"):
            np.testing.assert_array_equal(expl[e], F["explored"][t], err_msg=ctx + " explored")
        if cfg.n_rocks:
            assert_xy_close(rc[e], F["rock_centers"][t], XY_ATOL, ctx)
        if with_phero:
            ok = phero_close(ph[e], F["phero"][t], threshold=cfg.phero_threshold)
            assert ok.all(), "%s pheromone: %d cells off, max |d|=%g" % (
                ctx, (~ok).sum(), np.abs(ph[e] - F["phero"][t])[~ok].max())


def _modes():
    from antsrl_amd import config as cm
    return [cm.PHERO_AUTO, cm.PHERO_EXPLICIT_SWEEP]


@pytest.mark.parametrize("act", [1, 2], ids=["cell_meta", "single_kernel"])
@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "explicit_sweep"])
@pytest.mark.parametrize("name", fixture_names())
def test_hip_matches_reference_golden(torch_mod, name, mode, act):
    """Replay every recorded reference run through antsrl_step / antsrl_update / antsrl_observe,
    with the scaled pheromone units (auto) and with the explicit per-step sweep, on BOTH kernel paths
    (AntsCfg.act_path: k_move + k_perceive on the cell-meta layout, and the single k_act)."""
    from antsrl_amd import _lib
    from antsrl_amd.batched import BatchedAntsEnv
    n_envs = 3
    cfg, init, F, meta = load_fixture(name, n_envs)
    if mode == 1 and cfg.filter_radius != 0:
        pytest.skip("filters with a radius always use the tiled sweep")
    cfg.phero_mode = mode
    cfg.act_path = act
    try:
        env = BatchedAntsEnv(cfg)
    except _lib.AntsrlError as e:
        if act == 1 and "ANTSRL_ACT_CELL_META" in str(e):
            pytest.skip("a perception shape the cell-meta path does not take")
        raise
    assert env.query(0) == (1 if act == 1 else 0)
    env.reset(init)
    from antsrl_amd import config as cm
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_AREA))[1], F["init_anthill_area"].astype(np.uint8))
    if meta["deposit_strength"] != 1.0:
        env.set_activation(np.broadcast_to(F["init_activation"], (n_envs,) + F["init_activation"].shape).copy(),
                           meta["deposit_strength"])
    envs = (0, n_envs - 1)
    for t, op in enumerate(F["ops"]):
        ctx = "%s op %d kind %d" % (name, t, op)
        if op == OP_STEP:
            rot = np.broadcast_to(F["rot"][t], (n_envs, cfg.n_ants)) if F["has_rot"][t] else None
            ph = np.broadcast_to(F["ph"][t], (n_envs, cfg.n_ants)) if F["has_ph"][t] else None
            obs, ast, rew, done = env.step(rot, ph)
            assert (_cpu(done) == F["done"][t]).all(), ctx
        elif op == OP_OBSERVE:
            obs, ast, rew = env.observe()
        else:
            env.update(np.broadcast_to(F["jitter"][t], (n_envs, cfg.n_ants)).copy())
        if op != OP_UPDATE:
            o, a, r = _cpu(obs), _cpu(ast), _cpu(rew)
            for e in envs:
                check_obs(cfg, o[e], F["obs"][t], ctx)
                np.testing.assert_array_equal(a[e], F["agent_state"][t].astype(np.float32), err_msg=ctx)
                np.testing.assert_array_equal(r[e], F["reward"][t].astype(np.float32), err_msg=ctx + " reward")
        check_state(env, cfg, F, t, meta, ctx, envs, bool(F["stored"][t]))


def test_fused_step_update_equals_two_calls(torch_mod):
    """antsrl_step_update == antsrl_step then antsrl_update (the sweep is enqueued first there)."""
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd import config as cm
    cfg, init, F, meta = load_fixture("s03_walls_rocks", 2)
    a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg)
    a.reset(init)
    b.reset(init)
    for t in range(0, 40, 2):
        rot = np.broadcast_to(F["rot"][t], (2, cfg.n_ants))
        ph = np.broadcast_to(F["ph"][t], (2, cfg.n_ants))
        jit = np.broadcast_to(F["jitter"][t + 1], (2, cfg.n_ants)).copy()
        oa = [x.clone() for x in a.step(rot, ph)]
        a.update(jit)
        ob = b.step_update(rot, ph, jit)
        for x, y in zip(oa, ob):
            assert torch_mod.equal(x, y)
    for which in (cm.S_ANTS_XYT, cm.S_PHERO, cm.S_FOOD, cm.S_EXPLORED, cm.S_ANTHILL_FOOD, cm.S_ROCK_CENTERS):
        assert torch_mod.equal(a.read_state(which), b.read_state(which)), which
    check_state(b, cfg, F, 39, meta, "fused", (0, 1), True)


def _anthill_dist(init, xyt):
    """All_Rewards.compute_distance (reward_custom.py:59-60) of every ant, float64 [E, N]."""
    c = np.asarray(init["anthill_xyr"], np.float64)
    return ((xyt[..., 0] - c[:, None, 0]) ** 2 + (xyt[..., 1] - c[:, None, 1]) ** 2) ** 0.5


def _check_reward(cfg, init, got, want, xyt, prev_dist, ctx):
    """Rewards are bit-exact, with ONE documented exception: All_Rewards' heading term `previous_dist > new_dist`
    (reward_custom.py:100) on a mathematical TIE.  Ants that start on the anthill centre (radius 0) and turn by a
    constant angle walk a regular polygon through the centre, so vertex k and vertex n - k are exactly equidistant
    from it; which way the comparison falls is then decided by the last bit of sin / cos, where the device's
    `sincos` and the host libm may differ (ant coordinates are held to 1e-9, not to the bit — and numpy's own trig
    differs between CPUs).  Such an ant may be off by exactly the heading term; anything else fails."""
    from antsrl_amd import config as cm
    want32 = want.astype(np.float32)
    bad = got != want32
    if not bad.any():
        return
    assert cfg.reward_kind == cm.REWARD_ALL, "%s: %d rewards differ" % (ctx, bad.sum())
    nd = _anthill_dist(init, xyt)
    # (a last-bit difference of a COORDINATE, whose magnitude is up to the grid size, moves the distance by as much)
    tie = np.abs(prev_dist - nd) <= 4 * np.spacing(float(max(cfg.w, cfg.h)))
    step = np.float32(0.1 * cfg.fct_headinganthill)
    off_by_heading = np.abs(np.abs(got.astype(np.float64) - want) - step) <= 1e-6
    assert (bad <= (tie & off_by_heading)).all(), "%s: %d rewards differ beyond heading-term ties" % (
        ctx, (bad & ~(tie & off_by_heading)).sum())


def _compare_with_oracle(torch_mod, cfg, init, steps, seed, jitter_mode):
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd import config as cm
    from antsrl_amd.synth import random_actions
    from oracle.oracle import Oracle
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    orc = Oracle(cfg, init, n_threads=4)
    rot, ph = random_actions(cfg, steps, seed)
    rng = np.random.default_rng(seed)
    E = cfg.n_envs
    prev_dist = _anthill_dist(init, orc.ants_xyt)  # Reward.setup, reward_custom.py:73
    for t in range(steps):
        ctx = "step %d" % t
        obs, ast, rew, done = env.step(rot[t], ph[t])
        o_obs, o_ast, o_rew, o_done = orc.step(rot[t], ph[t])
        go = _cpu(obs)
        for e in range(E):
            check_obs(cfg, go[e], o_obs[e], ctx + " env %d" % e)
        np.testing.assert_array_equal(_cpu(ast), o_ast.astype(np.float32), err_msg=ctx)
        _check_reward(cfg, init, _cpu(rew), o_rew, orc.ants_xyt, prev_dist, ctx)
        prev_dist = _anthill_dist(init, orc.ants_xyt)
        np.testing.assert_array_equal(_cpu(done), o_done, err_msg=ctx)
        jit = rng.random((E, cfg.n_ants)) if jitter_mode == "injected" else None
        env.update(jit)
        orc.update(jit)
        xyt = _cpu(env.read_state(cm.S_ANTS_XYT))
        assert_xy_close(xyt, orc.ants_xyt, XY_ATOL, ctx)
        np.testing.assert_array_equal(np.floor(xyt[..., :2]), np.floor(orc.ants_xyt[..., :2]), err_msg=ctx)
        np.testing.assert_array_equal(_cpu(env.read_state(cm.S_HOLDING)), orc.holding, err_msg=ctx)
        np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), orc.food, err_msg=ctx)
        np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_FOOD)), orc.anthill_food, err_msg=ctx)
        if cfg.w * cfg.h <= 4096:  # small grids: the pheromone grid after every update
            okp = phero_close(_cpu(env.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold)
            assert okp.all(), "%s pheromone: %d cells off" % (ctx, (~okp).sum())
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_EXPLORED)), orc.explored)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_MANDIBLES)), orc.mandibles)
    ok = phero_close(_cpu(env.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold)
    assert ok.all(), "pheromone: %d cells off" % (~ok).sum()
    if cfg.n_rocks:
        assert_xy_close(_cpu(env.read_state(cm.S_ROCK_CENTERS)), orc.rock_centers, XY_ATOL)
    return env, orc


def test_bench_config_vs_oracle(torch_mod):
    """BASELINE config 3 shape (256x256, 512 ants, 8 rocks, walls, food) on a few envs, with the
    built-in counter-based wall jitter (device and oracle implement the same generator)."""
    from antsrl_amd.config import make_cfg
    from antsrl_amd.synth import synth_init
    cfg = make_cfg(6, 512, 256, 256, n_rocks=8, deposit_strength=256.0)
    _compare_with_oracle(torch_mod, cfg, synth_init(cfg, seed=77), steps=12, seed=5, jitter_mode="builtin")


def test_full_bench_batch_sampled_envs_vs_oracle(torch_mod):
    """The FULL BASELINE config-3 batch (1024 envs x 512 ants, 256x256, 8 rocks) stepped on the
    GPU; environments 0, 1, 511, 1022 and 1023 are replayed by the oracle from the same initial
    arrays, actions and wall-jitter draws (envs are independent, so a sample pins the env-major
    addressing at scale).  Also: the run is bit-reproducible, and an env's result does not depend
    on the batch around it."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    from oracle.oracle import Oracle
    E, N, steps, pick = 1024, 512, 6, [0, 1, 511, 1022, 1023]
    cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0)
    cfg_s = cm.make_cfg(len(pick), N, 256, 256, n_rocks=8, deposit_strength=256.0)
    init = synth_init(cfg, seed=1234)
    sub = {k: np.ascontiguousarray(v[pick]) for k, v in init.items()}
    rot, ph = random_actions(cfg, steps, seed=99)
    rng = np.random.default_rng(17)
    orc = Oracle(cfg_s, sub, n_threads=5)
    env, env2, env_s = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg), BatchedAntsEnv(cfg_s)
    env.reset(init)
    env2.reset(init)
    env_s.reset(sub)
    for t in range(steps):
        jit = rng.random((E, N))
        obs, ast, rew, done = env.step_update(rot[t], ph[t], jit)
        o2 = env2.step_update(rot[t], ph[t], jit)
        os_ = env_s.step_update(rot[t][pick], ph[t][pick], jit[pick])
        assert all(torch_mod.equal(a, b) for a, b in zip((obs, ast, rew, done), o2)), "run-to-run difference"
        assert torch_mod.equal(obs[pick], os_[0]) and torch_mod.equal(rew[pick], os_[2]), "batch dependence"
        o_obs, o_ast, o_rew, o_done = orc.step(rot[t][pick], ph[t][pick])
        orc.update(jit[pick])
        go, gr = _cpu(obs[pick]), _cpu(rew[pick])
        for j in range(len(pick)):
            check_obs(cfg_s, go[j], o_obs[j], "full batch step %d env %d" % (t, pick[j]))
        np.testing.assert_array_equal(gr, o_rew.astype(np.float32))
    xyt = _cpu(env.read_state(cm.S_ANTS_XYT))
    assert_xy_close(xyt[pick], orc.ants_xyt, XY_ATOL)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD))[pick], orc.food)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_EXPLORED))[pick], orc.explored)
    assert phero_close(_cpu(env.read_state(cm.S_PHERO))[pick], orc.phero).all()
    # whole-batch invariants: every env moved, stayed in range, kept its books
    assert np.isfinite(xyt).all() and xyt[..., 0].min() >= 0 and xyt[..., 0].max() < 256
    assert (np.abs(xyt[..., :2] - init["ants_xyt"][..., :2]).max(axis=(1, 2)) > 0).all()
    hold = _cpu(env.read_state(cm.S_HOLDING))
    assert hold.min() >= 0 and hold.max() <= 5
    assert (_cpu(env.read_state(cm.S_TIMESTEP)) == steps + 1).all()


def test_full_bench_batch_long_horizon_vs_oracle(torch_mod):
    """300 steps of the full config-3 batch (scaled pheromone units re-used across hundreds of updates,
    ants that have spread, picked food up and brought it home); three sampled environments follow the
    oracle step for step (reward every step, observation and state every 50)."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    E, N, steps, pick = 1024, 512, 300, [3, 600, 1020]
    cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0)
    cfg_s = cm.make_cfg(len(pick), N, 256, 256, n_rocks=8, deposit_strength=256.0)
    init = synth_init(cfg, seed=4321)
    orc = Oracle(cfg_s, {k: np.ascontiguousarray(v[pick]) for k, v in init.items()}, n_threads=3)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    rng = np.random.default_rng(2024)
    for t in range(steps):
        rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
        ph = rng.integers(0, 3, (E, N), dtype=np.int8)
        jit = rng.random((E, N))
        obs, ast, rew, done = env.step_update(rot, ph, jit)
        o_obs, o_ast, o_rew, o_done = orc.step(rot[pick], ph[pick])
        orc.update(jit[pick])
        np.testing.assert_array_equal(_cpu(rew[pick]), o_rew.astype(np.float32), err_msg="step %d" % t)
        if t % 50 == 49 or t == steps - 1:
            go = _cpu(obs[pick])
            for j in range(len(pick)):
                check_obs(cfg_s, go[j], o_obs[j], "long run step %d env %d" % (t, pick[j]))
            np.testing.assert_array_equal(_cpu(ast[pick]), o_ast.astype(np.float32))
            xyt = _cpu(env.read_state(cm.S_ANTS_XYT))[pick]
            assert_xy_close(xyt, orc.ants_xyt, XY_ATOL, "step %d" % t)
            np.testing.assert_array_equal(_cpu(env.read_state(cm.S_HOLDING))[pick], orc.holding)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD))[pick], orc.food)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_EXPLORED))[pick], orc.explored)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_FOOD))[pick], orc.anthill_food)
    assert_xy_close(_cpu(env.read_state(cm.S_ROCK_CENTERS))[pick], orc.rock_centers, XY_ATOL)
    assert phero_close(_cpu(env.read_state(cm.S_PHERO))[pick], orc.phero).all()
    assert orc.anthill_food.sum() > 0 and (orc.holding > 0).any(), "the run should exercise pickup and delivery"


def test_full_config4_shard_sampled_envs_vs_oracle(torch_mod):
    """The FULL per-GPU shard of BASELINE config 4 (1024 envs x 1024 ants, 512x512, radius-3 diffusion:
    explicit sweep kernel, two pheromone buffers, 1024-thread workgroups): three sampled environments
    follow the oracle for three steps; pheromone grids within 1e-5."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    from oracle.oracle import Oracle
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    g = g / g.sum() * (1 - 0.001)
    E, N, steps, pick = 1024, 1024, 3, [0, 517, 1023]
    cfg = cm.make_cfg(E, N, 512, 512, filt=g, deposit_strength=256.0)
    cfg_s = cm.make_cfg(len(pick), N, 512, 512, filt=g, deposit_strength=256.0)
    init = synth_init(cfg, seed=77)
    orc = Oracle(cfg_s, {k: np.ascontiguousarray(v[pick]) for k, v in init.items()}, n_threads=3)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    del init
    rot, ph = random_actions(cfg, steps, seed=5)
    rng = np.random.default_rng(41)
    for t in range(steps):
        jit = rng.random((E, N))  # injected draws: the built-in generator is keyed on the env's index in ITS batch
        obs, ast, rew, done = env.step_update(rot[t], ph[t], jit)
        o_obs, o_ast, o_rew, o_done = orc.step(rot[t][pick], ph[t][pick])
        orc.update(jit[pick])
        go = _cpu(obs[pick])
        for j in range(len(pick)):
            check_obs(cfg_s, go[j], o_obs[j], "config 4 step %d env %d" % (t, pick[j]))
        np.testing.assert_array_equal(_cpu(rew[pick]), o_rew.astype(np.float32))
    xyt = _cpu(env.read_state(cm.S_ANTS_XYT))
    assert_xy_close(xyt[pick], orc.ants_xyt, XY_ATOL)
    assert np.isfinite(xyt).all() and xyt[..., :2].min() >= 0 and xyt[..., :2].max() < 512
    phero = env.read_state(cm.S_PHERO)  # 2 GiB on the device; only the sampled envs come to the host
    food = env.read_state(cm.S_FOOD)
    for j, e in enumerate(pick):
        ph_e = _cpu(phero[e])
        assert phero_close(ph_e, orc.phero[j]).all() and ph_e.max() > 0
        np.testing.assert_array_equal(_cpu(food[e]), orc.food[j])


def test_config2_vs_oracle_injected_jitter(torch_mod):
    from antsrl_amd.config import make_cfg
    from antsrl_amd.synth import synth_init
    cfg = make_cfg(4, 256, 256, 256, n_rocks=0)
    _compare_with_oracle(torch_mod, cfg, synth_init(cfg, seed=3), steps=10, seed=6, jitter_mode="injected")


def test_config4_shape_radius3_vs_oracle(torch_mod):
    """512x512 grid, 1024 ants, radius-3 diffusion filter (BASELINE config 4 per-env shape)."""
    from antsrl_amd.config import make_cfg
    from antsrl_amd.synth import synth_init
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    g = g / g.sum() * 0.999
    cfg = make_cfg(2, 1024, 512, 512, filt=g, deposit_strength=256.0)
    _compare_with_oracle(torch_mod, cfg, synth_init(cfg, seed=11), steps=6, seed=8, jitter_mode="builtin")


def _stencil_filter(radius, separable, seed):
    rng = np.random.default_rng(seed)
    if separable:
        f = np.outer(0.2 + rng.random(2 * radius + 1), 0.2 + rng.random(2 * radius + 1))  # asymmetric rank-1
    else:
        f = 0.1 + rng.random((2 * radius + 1, 2 * radius + 1))
    return f / f.sum() * 0.97


@pytest.mark.parametrize("radius,separable", [(1, False), (1, True), (2, True), (2, False), (3, True), (3, False)])
@pytest.mark.parametrize("W,H", [
    (40, 40),      # one strip, one segment, most lanes outside the grid
    (33, 64),      # odd W: a ragged last row segment
    (70, 122),     # H just past one two-column strip of radius 3 (120 outputs) and just under radius 1's (128)
    (37, 250),     # several strips, the last one ragged
    (100, 41),     # odd H: the one-column march
    (131, 256),    # W not a multiple of any segment length
])
def test_stencil_shapes_vs_oracle(torch_mod, radius, separable, W, H):
    """Every marching stencil kernel (one column / two columns per lane, general / rank-1 filter) on grid shapes
    that exercise the strip and segment edges: walls zero the stencil's input (walls.py:30), zero fill outside the
    grid (pheromone.py:44), the 0.01 cut (:45).  Asymmetric filters, so a transposed or mirrored tap shows."""
    from antsrl_amd.config import make_cfg
    from antsrl_amd.synth import synth_init
    cfg = make_cfg(2, 96, W, H, filt=_stencil_filter(radius, separable, 10 * radius + separable), deposit_strength=256.0)
    _compare_with_oracle(torch_mod, cfg, synth_init(cfg, seed=W + H), steps=5, seed=radius, jitter_mode="builtin")


@pytest.mark.parametrize("E,N,W,H,rocks", [(6, 512, 256, 256, 8), (3, 100, 64, 48, 2), (2, 1024, 128, 128, 0), (5, 37, 40, 40, 3),
                                           (16, 64, 48, 48, 2), (32, 100, 40, 56, 0)])  # (multiples of 16 envs: split-batch candidates)
@pytest.mark.parametrize("explicit", [None, "sweep0", "3x3", "radius3"], ids=["scaled", "explicit_sweep", "diffuse3x3", "radius3"])
def test_deferred_update_is_bit_identical(torch_mod, E, N, W, H, rocks, explicit):
    """antsrl_update with the library's own wall jitter is deferred into the next step (k_update_move, include/antsrl.h
    "DEFERRED UPDATE").  Handle A runs the loop the way bench.py and main.py do (the update kernel always rides with
    the next move); handle B reads a state array after every update, which enqueues the deferred update at once
    (k_update_one, then k_move on its own at the next step).  Every output of every step and the final state must be
    the same bit for bit — and equal to the oracle's.  Steps without rotation / pheromone actions and a standalone
    observation (which also flushes) are mixed in."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    from oracle.oracle import Oracle
    kw = {}
    if explicit is not None:  # round 4: the update is deferred under an explicit sweep, too (the step's sweep then follows its kernels)
        if W * H > 16384 and explicit != "sweep0":
            pytest.skip("the oracle's float64 stencil over a 256 x 256 grid: covered at the smaller shapes")
        kw["phero_mode"] = cm.PHERO_EXPLICIT_SWEEP
        if explicit == "3x3":
            kw["filt"] = cm.diffuse_filter(0.02, 0.001)
        elif explicit == "radius3":
            ax = np.arange(-3, 4)
            g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
            kw["filt"] = g / g.sum() * 0.999
    cfg = cm.make_cfg(E, N, W, H, n_rocks=rocks, deposit_strength=256.0, reward_kind=cm.REWARD_ALL, act_path=cm.ACT_CELL_META, **kw)
    init = synth_init(cfg, seed=31 + N, wall_density=0.08)
    a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg)
    if a.query(cm.Q_DEFERRED_UPDATE) != 1:
        pytest.skip("updates are not deferred (ANTSRL_NO_DEFER_UPDATE in the profiling library)")
    a.reset(init)
    b.reset(init)
    orc = Oracle(cfg, init, n_threads=4)
    steps = 14
    rot, ph = random_actions(cfg, steps, seed=3)
    for t in range(steps):
        r = rot[t] if t != 5 else None
        q = ph[t] if t != 8 else None
        if t % 2 == 0:  # one call per step / the reference's two calls
            oa = [x.clone() for x in a.step_update(r, q, None)]
            ob = [x.clone() for x in b.step_update(r, q, None)]
        else:
            oa = [x.clone() for x in a.step(r, q)]
            a.update(None)
            ob = [x.clone() for x in b.step(r, q)]
            b.update(None)
        b.read_state(cm.S_TIMESTEP)  # B: the deferred update is enqueued now, on its own
        oo = orc.step(r, q)
        orc.update(None)
        for x, y in zip(oa, ob):
            assert torch_mod.equal(x, y), "step %d" % t
        np.testing.assert_array_equal(_cpu(oa[2]), oo[2].astype(np.float32), err_msg="reward, step %d" % t)
        np.testing.assert_array_equal(_cpu(oa[3]), oo[3], err_msg="done, step %d" % t)
        if t == 9:  # a standalone observation between an update and the next step (main.py:88)
            xa, xb = a.observe(), b.observe()
            orc.observe()
            for x, y in zip(xa, xb):
                assert torch_mod.equal(x, y)
    for which in (cm.S_ANTS_XYT, cm.S_PREV_XY, cm.S_HOLDING, cm.S_MANDIBLES, cm.S_PHERO, cm.S_FOOD, cm.S_EXPLORED,
                  cm.S_ANTHILL_FOOD, cm.S_TIMESTEP, cm.S_REWARD_STATE) + ((cm.S_ROCK_CENTERS,) if rocks else ()):
        assert torch_mod.equal(a.read_state(which), b.read_state(which)), which
    assert_xy_close(_cpu(a.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, XY_ATOL)
    np.testing.assert_array_equal(_cpu(a.read_state(cm.S_FOOD)), orc.food)
    np.testing.assert_array_equal(_cpu(a.read_state(cm.S_EXPLORED)), orc.explored)
    np.testing.assert_array_equal(_cpu(a.read_state(cm.S_ANTHILL_FOOD)), orc.anthill_food)
    assert phero_close(_cpu(a.read_state(cm.S_PHERO)), orc.phero, threshold=cfg.phero_threshold).all()


def test_long_run_crosses_the_explored_stamp_rebase(torch_mod):
    """The cell-meta path keeps 'explored before observation s' and 'an ant stands here in observation s' as stamps in
    the cell's META word and re-bases them every 16 381 observations (k_meta_rebase, antsrl_device.h META_NEVER).  A
    single episode longer than that (max_time is the caller's to choose; the reference has no limit) must cross the
    re-basing without a trace: exploration rewards (explored map), the presence channel and the final state against
    the oracle, with every step of the 150 around the crossing compared."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    from oracle.oracle import Oracle
    E, N = 2, 8
    cfg = cm.make_cfg(E, N, 640, 512, n_rocks=2, deposit_strength=256.0, max_time=1 << 30, act_path=cm.ACT_CELL_META)
    init = synth_init(cfg, seed=21, wall_density=0.04)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    orc = Oracle(cfg, init, n_threads=2)
    steps = 16381 + 120
    rng = np.random.default_rng(4)
    nonzero_after = 0
    for t in range(steps):
        rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
        ph = rng.integers(0, 3, (E, N), dtype=np.int8)
        obs, ast, rew, done = env.step_update(rot, ph, None)
        o_obs, o_ast, o_rew, o_done = orc.step(rot, ph)
        orc.update(None)
        if t % 257 == 0 or t > 16381 - 30:
            ctx = "step %d" % t
            np.testing.assert_array_equal(_cpu(rew), o_rew.astype(np.float32), err_msg=ctx)
            go = _cpu(obs)
            for e in range(E):
                check_obs(cfg, go[e], o_obs[e], ctx + " env %d" % e)
            if t > 16381:
                nonzero_after += int((o_rew > 0).sum())
    assert nonzero_after > 0, "the episode stopped exploring before the re-basing: the test no longer tests it"
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_EXPLORED)), orc.explored)
    assert_xy_close(_cpu(env.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, XY_ATOL)  # (16.5 k steps, two rocks: 2.3e-13 seen)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), orc.food)


def test_scaled_units_edge_cases(torch_mod):
    """Scaled pheromone units against the oracle where they need care: an initial grid with
    pheromone on wall cells, ants that START on wall cells (their deposits live for exactly one
    observation), fast decay forcing several re-basings of the units, f0 == 1."""
    from antsrl_amd import config as cm
    from antsrl_amd.synth import synth_init
    for f0, steps, max_val in [(0.999, 12, 255.0), (0.55, 200, 255.0), (1.0, 10, 255.0), (0.9, 30, 100.0)]:
        cfg = cm.make_cfg(3, 48, 40, 40, n_rocks=2, deposit_strength=256.0, filt=np.array([[f0]]),
                          phero_max_val=max_val)
        init = synth_init(cfg, seed=21, wall_density=0.15, n_food_discs=4, food_rmin=2, food_rmax=4)
        rng = np.random.default_rng(4)
        init["phero"] = (rng.random((3, 2, 40, 40)) * 300 * (rng.random((3, 2, 40, 40)) < 0.3)).astype(np.float32)
        if max_val is not None:
            init["phero"] = np.minimum(init["phero"], max_val)
        # put a third of the ants exactly on wall cells
        for e in range(3):
            wx, wy = np.nonzero(init["walls"][e])
            pick = rng.integers(0, len(wx), 16)
            init["ants_xyt"][e, :16, 0] = wx[pick] + 0.5
            init["ants_xyt"][e, :16, 1] = wy[pick] + 0.5
        _compare_with_oracle(torch_mod, cfg, init, steps=steps, seed=9, jitter_mode="builtin")


def test_small_and_odd_shapes(torch_mod):
    """Ragged sizes: non-power-of-two grid whose cell count is not a multiple of 32, ant count
    not a multiple of the wave, 1 ant, no mask, 5x5 perception, a single pheromone channel seen."""
    from antsrl_amd import config as cm
    from antsrl_amd.synth import synth_init
    for (E, N, W, H, kw) in [
        (2, 70, 37, 51, dict(n_rocks=2)),
        (3, 1, 16, 16, dict()),
        (2, 33, 40, 24, dict(mask=None, perception_radius=2)),
        (1, 100, 64, 64, dict(channels=[(cm.CH_FOOD, 0), (cm.CH_PHERO, 1), (cm.CH_ANTHILL, 0)],
                              reward_kind=cm.REWARD_ALL, fct_explore_holding=0.5)),
        (2, 50, 33, 33, dict(n_phero=1, channels=[(cm.CH_ANTS, 0), (cm.CH_PHERO, 0), (cm.CH_WALLS, 0)])),
    ]:
        cfg = cm.make_cfg(E, N, W, H, **kw)
        init = synth_init(cfg, seed=5, n_food_discs=6, food_rmin=2, food_rmax=5)
        if cfg.n_phero != 2:
            # pheromone actions need two channels (ants.py:89-96): drive with rotation only
            from antsrl_amd.batched import BatchedAntsEnv
            from oracle.oracle import Oracle
            env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init)
            env.reset(init)
            rng = np.random.default_rng(0)
            for t in range(8):
                rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
                obs, ast, rew, done = env.step(rot, None)
                o_obs, o_ast, o_rew, o_done = orc.step(rot, None)
                for e in range(E):
                    check_obs(cfg, _cpu(obs)[e], o_obs[e], "1-phero step %d" % t)
                np.testing.assert_array_equal(_cpu(rew), o_rew.astype(np.float32))
                env.update(None)
                orc.update(None)
            continue
        _compare_with_oracle(torch_mod, cfg, init, steps=8, seed=2, jitter_mode="injected")


def test_large_and_ragged_ant_counts(torch_mod):
    """Ant counts on either side of the kernels' one-ant-per-thread fast paths: N > 1024 (k_update's
    per-phase loops, three ants per thread in k_act), an odd N just under 1024 (1024-thread
    k_update_one, odd tail of a wave's run in the pipelined perception loop), and an 11x11 perception
    (two passes of the generic loop)."""
    from antsrl_amd import config as cm
    from antsrl_amd.synth import synth_init
    for (E, N, W, H, kw, steps) in [
        (1, 1500, 64, 64, dict(n_rocks=2, deposit_strength=256.0), 5),
        (1, 2400, 96, 96, dict(n_rocks=2, deposit_strength=256.0), 3),  # the most ants k_act's workgroup LDS holds
        # the cell-meta path's documented limit (include/antsrl.h: 4096 ants per env) and the first size past 2400: k_move's
        # > 64 KiB dynamic-LDS opt-in (a hash table of 8192 slots), several ants per thread in move_body and k_update
        (2, 4096, 96, 96, dict(n_rocks=2, deposit_strength=256.0, act_path=cm.ACT_CELL_META), 3),
        (1, 2500, 64, 80, dict(n_rocks=3, deposit_strength=256.0, act_path=cm.ACT_CELL_META), 3),
        (2, 1023, 96, 80, dict(n_rocks=3, deposit_strength=256.0), 5),
        (2, 300, 64, 64, dict(perception_radius=5, mask=None), 4),
    ]:
        cfg = cm.make_cfg(E, N, W, H, **kw)
        init = synth_init(cfg, seed=9, n_food_discs=5, food_rmin=2, food_rmax=5)
        _compare_with_oracle(torch_mod, cfg, init, steps=steps, seed=4, jitter_mode="injected")


def test_known_answers(torch_mod):
    """Hand-derived cases for the semantics catalogued in SURVEY.md §8(a)."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    W = H = 16
    N = 4
    cfg = cm.make_cfg(1, N, W, H, deposit_strength=256.0, max_time=3)
    walls = np.zeros((1, W, H), np.uint8)
    food = np.zeros((1, W, H), np.float32)
    food[0, 5, 5] = 3.0  # two ants on one food cell: both take min(5,3)=3, cell rewritten once
    xyt = np.array([[[5.2, 5.7, 0.0], [5.9, 5.1, 1.0], [9.5, 9.5, 2.0], [9.5, 9.5, 3.0]]])
    init = dict(ants_xyt=xyt, seed=np.full((1, N), 0.25), walls=walls, food=food,
                anthill_xyr=np.array([[12, 12, 1]], np.int32))
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    obs, ast, rew, done = env.step(np.zeros((1, N), np.int8), np.array([[1, 2, 1, 2]], np.int8))
    assert _cpu(ast)[0, :, 0].tolist() == [3.0, 3.0, 0.0, 0.0]      # both gained (ants.py:111,117)
    assert _cpu(env.read_state(cm.S_FOOD))[0, 5, 5] == 0.0           # decremented once, last writer
    assert _cpu(done)[0] == 0
    env.update(None)
    # deposit: ants 2 and 3 share cell (10,9)/(9,..)? use their floor cells; last ant wins per cell
    xy = np.floor(_cpu(env.read_state(cm.S_ANTS_XYT))[0, :, :2]).astype(int)
    ph = _cpu(env.read_state(cm.S_PHERO))[0]
    act = _cpu(env.read_state(cm.S_ACTIVATION))[0]
    for i in range(N):
        last = max(j for j in range(N) if (xy[j] == xy[i]).all())
        for c in range(2):
            assert ph[c, xy[i][0], xy[i][1]] == min(255.0, act[last, c]), (i, c)
    # done only on the step where timestep == max_time (RL_api.py:200); timestep is now 2
    _, _, _, done = env.step(None, None)
    assert _cpu(done)[0] == 0
    env.update(None)
    _, _, _, done = env.step(None, None)
    assert _cpu(done)[0] == 1
    env.update(None)
    _, _, _, done = env.step(None, None)
    assert _cpu(done)[0] == 0
    # masked perception cells are exactly -1; presence channel is 0/1
    o = _cpu(obs)[0]
    mask = cm.mask_array(cfg)
    assert (o[:, ~mask, :] == -1).all()
    assert set(np.unique(o[:, mask, 0])) <= {0.0, 1.0}


def test_large_grids_and_many_envs(torch_mod):
    """The ends of the size range: grids whose presence / explored bit maps do not fit the workgroup's
    LDS and live in HBM scratch (1024 x 768, 1024 x 1024), a long thin grid, and thousands of tiny envs
    in one handle (grid dimension = env count), with and without the explicit sweep kernel."""
    from antsrl_amd import config as cm
    from antsrl_amd.synth import synth_init
    for (E, N, W, H, kw, steps) in [
        (1, 200, 1024, 768, dict(n_rocks=3, deposit_strength=256.0), 3),
        (2, 300, 1024, 1024, dict(deposit_strength=256.0, reward_kind=cm.REWARD_ALL, fct_explore_holding=0.5), 3),
        (2, 64, 2048, 16, dict(deposit_strength=256.0), 3),
        (1, 96, 640, 640, dict(deposit_strength=256.0, phero_mode=cm.PHERO_EXPLICIT_SWEEP), 2),
        (3000, 3, 16, 16, dict(deposit_strength=256.0), 3),
        (2500, 5, 16, 24, dict(n_rocks=1, deposit_strength=256.0, phero_mode=cm.PHERO_EXPLICIT_SWEEP), 2),
    ]:
        cfg = cm.make_cfg(E, N, W, H, **kw)
        init = synth_init(cfg, seed=17, n_food_discs=5, food_rmin=2, food_rmax=5)
        _compare_with_oracle(torch_mod, cfg, init, steps=steps, seed=6, jitter_mode="builtin")


def test_bfloat16_observation_format(torch_mod):
    """antsrl_set_obs_format(BF16): the observation tensor holds exactly the float32 observation rounded
    to nearest even, for both default layouts (with and without rocks), odd row alignments and an odd
    ant count; unsupported layouts are refused, state and rewards do not depend on the format."""
    from antsrl_amd import _lib, config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    torch = torch_mod
    for (E, N, W, H, R) in [(3, 129, 64, 48, 0), (2, 512, 256, 256, 8), (2, 77, 40, 40, 3)]:
        cfg = cm.make_cfg(E, N, W, H, n_rocks=R, deposit_strength=256.0)
        init = synth_init(cfg, seed=31, n_food_discs=5, food_rmin=2, food_rmax=5)
        a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg, obs_dtype=torch.bfloat16)
        a.reset(init)
        b.reset(init)
        rot, ph = random_actions(cfg, 6, seed=3)
        oa, ob = a.observe()[0], b.observe()[0]
        assert ob.dtype == torch.bfloat16 and torch.equal(oa.to(torch.bfloat16), ob)
        for t in range(6):
            ra = a.step_update(rot[t], ph[t], None)
            rb = b.step_update(rot[t], ph[t], None)
            assert torch.equal(ra[0].to(torch.bfloat16), rb[0]), "step %d" % t
            assert all(torch.equal(x, y) for x, y in zip(ra[1:], rb[1:]))
        assert torch.equal(a.read_state(cm.S_ANTS_XYT), b.read_state(cm.S_ANTS_XYT))
    cfg = cm.make_cfg(1, 20, 32, 32, channels=[(cm.CH_FOOD, 0), (cm.CH_ANTS, 0)])
    env = BatchedAntsEnv(cfg, obs_dtype=torch.bfloat16)
    env.reset(synth_init(cfg, seed=1, n_food_discs=2, food_rmin=2, food_rmax=3))
    with pytest.raises(_lib.AntsrlError, match="bfloat16 observations"):
        env.observe()
