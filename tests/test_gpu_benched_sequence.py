"""What bench.py times, at the size it is timed (VERDICT r3 "Next round" #2).

bench.py's loop is `antsrl_step_update(rot, ph, NULL jitter)` with NO state read between steps: on the cell-meta path with
scaled pheromone units that is k_update_move (the deferred update of step t fused with the move of step t + 1, the
library's own wall jitter) + k_perceive, every step, on tiled interleaved cell records, from an AGED episode.  Earlier
full-batch parity tests injected the jitter (which disables the deferral) — the exact launch sequence met the oracle at
E <= 32 only.  With AntsCfg.env_id_base the oracle can follow any environment of a full batch through the library's own
jitter, so here the benched sequence itself is compared:

  * the full batch of BASELINE configs[2] (1024 envs x 512 ants, 256x256, rocks + walls + food), of configs[1] (256 x 256)
    and of configs[4]'s per-GPU shard (512 x 512, the in-loop bf16 policy driving the ants, observation tensor and act-only);
  * bench.py's inputs (synth_init(seed=1234, env_offset=rank * E), a ring of 8 device-generated random action sets),
    a non-zero env_id_base (what rank 3 of 8 would run);
  * AGE + 60 steps with nothing but the step calls on the stream — the outputs of five sampled environments are copied
    aside by device-side ops, never by a state read, so no pending update is ever flushed;
  * afterwards the oracle replays the five environments (env_id_base = their global ids): reward EVERY step, observation
    and agent_state every 10th step, the complete state at the end.

Reference: main.py:98,131 (the loop), RL_api.py:168-204, environment.py:42-47."""
import numpy as np
import pytest

from helpers import assert_xy_close, phero_close
from test_gpu_parity import XY_ATOL, check_obs

pytestmark = pytest.mark.gpu

RING = 8


def _oracles(cm, Oracle, cfg_kw, N, W, H, init, pick, base):
    out = []
    for g in pick:
        c = cm.make_cfg(1, N, W, H, env_id_base=base + g, **cfg_kw)
        out.append(Oracle(c, {k: np.ascontiguousarray(v[g:g + 1]) for k, v in init.items()}))
    return out


# Coordinates over long horizons WITH circle obstacles.  The bar is EXACT cell indices (north_star); float64 coordinates are
# held to 1e-9 over the suite's other horizons (largest seen without rocks: 8.5e-12, round 5's ANTSRL_ERR_LOG run), but a
# circle obstacle projects an ant that walks into it radially onto its rim (circle_obstacles.py:53-58), which multiplies the
# TANGENTIAL part of any difference by radius / distance (~1.14 for an ant one step inside a radius-8 rock) at every push:
# the 1-ulp difference between the device's sincos and glibc's grows geometrically for an ant that keeps pushing against a
# rock (the reference on another libm would differ from itself the same way; theta, which no rock touches, differs by
# exactly 0).  MEASURED on the full c3 batch (test_full_reference_episode_2000_steps_c3, gpurun_out/parity_errors_c3_2000.json):
# 3.8e-13 after 100 steps, 3.6e-10 after 400, 3.1e-9 after 460, 5.1e-8 after 1000, 7.6e-7 after 1600-2000 — cells, holding,
# food, explored map and every reward exact throughout.  Hence 2e-8 for the 460-step runs and 1e-6 for the 2000-step episode.
XY_ATOL_460 = 2e-8
XY_ATOL_LONG = 1e-6


def _final_state_checks(cm, env, orcs, pick, rocks):
    xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()
    hold = env.read_state(cm.S_HOLDING).cpu().numpy()
    mand = env.read_state(cm.S_MANDIBLES).cpu().numpy()
    af = env.read_state(cm.S_ANTHILL_FOOD).cpu().numpy()
    ts = env.read_state(cm.S_TIMESTEP).cpu().numpy()
    food, expl, ph = env.read_state(cm.S_FOOD), env.read_state(cm.S_EXPLORED), env.read_state(cm.S_PHERO)
    rc = env.read_state(cm.S_ROCK_CENTERS).cpu().numpy() if rocks else None
    for j, g in enumerate(pick):
        o = orcs[j]
        ctx = "final state, env %d" % g
        assert_xy_close(xyt[g], o.ants_xyt[0], XY_ATOL_460 if rocks else XY_ATOL, ctx)
        np.testing.assert_array_equal(np.floor(xyt[g][:, :2]), np.floor(o.ants_xyt[0][:, :2]), err_msg=ctx + " cells")
        np.testing.assert_array_equal(hold[g], o.holding[0], err_msg=ctx + " holding")
        np.testing.assert_array_equal(mand[g], o.mandibles[0], err_msg=ctx + " mandibles")
        assert af[g] == o.anthill_food[0] and ts[g] == o.timestep[0], ctx
        np.testing.assert_array_equal(food[g].cpu().numpy(), o.food[0], err_msg=ctx + " food")
        np.testing.assert_array_equal(expl[g].cpu().numpy(), o.explored[0], err_msg=ctx + " explored")
        ok = phero_close(ph[g].cpu().numpy(), o.phero[0])
        assert ok.all(), "%s pheromone: %d cells off" % (ctx, (~ok).sum())
        if rocks:
            assert_xy_close(rc[g], o.rock_centers[0], XY_ATOL_460, ctx)


@pytest.mark.parametrize("name,E,N,rocks,age", [("c3", 1024, 512, 8, 400), ("c2", 256, 256, 0, 400)])
def test_benched_random_policy_loop_vs_oracle(name, E, N, rocks, age):
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    W = H = 256
    rank, world = 3, 8                       # what rank 3 of the 8-GPU run steps
    base = rank * E
    kw = dict(n_rocks=rocks, deposit_strength=256.0, max_time=1 << 30)
    cfg = cm.make_cfg(E, N, W, H, env_id_base=base, n_envs_total=world * E, **kw)
    init = synth_init(cfg, seed=1234, env_offset=base)  # bench.py's inputs for this rank
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    if not (env.query(cm.Q_INTERLEAVED) == 1 and env.query(cm.Q_DEFERRED_UPDATE) == 1):
        pytest.skip("a profiling switch took the handle off the benched path (tests/alt_paths.sh: ANTSRL_NO_INTERLEAVE / ANTSRL_NO_DEFER_UPDATE)")
    assert env.query(cm.Q_CELL_META) == 1 and env.query(cm.Q_SCALED_UNITS) == 1, "the benched sequence is k_update_move + k_perceive"
    dev = env.device
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    pick = [0, 1, E // 2 - 1, E - 2, E - 1]
    pidx = torch.tensor(pick, device=dev)
    steps = age + 60
    rew_log = torch.empty((steps, len(pick), N), dtype=torch.float32, device=dev)
    done_log = torch.empty((steps, len(pick)), dtype=torch.uint8, device=dev)
    obs_log, ast_log = {}, {}
    for t in range(steps):  # nothing but the step calls and device-side copies of OUTPUT rows: no read of the state
        obs, ast, rew, done = env.step_update(rot[t % RING], ph[t % RING], None)
        rew_log[t] = rew[pidx]
        done_log[t] = done[pidx]
        if t >= age and (t - age) % 10 == 9:
            obs_log[t] = obs[pidx].clone()
            ast_log[t] = ast[pidx].clone()
    torch.cuda.synchronize(dev)
    rot_h, ph_h = rot[:, pidx].cpu().numpy(), ph[:, pidx].cpu().numpy()
    rew_h, done_h = rew_log.cpu().numpy(), done_log.cpu().numpy()
    orcs = _oracles(cm, Oracle, kw, N, W, H, init, pick, base)
    cfg1 = cm.make_cfg(1, N, W, H, **kw)
    for t in range(steps):
        for j, o in enumerate(orcs):
            want_obs = t in obs_log
            o_obs, o_ast, o_rew, o_done = o.step(rot_h[t % RING, j:j + 1], ph_h[t % RING, j:j + 1], want_obs=want_obs)
            o.update(None)  # the library's jitter, keyed on (rng_seed, base + g, timestep, ant)
            np.testing.assert_array_equal(rew_h[t, j], o_rew[0].astype(np.float32), err_msg="%s step %d env %d reward" % (name, t, pick[j]))
            assert done_h[t, j] == o_done[0]
            if want_obs:
                check_obs(cfg1, obs_log[t][j].cpu().numpy(), o_obs[0], "%s step %d env %d" % (name, t, pick[j]))
                np.testing.assert_array_equal(ast_log[t][j].cpu().numpy(), o_ast[0].astype(np.float32))
    _final_state_checks(cm, env, orcs, pick, rocks)
    # the run did what an aged episode does: walls were hit (the jitter mattered), food was carried
    assert sum(float((o.holding > 0).sum()) for o in orcs) > 0


def test_benched_sequence_with_the_reference_drivers_reward():
    """main.py:42 does not run ExplorationReward but All_Rewards(fct_explore=1, fct_food=2, fct_anthill=10,
    fct_explore_holding=1, fct_headinganthill=3), with reward_threshold 1 (main.py:44): the full c3 batch through the benched
    call sequence with THAT reward (k_perceive's epilogue then reads previous holding / previous distance and writes them
    back, and the tint is set from the weighted sum), 300 steps, five sampled environments against the oracle — reward
    bit-exact at every step (on these inputs no ant sits on a mathematical tie of the heading term: test_gpu_parity's
    _check_reward documents that one exception), reward_state (the viewer's tint, ants.py:119-130) and the rest of the state
    exact at the end."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    E, N, W, H, rocks, steps = 1024, 512, 256, 256, 8, 300
    rank, world = 2, 8
    base = rank * E
    kw = dict(n_rocks=rocks, deposit_strength=256.0, max_time=1 << 30, reward_kind=cm.REWARD_ALL, reward_threshold=1.0,
              fct_explore=1.0, fct_food=2.0, fct_anthill=10.0, fct_explore_holding=1.0, fct_headinganthill=3.0)
    cfg = cm.make_cfg(E, N, W, H, env_id_base=base, n_envs_total=world * E, **kw)
    init = synth_init(cfg, seed=1234, env_offset=base)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    if not (env.query(cm.Q_INTERLEAVED) == 1 and env.query(cm.Q_DEFERRED_UPDATE) == 1):
        pytest.skip("a profiling switch took the handle off the benched path (tests/alt_paths.sh)")
    dev = env.device
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    pick = [0, 1, E // 2 - 1, E - 2, E - 1]
    pidx = torch.tensor(pick, device=dev)
    rew_log = torch.empty((steps, len(pick), N), dtype=torch.float32, device=dev)
    for t in range(steps):
        obs, ast, rew, done = env.step_update(rot[t % RING], ph[t % RING], None)
        rew_log[t] = rew[pidx]
    torch.cuda.synchronize(dev)
    rot_h, ph_h, rew_h = rot[:, pidx].cpu().numpy(), ph[:, pidx].cpu().numpy(), rew_log.cpu().numpy()
    orcs = _oracles(cm, Oracle, kw, N, W, H, init, pick, base)
    seen = set()
    for t in range(steps):
        for j, o in enumerate(orcs):
            _, _, o_rew, _ = o.step(rot_h[t % RING, j:j + 1], ph_h[t % RING, j:j + 1], want_obs=False)
            o.update(None)
            np.testing.assert_array_equal(rew_h[t, j], o_rew[0].astype(np.float32), err_msg="All_Rewards, step %d env %d" % (t, pick[j]))
            seen.update(np.unique(np.round(o_rew[0], 6)).tolist())
    # every term of the sum has fired somewhere: exploration (multiples of 0.1), food (2 per unit picked up), anthill (10)
    assert any(abs(v - 10.0) < 1e-6 or v > 10.0 for v in seen) and any(1.9 < v < 10.0 for v in seen) and any(0 < v < 1.9 for v in seen), sorted(seen)[-8:]
    _final_state_checks(cm, env, orcs, pick, rocks)
    rs = env.read_state(cm.S_REWARD_STATE).cpu().numpy()
    for j, g_ in enumerate(pick):
        np.testing.assert_array_equal(rs[g_], orcs[j].reward_state[0], err_msg="reward_state (tint), env %d" % g_)


def test_full_reference_episode_2000_steps_c3():
    """The reference's episode is 2000 steps (main.py:31): the full BASELINE configs[2] batch run that long through the
    benched call sequence, five sampled environments replayed by the oracle — reward bit-exact at EVERY step, and every 100
    steps a checkpoint that reads the state (the pending update is flushed there: bit-identical to the deferred form,
    test_deferred_update_is_bit_identical): ant CELLS, holding, mandibles, food, explored map exact; the float64 coordinate
    and float32 pheromone errors the kernels actually make are written to gpurun_out/parity_errors_c3_2000.json with the
    step at which the largest coordinate difference appeared."""
    import json
    import os
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    E, N, W, H, rocks, steps, every = 1024, 512, 256, 256, 8, 2000, 100
    rank, world = 3, 8
    base = rank * E
    kw = dict(n_rocks=rocks, deposit_strength=256.0, max_time=1 << 30)
    cfg = cm.make_cfg(E, N, W, H, env_id_base=base, n_envs_total=world * E, **kw)
    init = synth_init(cfg, seed=1234, env_offset=base)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    dev = env.device
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, N), generator=g, device=dev, dtype=torch.int8)
    pick = [0, 1, E // 2 - 1, E - 2, E - 1]
    pidx = torch.tensor(pick, device=dev)
    rew_log = torch.empty((steps, len(pick), N), dtype=torch.float32, device=dev)
    snaps = {}
    for t in range(steps):
        obs, ast, rew, done = env.step_update(rot[t % RING], ph[t % RING], None)
        rew_log[t] = rew[pidx]
        if (t + 1) % every == 0:  # a checkpoint: state of the sampled environments after step t's update
            snaps[t] = dict(xyt=env.read_state(cm.S_ANTS_XYT)[pidx].cpu().numpy(), hold=env.read_state(cm.S_HOLDING)[pidx].cpu().numpy(),
                            mand=env.read_state(cm.S_MANDIBLES)[pidx].cpu().numpy(), food=env.read_state(cm.S_FOOD)[pidx].cpu().numpy(),
                            phero=env.read_state(cm.S_PHERO)[pidx].cpu().numpy(), rc=env.read_state(cm.S_ROCK_CENTERS)[pidx].cpu().numpy(),
                            expl=env.read_state(cm.S_EXPLORED)[pidx].cpu().numpy(), af=env.read_state(cm.S_ANTHILL_FOOD)[pidx].cpu().numpy())
    torch.cuda.synchronize(dev)
    rot_h, ph_h, rew_h = rot[:, pidx].cpu().numpy(), ph[:, pidx].cpu().numpy(), rew_log.cpu().numpy()
    orcs = _oracles(cm, Oracle, kw, N, W, H, init, pick, base)
    rec = dict(config="c3 full batch, 5 sampled envs", steps=steps, checkpoints=[])
    worst_xy, worst_xy_step = 0.0, None
    for t in range(steps):
        for j, o in enumerate(orcs):
            _, _, o_rew, _ = o.step(rot_h[t % RING, j:j + 1], ph_h[t % RING, j:j + 1], want_obs=False)
            o.update(None)
            np.testing.assert_array_equal(rew_h[t, j], o_rew[0].astype(np.float32), err_msg="step %d env %d reward" % (t, pick[j]))
        if t in snaps:
            sn = snaps[t]
            cp = dict(step=t + 1, xy_max_abs=0.0, theta_max_abs=0.0, rock_max_abs=0.0, phero_max_abs=0.0, phero_max_rel=0.0)
            for j, o in enumerate(orcs):
                ctx = "checkpoint after step %d, env %d" % (t, pick[j])
                np.testing.assert_array_equal(np.floor(sn["xyt"][j][:, :2]), np.floor(o.ants_xyt[0][:, :2]), err_msg=ctx + " cells")
                np.testing.assert_array_equal(sn["hold"][j], o.holding[0], err_msg=ctx + " holding")
                np.testing.assert_array_equal(sn["mand"][j], o.mandibles[0], err_msg=ctx + " mandibles")
                np.testing.assert_array_equal(sn["food"][j], o.food[0], err_msg=ctx + " food")
                np.testing.assert_array_equal(sn["expl"][j], o.explored[0], err_msg=ctx + " explored")
                assert sn["af"][j] == o.anthill_food[0], ctx
                assert_xy_close(sn["xyt"][j], o.ants_xyt[0], XY_ATOL_LONG, ctx)
                assert_xy_close(sn["rc"][j], o.rock_centers[0], XY_ATOL_LONG, ctx)
                ok = phero_close(sn["phero"][j], o.phero[0])
                assert ok.all(), "%s pheromone: %d cells off" % (ctx, (~ok).sum())
                d = np.abs(sn["xyt"][j] - o.ants_xyt[0])
                cp["xy_max_abs"] = max(cp["xy_max_abs"], float(d[:, :2].max()))
                cp["theta_max_abs"] = max(cp["theta_max_abs"], float(d[:, 2].max()))
                cp["rock_max_abs"] = max(cp["rock_max_abs"], float(np.abs(sn["rc"][j] - o.rock_centers[0]).max()))
                want = o.phero[0].astype(np.float64)
                dp = np.abs(sn["phero"][j].astype(np.float64) - want)
                far = np.abs(want - 0.01) > 4e-7  # (outside the band at the cut)
                cp["phero_max_abs"] = max(cp["phero_max_abs"], float(dp[far].max()))
                nz = far & (want != 0)
                cp["phero_max_rel"] = max(cp["phero_max_rel"], float((dp[nz] / want[nz]).max()) if nz.any() else 0.0)
            if cp["xy_max_abs"] > worst_xy:
                worst_xy, worst_xy_step = cp["xy_max_abs"], t + 1
            rec["checkpoints"].append(cp)
    rec.update(xy_max_abs=worst_xy, xy_max_abs_first_seen_at_step=worst_xy_step,
               phero_max_rel=max(c["phero_max_rel"] for c in rec["checkpoints"]),
               phero_max_abs=max(c["phero_max_abs"] for c in rec["checkpoints"]),
               cells_exact_at_every_checkpoint=True, reward_exact_at_every_step=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(rec, open(os.path.join(out, "parity_errors_c3_2000.json"), "w"), indent=1)
    print("c3, 2000 steps: xy max |d| %.3g (first at step %s), pheromone max rel %.3g, max abs %.3g"
          % (worst_xy, worst_xy_step, rec["phero_max_rel"], rec["phero_max_abs"]))


@pytest.mark.parametrize("want_obs", [True, False], ids=["obs_tensor", "act_only"])
def test_benched_inloop_policy_loop_vs_oracle(want_obs):
    """configs[4]'s per-GPU shard as bench.py --config c5 [--no-obs] runs it: the DQN net inside k_perceive picks the next
    step's actions (env.next_rotation / next_pheromone), antsrl_step_update consumes them, library jitter, no read between
    steps.  The oracle follows five environments with the actions the device chose (copied aside each step)."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.policy import LinearPolicy
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    E, N, W, H, age = 512, 512, 256, 256, 200
    rank, world = 5, 8
    base = rank * E
    kw = dict(n_rocks=0, deposit_strength=256.0, max_time=1 << 30)
    cfg = cm.make_cfg(E, N, W, H, env_id_base=base, n_envs_total=world * E, **kw)
    init = synth_init(cfg, seed=1234, env_offset=base)
    env = BatchedAntsEnv(cfg, obs_dtype=torch.bfloat16)
    env.reset(init)
    if env.query(cm.Q_DEFERRED_UPDATE) != 1:
        pytest.skip("a profiling switch took the handle off the benched path (tests/alt_paths.sh: ANTSRL_NO_DEFER_UPDATE)")
    dev = env.device
    pol = LinearPolicy(cfg.pside * cfg.pside * cfg.n_channels, dev, seed=5 + rank)
    if env.query(cm.Q_PERCEIVE_RUN) * 4 > 32:
        pytest.skip("the in-loop policy needs a 32-ant tile per workgroup (tests/alt_paths.sh: ANTSRL_PRC_RUN)")
    pol.attach(env)
    env.observe(want_obs=want_obs)  # main.py:88
    pick = [0, 1, 255, 510, 511]
    pidx = torch.tensor(pick, device=dev)
    steps = age + 60
    rot_log = torch.empty((steps, len(pick), N), dtype=torch.int8, device=dev)
    ph_log = torch.empty((steps, len(pick), N), dtype=torch.int8, device=dev)
    rew_log = torch.empty((steps, len(pick), N), dtype=torch.float32, device=dev)
    obs_log, ast_log = {}, {}
    for t in range(steps):
        rot_log[t] = env.next_rotation[pidx]   # the actions this step consumes (outputs of the previous observation)
        ph_log[t] = env.next_pheromone[pidx]
        obs, ast, rew, done = env.step_update(env.next_rotation, env.next_pheromone, None, want_obs=want_obs)
        rew_log[t] = rew[pidx]
        if t >= age and (t - age) % 10 == 9:
            ast_log[t] = ast[pidx].clone()
            if want_obs:
                obs_log[t] = obs[pidx].to(torch.float32)
    torch.cuda.synchronize(dev)
    rot_h, ph_h, rew_h = rot_log.cpu().numpy(), ph_log.cpu().numpy(), rew_log.cpu().numpy()
    assert set(np.unique(rot_h)) <= {-1, 0, 1} and set(np.unique(ph_h)) <= {0, 1, 2}
    orcs = _oracles(cm, Oracle, kw, N, W, H, init, pick, base)
    for o in orcs:
        o.observe(want_obs=False)  # main.py:88: the first observation marks the explored map
    for t in range(steps):
        for j, o in enumerate(orcs):
            w = t in obs_log
            o_obs, o_ast, o_rew, _ = o.step(rot_h[t, j:j + 1], ph_h[t, j:j + 1], want_obs=w)
            o.update(None)
            np.testing.assert_array_equal(rew_h[t, j], o_rew[0].astype(np.float32), err_msg="c5 step %d env %d reward" % (t, pick[j]))
            if t in ast_log:
                np.testing.assert_array_equal(ast_log[t][j].cpu().numpy(), o_ast[0].astype(np.float32))
            if w:  # the bf16 tensor: integer channels exact, pheromone channels one bf16 ulp around a 1e-5 difference
                got = obs_log[t][j].cpu().numpy()
                want = torch.from_numpy(o_obs[0].astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
                np.testing.assert_array_equal(got[..., [0, 3, 4, 5]], want[..., [0, 3, 4, 5]])
                assert np.abs(got[..., 1:3] - want[..., 1:3]).max() <= 2 ** -8
    _final_state_checks(cm, env, orcs, pick, 0)


def test_benched_config4_loop_vs_oracle():
    """configs[3]'s per-GPU shard as bench.py --config c4 runs it (1024 envs x 1024 ants, 512 x 512, radius-3 diffusion):
    k_sweep_sep2 + k_move + k_perceive + k_update_one every step — explicit-sweep records ({food, META} in 4 x 4-cell blocks,
    two gathers per cell), the library's wall jitter, no read between steps.  Three environments follow the oracle (a 7 x 7
    float64 convolution over 2 x 262 144 cells per step: a short horizon)."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    from oracle.oracle import Oracle
    E, N, W, H, steps = 1024, 1024, 512, 512, 24
    rank, world = 6, 8
    base = rank * E
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    kw = dict(n_rocks=0, deposit_strength=256.0, max_time=1 << 30, filt=g / g.sum() * (1 - 0.001))  # bench.py's filter
    cfg = cm.make_cfg(E, N, W, H, env_id_base=base, n_envs_total=world * E, **kw)
    init = synth_init(cfg, seed=1234, env_offset=base)
    env = BatchedAntsEnv(cfg)
    env.reset(init)
    assert env.query(cm.Q_CELL_META) == 1 and env.query(cm.Q_SCALED_UNITS) == 0
    if env.query(cm.Q_FILTER_SEPARABLE) != 1:
        pytest.skip("a profiling switch took the separable stencil away (tests/alt_paths.sh: ANTSRL_NO_SEPARABLE)")
    dev = env.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, N), generator=gen, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, N), generator=gen, device=dev, dtype=torch.int8)
    pick = [0, 517, E - 1]
    pidx = torch.tensor(pick, device=dev)
    rew_log = torch.empty((steps, len(pick), N), dtype=torch.float32, device=dev)
    obs_log = {}
    for t in range(steps):
        obs, ast, rew, done = env.step_update(rot[t % RING], ph[t % RING], None)
        rew_log[t] = rew[pidx]
        if t % 8 == 7:
            obs_log[t] = obs[pidx].clone()
    torch.cuda.synchronize(dev)
    rot_h, ph_h, rew_h = rot[:, pidx].cpu().numpy(), ph[:, pidx].cpu().numpy(), rew_log.cpu().numpy()
    sub = {k: np.ascontiguousarray(v[pick]) for k, v in init.items()}
    del init
    orcs = _oracles(cm, Oracle, kw, N, W, H, {k: v for k, v in sub.items()}, list(range(len(pick))), 0)
    for j, o in enumerate(orcs):  # (the oracles were built on the sub-batch's rows: give each its GLOBAL env id)
        o.cfg.env_id_base = base + pick[j]
    cfg1 = cm.make_cfg(1, N, W, H, **kw)
    for t in range(steps):
        for j, o in enumerate(orcs):
            w = t in obs_log
            o_obs, _, o_rew, _ = o.step(rot_h[t % RING, j:j + 1], ph_h[t % RING, j:j + 1], want_obs=w)
            o.update(None)
            np.testing.assert_array_equal(rew_h[t, j], o_rew[0].astype(np.float32), err_msg="c4 step %d env %d reward" % (t, pick[j]))
            if w:
                check_obs(cfg1, obs_log[t][j].cpu().numpy(), o_obs[0], "c4 step %d env %d" % (t, pick[j]))
    xyt = env.read_state(cm.S_ANTS_XYT)
    phero, food, expl = env.read_state(cm.S_PHERO), env.read_state(cm.S_FOOD), env.read_state(cm.S_EXPLORED)
    for j, g_ in enumerate(pick):
        o = orcs[j]
        assert_xy_close(xyt[g_].cpu().numpy(), o.ants_xyt[0], XY_ATOL)
        np.testing.assert_array_equal(food[g_].cpu().numpy(), o.food[0])
        np.testing.assert_array_equal(expl[g_].cpu().numpy(), o.explored[0])
        ok = phero_close(phero[g_].cpu().numpy(), o.phero[0])
        assert ok.all(), "c4 env %d pheromone: %d cells off" % (g_, (~ok).sum())
