"""An environment's identity survives sharding (AntsCfg.env_id_base, ABI 5).

The reference seeds every environment by itself (generator/environment_generator.py:53-55; the np.random stream
Walls.update draws from, walls.py:28, belongs to that environment), so the trajectory of global environment g must not
depend on which handle, rank or batch position it lands on.  ONE GPU is enough to test that: a handle over the
environments [lo, hi) of a batch, created with env_id_base = lo and stepped with THE LIBRARY'S OWN random streams (wall
jitter, both device generators, auto-reset), must reproduce rows lo:hi of the whole-batch handle bit for bit —
observation, reward, done, agent_state every step and the complete state at the end — for ragged lo / hi, on both kernel
paths and in both pheromone modes.  And the oracle, given the same env_id_base, follows either."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state_keys(cm, rocks):
    keys = [cm.S_ANTS_XYT, cm.S_PREV_XY, cm.S_HOLDING, cm.S_MANDIBLES, cm.S_ACTIVATION, cm.S_PHERO, cm.S_FOOD, cm.S_EXPLORED,
            cm.S_ANTHILL_FOOD, cm.S_TIMESTEP, cm.S_REWARD_STATE, cm.S_WALLS, cm.S_ANTHILL_AREA, cm.S_SEED, cm.S_ANTHILL_XYR]
    if rocks:
        keys += [cm.S_ROCK_CENTERS, cm.S_ROCK_RW]
    return keys


def _shards(E, cuts):
    edges = [0] + list(cuts) + [E]
    return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]


def _assert_rows(torch, whole_outs, lo, hi, part_outs, ctx):
    for name, a, b in zip(("obs", "agent_state", "reward", "done"), whole_outs, part_outs):
        assert torch.equal(a[lo:hi], b), "%s, envs [%d, %d): %s differs from the whole batch" % (ctx, lo, hi, name)


def _assert_states(torch, cm, whole, parts, rocks):
    for which in _state_keys(cm, rocks):
        sw = whole.read_state(which)
        for lo, hi, env in parts:
            assert torch.equal(sw[lo:hi], env.read_state(which)), "state selector %d of envs [%d, %d)" % (which, lo, hi)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.mark.parametrize("act,mode", [(1, 0), (1, 1), (2, 0)], ids=["cell_meta-scaled", "cell_meta-explicit", "single_kernel"])
@pytest.mark.parametrize("E,N,W,H,rocks,cuts", [
    (37, 96, 64, 64, 2, (5, 20)),        # ragged blocks: 5 / 15 / 17 environments
    (9, 200, 48, 40, 0, (1, 8)),         # a one-environment shard at either end
    (16, 64, 64, 64, 3, (8,)),           # two equal halves
])
def test_shard_reproduces_its_rows_of_the_whole_batch(torch_mod, act, mode, E, N, W, H, rocks, cuts):
    """Uploaded initial state + the library's wall jitter (step_update(..., None): k_update_move every step on the
    cell-meta path with scaled units), walls at 8 % so that ants hit one in every environment at nearly every step."""
    torch = torch_mod
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    steps = 12
    kw = dict(n_rocks=rocks, deposit_strength=256.0, act_path=act, phero_mode=mode, rng_seed=0xABCDEF)
    cfg = cm.make_cfg(E, N, W, H, **kw)
    init = synth_init(cfg, seed=31, wall_density=0.08, n_food_discs=5, food_rmin=2, food_rmax=5)
    whole = BatchedAntsEnv(cfg)
    whole.reset(init)
    rot, ph = random_actions(cfg, steps, seed=3)
    parts = []
    for lo, hi in _shards(E, cuts):
        env = BatchedAntsEnv(cm.make_cfg(hi - lo, N, W, H, env_id_base=lo, n_envs_total=E, **kw))
        env.reset({k: np.ascontiguousarray(v[lo:hi]) for k, v in init.items()})
        parts.append((lo, hi, env))
    # ... and a shard that FORGETS its base: the same inputs under local ids draw other jitter (the test has teeth)
    lo_f, hi_f = _shards(E, cuts)[-1]
    forgot = BatchedAntsEnv(cm.make_cfg(hi_f - lo_f, N, W, H, **kw))
    forgot.reset({k: np.ascontiguousarray(v[lo_f:hi_f]) for k, v in init.items()})
    for t in range(steps):
        ow = [x.clone() for x in whole.step_update(rot[t], ph[t], None)]
        for lo, hi, env in parts:
            op = env.step_update(np.ascontiguousarray(rot[t][lo:hi]), np.ascontiguousarray(ph[t][lo:hi]), None)
            _assert_rows(torch, ow, lo, hi, op, "step %d" % t)
        forgot.step_update(np.ascontiguousarray(rot[t][lo_f:hi_f]), np.ascontiguousarray(ph[t][lo_f:hi_f]), None)
    _assert_states(torch, cm, whole, parts, rocks)
    assert not torch.equal(whole.read_state(cm.S_ANTS_XYT)[lo_f:hi_f], forgot.read_state(cm.S_ANTS_XYT)), \
        "no wall was hit (or the jitter ignores the environment id): the test checks nothing"


@pytest.mark.parametrize("rng,walls", [("counter", "bernoulli"), ("counter", "perlin"), ("reference", "perlin"), ("reference", "input")])
def test_device_generator_and_auto_reset_follow_the_global_id(torch_mod, rng, walls):
    """antsrl_generate on a shard draws what the whole batch draws for those environments — both random sources, all wall
    kinds — and keeps doing so across auto-resets (three short episodes): with the reference's streams global env g of
    episode k takes seed episode_seed + k * n_envs_total + g whatever the sharding."""
    torch = torch_mod
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions
    E, N, W, H, rocks, max_time, seed0 = 11, 80, 64, 48, 2, 4, 77
    kw = dict(n_rocks=rocks, deposit_strength=256.0, max_time=max_time)
    cfg = cm.make_cfg(E, N, W, H, **kw)
    steps = 3 * max_time + 2
    rot, ph = random_actions(cfg, steps, seed=12)
    wall_maps = (np.random.default_rng(4).random((E, W, H)) < 0.1).astype(np.uint8) if walls == "input" else None

    def gen():
        return cm.make_gen(wall_density=0.1 if walls != "perlin" else 0.05, n_food_discs=4, food_rmin=2, food_rmax=5, auto_reset=True,
                           walls=walls, rng=rng)
    whole = BatchedAntsEnv(cfg)
    whole.generate(gen(), seed0, walls=wall_maps)
    parts = []
    for lo, hi in _shards(E, (4, 5)):
        env = BatchedAntsEnv(cm.make_cfg(hi - lo, N, W, H, env_id_base=lo, n_envs_total=E, **kw))
        env.generate(gen(), seed0, walls=None if wall_maps is None else wall_maps[lo:hi])
        parts.append((lo, hi, env))
    _assert_states(torch, cm, whole, parts, rocks)
    first = whole.read_state(cm.S_ANTHILL_XYR).clone()
    dones = 0
    for t in range(steps):
        ow = [x.clone() for x in whole.step_update(rot[t], ph[t], None)]
        dones += int(ow[3].sum().item())
        for lo, hi, env in parts:
            op = env.step_update(np.ascontiguousarray(rot[t][lo:hi]), np.ascontiguousarray(ph[t][lo:hi]), None)
            _assert_rows(torch, ow, lo, hi, op, "step %d" % t)
    _assert_states(torch, cm, whole, parts, rocks)
    assert dones == 3 * E, "three episodes should have ended (done fired %d times over %d envs)" % (dones, E)
    assert not torch.equal(first, whole.read_state(cm.S_ANTHILL_XYR)), "the auto-reset drew the first episode again"


def test_reference_streams_on_a_shard_equal_the_host_generator(torch_mod):
    """ANTSRL_RNG_REFERENCE with env_id_base = lo: env e of the shard is EnvironmentGenerator(seed = seed + lo + e) of the
    reference — checked against the host generator (tests/test_generator.py pins that one to the golden fixtures)."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.generator import CirclesGenerator, EnvironmentGenerator
    lo, n, N, W, H, seed = 6, 3, 40, 64, 64, 5

    class NoWalls:
        def generate(self, w, h):
            return np.zeros((w, h), bool)
    host = EnvironmentGenerator(W, H, N, 2, 2, CirclesGenerator(6, 3, 6), NoWalls(), 100, seed=seed, n_envs=n, env_id_base=lo).draw()
    one = [EnvironmentGenerator(W, H, N, 2, 2, CirclesGenerator(6, 3, 6), NoWalls(), 100, seed=seed + lo + e).draw() for e in range(n)]
    for k in host:
        np.testing.assert_array_equal(host[k], np.concatenate([o[k] for o in one]), err_msg=k)
    env = BatchedAntsEnv(cm.make_cfg(n, N, W, H, n_rocks=2, env_id_base=lo, n_envs_total=20))
    env.generate(cm.make_gen(0.0, 6, 3, 6, rng="reference"), seed)
    np.testing.assert_array_equal(env.read_state(cm.S_ANTHILL_XYR).cpu().numpy(), host["anthill_xyr"])
    np.testing.assert_array_equal(env.read_state(cm.S_FOOD).cpu().numpy(), host["food"])
    np.testing.assert_array_equal(env.read_state(cm.S_SEED).cpu().numpy(), host["seed"].astype(np.float32))
    np.testing.assert_allclose(env.read_state(cm.S_ANTS_XYT).cpu().numpy(), host["ants_xyt"], rtol=0, atol=1e-11)
    rk = np.concatenate([env.read_state(cm.S_ROCK_CENTERS).cpu().numpy(), env.read_state(cm.S_ROCK_RW).cpu().numpy()], axis=-1)
    np.testing.assert_array_equal(rk, host["rocks"])


def test_oracle_follows_a_shard_through_env_id_base(torch_mod):
    """The oracle with the same env_id_base draws the same built-in jitter as the device: a shard [lo, hi) of a batch against
    the oracle's restatement of exactly those environments (cell indices exact, rewards bit for bit)."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    from oracle.oracle import Oracle
    lo, n, N, W, H, steps = 1000, 4, 128, 48, 48, 10
    cfg = cm.make_cfg(n, N, W, H, n_rocks=2, deposit_strength=256.0, env_id_base=lo)
    init = synth_init(cfg, seed=8, wall_density=0.15, n_food_discs=3, food_rmin=2, food_rmax=4, env_offset=lo)
    rot, ph = random_actions(cfg, steps, seed=1)
    env, orc = BatchedAntsEnv(cfg), Oracle(cfg, init, n_threads=2)
    env.reset(init)
    cfg0 = cm.make_cfg(n, N, W, H, n_rocks=2, deposit_strength=256.0)  # the same inputs under env ids 0..3
    orc0 = Oracle(cfg0, init, n_threads=2)
    for t in range(steps):
        obs, ast, rew, done = env.step_update(rot[t], ph[t], None)
        _, _, o_rew, _ = orc.step(rot[t], ph[t], want_obs=False)
        orc.update(None)
        orc0.step(rot[t], ph[t], want_obs=False)
        orc0.update(None)
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew.astype(np.float32), err_msg="step %d" % t)
    xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()
    np.testing.assert_allclose(xyt, orc.ants_xyt, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(np.floor(xyt[..., :2]), np.floor(orc.ants_xyt[..., :2]))
    np.testing.assert_array_equal(env.read_state(cm.S_EXPLORED).cpu().numpy(), orc.explored)
    assert np.abs(orc.ants_xyt[..., 2] - orc0.ants_xyt[..., 2]).max() > 1e-3, "no wall hit: the base was not exercised"
