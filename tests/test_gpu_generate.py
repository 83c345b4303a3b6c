"""Device-side episode generation and auto-reset against the oracle's
restatement of the same generator (oracle_generate_init)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cpu(t):
    return t.detach().cpu().numpy()


def _check_fresh_episode(env, cfg, gen, seed):
    from antsrl_amd import config as cm
    from oracle.oracle import Oracle, generate_init
    init = generate_init(cfg, gen, seed)
    orc = Oracle(cfg, init)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_WALLS)), init["walls"])
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), init["food"])
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ANTHILL_AREA)), orc.anthill_area)
    np.testing.assert_allclose(_cpu(env.read_state(cm.S_ANTS_XYT)), orc.ants_xyt, rtol=0, atol=1e-11)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_SEED)), init["seed"].astype(np.float32))
    if cfg.n_rocks:
        np.testing.assert_array_equal(_cpu(env.read_state(cm.S_ROCK_CENTERS)), init["rocks"][..., :2])
    assert (_cpu(env.read_state(cm.S_TIMESTEP)) == 1).all()
    assert _cpu(env.read_state(cm.S_PHERO)).max() == 0 and _cpu(env.read_state(cm.S_EXPLORED)).max() == 0
    assert _cpu(env.read_state(cm.S_HOLDING)).max() == 0
    # sanity of what was drawn (environment_generator.py:60-72): anthill in the central half, free of walls
    xyr = init["anthill_xyr"]
    assert (xyr[:, 0] >= cfg.w // 4).all() and (xyr[:, 0] < 3 * cfg.w // 4 + 1).all()
    assert (init["walls"][orc.anthill_area.astype(bool)] == 0).all()
    assert (init["food"][init["walls"].astype(bool)] == 0).all()
    return orc


def test_generate_matches_oracle_and_steps_in_parity():
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions
    from test_gpu_parity import check_obs
    cfg = cm.make_cfg(5, 96, 128, 96, n_rocks=3, deposit_strength=256.0)
    gen = cm.make_gen(wall_density=0.07, n_food_discs=12, food_rmin=3, food_rmax=8)
    env = BatchedAntsEnv(cfg)
    env.generate(gen, episode_seed=42)
    orc = _check_fresh_episode(env, cfg, gen, 42)
    rot, ph = random_actions(cfg, 10, seed=1)
    for t in range(10):
        obs, ast, rew, done = env.step_update(rot[t], ph[t])
        o_obs, o_ast, o_rew, o_done = orc.step(rot[t], ph[t])
        orc.update(None)
        for e in range(cfg.n_envs):
            check_obs(cfg, _cpu(obs)[e], o_obs[e], "generated env %d step %d" % (e, t))
        np.testing.assert_array_equal(_cpu(rew), o_rew.astype(np.float32))
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_FOOD)), orc.food)
    # a different seed gives a different world; the same seed the same one
    env.generate(gen, episode_seed=43)
    assert not np.array_equal(_cpu(env.read_state(cm.S_WALLS)), orc.walls)
    env.generate(gen, episode_seed=42)
    np.testing.assert_array_equal(_cpu(env.read_state(cm.S_WALLS)), orc.walls)


def test_auto_reset_on_done():
    """max_time = 4: the step with timestep == 4 reports done (RL_api.py:200); right after its update
    every env is regenerated with episode_seed + 1, and again one episode later."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    cfg = cm.make_cfg(3, 40, 64, 64, max_time=4, deposit_strength=256.0)
    gen = cm.make_gen(wall_density=0.05, n_food_discs=6, food_rmin=3, food_rmax=6, auto_reset=True)
    env = BatchedAntsEnv(cfg)
    env.generate(gen, episode_seed=7)
    dones = []
    rot = np.zeros((3, 40), np.int8)
    for t in range(9):
        obs, ast, rew, done = env.step_update(rot, rot)
        dones.append(int(_cpu(done)[0]))
        if t == 3:
            _check_fresh_episode(env, cfg, gen, 8)
        if t == 7:
            _check_fresh_episode(env, cfg, gen, 9)
    # timestep runs 1,2,3,4(done) | 1,2,3,4(done) | 1
    assert dones == [0, 0, 0, 1, 0, 0, 0, 1, 0]


def test_device_generator_behind_the_reference_surface():
    """DeviceEnvironmentGenerator: same RLApi / Environment wiring as the host generator, state
    drawn on the GPU, auto-reset at max_steps."""
    from antsrl_amd.generator import DeviceEnvironmentGenerator
    from antsrl_amd.rl_api import ExplorationReward, Pheromone, RLApi
    api = RLApi(ExplorationReward(), 1, 1, 40 / 180 * np.pi, 0.05, 0.5)
    env = DeviceEnvironmentGenerator(64, 64, 24, 2, 2, max_steps=3, seed=5, n_envs=1, n_food_discs=5,
                                     food_rmin=2, food_rmax=5, auto_reset=True).generate(api)
    assert api.ants.n_ants == 24 and len(api.perceived_objects) == 7
    assert sum(isinstance(o, Pheromone) for o in api.perceived_objects) == 2
    anthill = [o for o in env.objects if type(o).__name__ == "_DeviceAnthill"][0]
    assert 16 <= anthill.x <= 48 and 16 <= anthill.y <= 48 and 3 <= anthill.radius <= 6
    obs, ast, state = api.observation()
    assert obs.shape == (24, 7, 7, 7)
    flags = []
    for t in range(4):
        o, a, r, d = api.step(np.zeros(24, np.int8), None)
        env.update()
        flags.append((d, env.timestep))
    # separate step()/update() calls do not auto-reset (only the fused antsrl_step_update does)
    assert [f[0] for f in flags] == [False, False, True, False] and flags[-1][1] == 5


@pytest.mark.parametrize("seed", range(16))
def test_generate_random_shapes_and_parameters(seed):
    """The device generator against the oracle's restatement over random grid shapes, ant counts,
    rock counts and generator parameters (including no food discs, dense walls, tiny grids)."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    rng = np.random.default_rng(500 + seed)
    E, N = int(rng.integers(1, 5)), int(rng.choice([1, 9, 64, 130, 300]))
    W, H = int(rng.integers(16, 150)), int(rng.integers(16, 150))
    cfg = cm.make_cfg(E, N, W, H, n_rocks=int(rng.integers(0, 5)), deposit_strength=256.0)
    rmin = int(rng.integers(1, 6))
    gen = cm.make_gen(wall_density=float(rng.choice([0.0, 0.05, 0.3])), n_food_discs=int(rng.choice([0, 1, 7, 20])),
                      food_rmin=rmin, food_rmax=rmin + int(rng.integers(0, 6)))
    env = BatchedAntsEnv(cfg)
    env.generate(gen, episode_seed=1000 + seed)
    orc = _check_fresh_episode(env, cfg, gen, 1000 + seed)
    rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
    ph = rng.integers(0, 3, (E, N), dtype=np.int8)
    obs, ast, rew, done = env.step_update(rot, ph)
    o_obs, o_ast, o_rew, o_done = orc.step(rot, ph)
    np.testing.assert_array_equal(_cpu(rew), o_rew.astype(np.float32))
    np.testing.assert_array_equal(_cpu(obs)[..., [0, 3, 4, 5]], o_obs[..., [0, 3, 4, 5]])


@pytest.mark.parametrize("case", range(6))
def test_generate_perlin_walls(case):
    """walls="perlin" (PerlinGenerator, main.py:75) on the device: every env's wall map equals the host
    generator's noise field at that env's offsets, thresholded and cleared on the anthill area — bit for
    bit (device float32 == numpy float32 == the oracle's C), for the reference's parameters and others."""
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.generator import perlin_noise
    from oracle import oracle
    rng = np.random.default_rng(900 + case)
    E, N = int(rng.integers(1, 6)), int(rng.choice([8, 50]))
    W, H = int(rng.integers(24, 200)), int(rng.integers(24, 200))
    kw = [dict(), dict(perlin_scale=22.0, wall_density=0.3), dict(perlin_scale=9.5, perlin_octaves=3, wall_density=0.1),
          dict(perlin_octaves=1, wall_density=0.0), dict(perlin_persistence=0.7, perlin_lacunarity=1.7, perlin_octaves=4,
                                                         wall_density=-0.1), dict(wall_density=0.05)][case]
    gen = cm.make_gen(walls="perlin", n_food_discs=6, food_rmin=2, food_rmax=5, **kw)
    cfg = cm.make_cfg(E, N, W, H, deposit_strength=256.0)
    env = BatchedAntsEnv(cfg)
    seed = 77 + case
    env.generate(gen, episode_seed=seed)
    orc = _check_fresh_episode(env, cfg, gen, seed)          # device == oracle (walls, food, ants, ...)
    walls = _cpu(env.read_state(cm.S_WALLS)).astype(bool)
    area = _cpu(env.read_state(cm.S_ANTHILL_AREA)).astype(bool)
    SALT = 0x6A09E667F3BCC909
    for e in range(E):                                       # device == the host PerlinGenerator's field
        ox = int(oracle.jitter_u01(seed ^ SALT, e, 8, 0) * 20001.0) - 10000
        oy = int(oracle.jitter_u01(seed ^ SALT, e, 8, 1) * 20001.0) - 10000
        assert -10000 <= ox <= 10000 and -10000 <= oy <= 10000
        field = perlin_noise(W, H, ox, oy, gen.perlin_scale, gen.perlin_octaves, gen.perlin_persistence, gen.perlin_lacunarity)
        np.testing.assert_array_equal(walls[e], (field > gen.wall_density) & ~area[e])
    if E > 1:
        assert not np.array_equal(walls[0], walls[1])       # per-env offsets
    rot = rng.integers(-1, 2, (E, N), dtype=np.int8)
    ph = rng.integers(0, 3, (E, N), dtype=np.int8)
    obs, ast, rew, done = env.step_update(rot, ph)
    o_obs, o_ast, o_rew, o_done = orc.step(rot, ph)
    np.testing.assert_array_equal(_cpu(rew), o_rew.astype(np.float32))


def test_generate_rejects_bad_wall_parameters():
    from antsrl_amd import _lib, config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    env = BatchedAntsEnv(cm.make_cfg(1, 4, 16, 16))
    for bad in (dict(walls="perlin", perlin_octaves=0), dict(walls="perlin", perlin_scale=0.0),
                dict(walls="perlin", wall_density=1.5), dict(wall_density=1.5)):
        with pytest.raises(_lib.AntsrlError):
            env.generate(cm.make_gen(**bad), episode_seed=1)
    g = cm.make_gen()
    g.wall_kind = 7
    with pytest.raises(_lib.AntsrlError, match="wall_kind"):
        env.generate(g, episode_seed=1)
    # np.random.seed's 32 bits (environment_generator.py:55): the last accepted seed, the first refused one, and a base past
    # 0xFFFFFFFF / 5 (where the limit's own subtraction used to wrap: ADVICE r4)
    lim = 0xFFFFFFFF // 5
    env.generate(cm.make_gen(wall_density=0.0, rng="reference"), episode_seed=lim - 1)
    with pytest.raises(_lib.AntsrlError, match="np.random.seed"):
        env.generate(cm.make_gen(wall_density=0.0, rng="reference"), episode_seed=lim)
    far = BatchedAntsEnv(cm.make_cfg(1, 4, 16, 16, env_id_base=900000000))
    with pytest.raises(_lib.AntsrlError, match="np.random.seed"):
        far.generate(cm.make_gen(wall_density=0.0, rng="reference"), episode_seed=1)


def test_device_generator_takes_the_reference_walls_generator():
    """main.py:70-77 with the draws on the GPU: PerlinGenerator(scale=22.0, density=0.3) as the walls
    generator of DeviceEnvironmentGenerator maps to the device's Perlin walls with the same parameters."""
    from antsrl_amd import config as cm
    from antsrl_amd.generator import DeviceEnvironmentGenerator, PerlinGenerator
    from antsrl_amd.rl_api import ExplorationReward, RLApi
    api = RLApi(ExplorationReward(), 1, 1, 40 / 180 * np.pi, 0.05, 0.5)
    g = DeviceEnvironmentGenerator(96, 80, 50, 2, 0, max_steps=50, seed=9, n_envs=3,
                                   walls_generator=PerlinGenerator(scale=22.0, density=0.3))
    assert g.gen.wall_kind == cm.WALLS_PERLIN and g.gen.wall_density == 0.3 and g.gen.perlin_scale == 22.0
    env = g.generate(api)
    obs, ast, state = api.observation()
    assert obs.shape == (150, 7, 7, 6)
    walls = _cpu(api._backend.read_state(cm.S_WALLS))
    assert walls.shape == (3, 96, 80) and 0 < walls.mean() < 0.3
    with pytest.raises(TypeError):
        DeviceEnvironmentGenerator(32, 32, 4, 2, 0, max_steps=5, walls_generator=object())


def _bernoulli_walls(seed, w, h, density=0.05):
    return np.random.default_rng(1000 + seed).random((w, h)) < density


@pytest.mark.parametrize("walls_kind", ["input", "perlin", "none"])
def test_generate_with_the_reference_streams_equals_the_host_generator(walls_kind):
    """ANTSRL_RNG_REFERENCE: env e is drawn like the reference's EnvironmentGenerator(seed = episode_seed + e) —
    Python's and numpy's MT19937 streams on the device.  The host generator (antsrl_amd.generator, pinned to the
    reference's initial states by tests/test_generator.py, itself running on Python's / numpy's own MT19937) draws
    the same episodes: anthill, walls, food, rocks, seeds bit for bit, ant positions to the last bits of cos / sin."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.generator import CirclesGenerator, EmptyGenerator, EnvironmentGenerator, PerlinGenerator
    E, N, W, H, R, seed = 5, 37, 96, 64, 3, 4242

    class Input:
        def __init__(self):
            self.k = 0

        def generate(self, w, h):
            self.k += 1
            return _bernoulli_walls(seed + self.k - 1, w, h)
    wg = {"input": Input(), "perlin": PerlinGenerator(scale=9.0, density=0.15), "none": EmptyGenerator()}[walls_kind]
    host = EnvironmentGenerator(W, H, N, 2, R, CirclesGenerator(7, 3, 8), wg, 100, seed=seed, n_envs=E).draw()
    cfg = cm.make_cfg(E, N, W, H, n_rocks=R, deposit_strength=256.0)
    env = BatchedAntsEnv(cfg)
    if walls_kind == "input":
        gen = cm.make_gen(0.0, 7, 3, 8, walls="input", rng="reference")
        env.generate(gen, seed, walls=np.stack([_bernoulli_walls(seed + e, W, H) for e in range(E)]))
    elif walls_kind == "perlin":
        env.generate(cm.make_gen(0.15, 7, 3, 8, walls="perlin", perlin_scale=9.0, rng="reference"), seed)
    else:
        env.generate(cm.make_gen(0.0, 7, 3, 8, rng="reference"), seed)
    rd = lambda w: env.read_state(w).cpu().numpy()  # noqa: E731
    np.testing.assert_array_equal(rd(cm.S_ANTHILL_XYR), host["anthill_xyr"])
    np.testing.assert_array_equal(rd(cm.S_WALLS), host["walls"])
    np.testing.assert_array_equal(rd(cm.S_FOOD), host["food"])
    np.testing.assert_array_equal(rd(cm.S_SEED), host["seed"].astype(np.float32))
    np.testing.assert_array_equal(rd(cm.S_ROCK_CENTERS), host["rocks"][..., :2])
    np.testing.assert_array_equal(rd(cm.S_ROCK_RW), host["rocks"][..., 2:])
    xyt = rd(cm.S_ANTS_XYT)
    np.testing.assert_array_equal(xyt[..., 2], host["ants_xyt"][..., 2])
    np.testing.assert_allclose(xyt[..., :2], host["ants_xyt"][..., :2], rtol=0, atol=1e-11)
    # and it steps: one observation + step from the generated state
    obs, ast, rew = env.observe()
    assert torch.isfinite(obs).all()


def test_generate_with_the_reference_streams_draws_the_golden_initial_states():
    """Directly against the reference: the initial ants, per-ant seeds and anthill the 13 golden fixtures recorded
    (reference EnvironmentGenerator with the fixture's seed) come out of the device generator."""
    from helpers import fixture_names, load_fixture
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    for name in fixture_names():
        cfg, init, F, meta = load_fixture(name)
        if meta["n_rocks"]:
            continue  # the fixtures' rocks were built by hand (the reference's rock branch raises NameError)
        env = BatchedAntsEnv(cfg)
        env.generate(cm.make_gen(0.0, 6, 3, 6, rng="reference"), int(meta["seed"]))
        np.testing.assert_array_equal(env.read_state(cm.S_ANTHILL_XYR).cpu().numpy()[0], F["init_anthill_xyr"], err_msg=name)
        np.testing.assert_array_equal(env.read_state(cm.S_SEED).cpu().numpy()[0], F["init_seed"].astype(np.float32), err_msg=name)
        xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()[0]
        np.testing.assert_array_equal(xyt[:, 2], F["init_ants_xyt"][:, 2], err_msg=name)
        np.testing.assert_allclose(xyt[:, :2], F["init_ants_xyt"][:, :2], rtol=0, atol=1e-11, err_msg=name)
