"""The reference-shaped Python surface (antsrl_amd.RLApi / EnvironmentGenerator / views) driven
exactly like main.py:69-131 and compared with the reference's golden run of the same seed."""
import random

import numpy as np
import pytest

from helpers import OP_STEP, OP_UPDATE, load_fixture, phero_close
from test_generator import BernoulliWalls, FoodNearAnthill, WALL_DENSITY

pytestmark = pytest.mark.gpu


def build(name, as_numpy=True):
    from antsrl_amd.generator import EnvironmentGenerator
    from antsrl_amd.rl_api import All_Rewards, ExplorationReward, Food_Reward, Reward, RLApi
    cfg, init, F, meta = load_fixture(name)
    seed, w, h = meta["seed"], meta["w"], meta["h"]
    rng = np.random.default_rng(1000 + seed)
    random.seed(seed)
    ax = int(random.random() * w * 0.5 + w * 0.25)
    ay = int(random.random() * h * 0.5 + h * 0.25)
    reward = {"exploration": ExplorationReward, "food": Food_Reward, "none": Reward}.get(meta["reward"])
    reward = All_Rewards(**meta["weights"]) if meta["reward"] == "all" else reward()
    api = RLApi(reward=reward, reward_threshold=1, max_speed=1, max_rot_speed=40 / 180 * np.pi,
                carry_speed_reduction=0.05, backward_speed_reduction=0.5, as_numpy=as_numpy)
    gen = EnvironmentGenerator(w, h, meta["n_ants"], meta["n_phero"], 0,
                               FoodNearAnthill(6, 3, 6, (ax + 2, ay + 1, 6)),
                               BernoulliWalls(WALL_DENSITY.get(name, 0.05), rng), meta["max_time"], seed=seed)
    env = gen.generate(api)
    return api, env, F, meta


@pytest.mark.parametrize("name", ["s02_walls", "s05_all_rewards", "s06_food_reward"])
def test_rlapi_shim_replays_reference_run(name):
    from antsrl_amd.rl_api import Pheromone
    api, env, F, meta = build(name)
    # what agents read (agents/agent.py:22-25, collect_agent_memory.py:129-131)
    assert api.perception_coords.shape[:2] == (7, 7) and len(api.perceived_objects) == 6
    assert api.ants.n_ants == meta["n_ants"]
    n_ph = sum(isinstance(o, Pheromone) for o in api.perceived_objects)
    assert n_ph == 2
    if meta["deposit_strength"] != 1.0:
        api.ants.activate_all_pheromones(np.ones((api.ants.n_ants, n_ph)) * 10)
    for t, op in enumerate(F["ops"]):
        if op == OP_STEP:
            obs, ast, rew, done = api.step(F["rot"][t], F["ph"][t])
            assert obs.shape == F["obs"][t].shape and isinstance(done, bool) and done == bool(F["done"][t])
            np.testing.assert_array_equal(rew, F["reward"][t].astype(np.float32))
            np.testing.assert_array_equal(ast, F["agent_state"][t].astype(np.float32))
            ints = [k for k in range(6) if k not in (1, 2)]
            np.testing.assert_array_equal(obs[..., ints], F["obs"][t][..., ints])
            assert np.abs(obs[..., 1:3] - F["obs"][t][..., 1:3]).max() < 2e-5
        elif op == OP_UPDATE:
            env.update(F["jitter"][t][None])
            assert env.timestep == F["timestep"][t]
    np.testing.assert_allclose(api.ants.ants, F["ants"][-1], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(api.ants.holding, F["holding"][-1])
    food = [o for o in env.objects if type(o).__name__ == "Food"][0]
    np.testing.assert_array_equal(food.qte, F["food"][-1])
    anthill = [o for o in env.objects if type(o).__name__ == "Anthill"][0]
    assert anthill.food == F["anthill_food"][-1]
    for c, p in enumerate(api.ants.pheromones):
        assert phero_close(p.phero, F["phero"][-1][c]).all()
    if meta["reward"] in ("exploration", "all"):
        np.testing.assert_array_equal(api.reward.explored_map, F["explored"][-1].astype(bool))
    snap = env.save_state()  # Environment.save_state, environment.py:36-40
    assert len(snap.objects) == len(env.objects)


def test_rlapi_initial_observation_and_state():
    api, env, F, meta = build("s02_walls")
    obs, ast, state = api.observation()  # main.py:88
    assert obs.shape == (32, 7, 7, 6) and ast.shape == (32, 2) and state.shape == (32, 4)
    assert (state[:, 0] == 0).all() and (state[:, 2:] == 0).all()


def test_rlapi_batched_fold_and_device_tensors():
    """n_envs > 1: env axis folded into the ant axis; as_numpy=False keeps torch tensors on the GPU."""
    import torch
    from antsrl_amd.generator import BernoulliGenerator, CirclesGenerator, EnvironmentGenerator
    from antsrl_amd.rl_api import ExplorationReward, RLApi
    api = RLApi(ExplorationReward(), 1, 1, 40 / 180 * np.pi, 0.05, 0.5, as_numpy=False)
    env = EnvironmentGenerator(64, 64, 16, 2, 3, CirclesGenerator(5, 3, 6), BernoulliGenerator(0.05), 50,
                               seed=7, n_envs=4).generate(api)
    assert api.ants.n_ants == 64 and len(api.perceived_objects) == 7
    rot = torch.randint(-1, 2, (64,), dtype=torch.int8, device="cuda")
    obs, ast, rew, done = api.step(rot, None)
    assert obs.is_cuda and obs.shape == (64, 7, 7, 7) and rew.shape == (64,) and done.shape == (4,)
    env.update()
    assert (env.timestep == 2).all()


def test_saved_states_pickle_in_the_reference_layout():
    """Environment.save_state (environment.py:36-40) per step, as main.py:138-144 does, then the
    pickled file read back: the last snapshot equals the golden run's final state."""
    from antsrl_amd import snapshot as S
    api, env, F, meta = build("s02_walls")
    states = []
    for t, op in enumerate(F["ops"]):
        if op == OP_STEP:
            api.step(F["rot"][t], F["ph"][t])
        elif op == OP_UPDATE:
            env.update(F["jitter"][t][None])
            states.append(env.save_state())
    back = S.loads(S.dumps(states))
    assert len(back) == len(states) and back[-1].timestep == 1 and back[-1].sim_timestep == F["timestep"][-1]
    last = {type(o).__name__: o for o in back[-1].objects}
    np.testing.assert_allclose(last["AntsVisualization"].ants, F["ants"][-1], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(last["AntsVisualization"].holding, F["holding"][-1])
    np.testing.assert_array_equal(last["FoodVisualization"].qte, F["food"][-1].astype(np.uint8))
    assert last["AnthillVisualization"].food == F["anthill_food"][-1]
    assert last["Walls"].map.dtype == bool and last["Walls"].map.shape == (meta["w"], meta["h"])
    np.testing.assert_array_equal(last["RLVisualization"].heatmap, F["explored"][-1].astype(bool))
    ph = [o for o in back[-1].objects if type(o).__name__ == "PheromoneVisualization"]
    assert len(ph) == 2 and ph[0].phero.dtype == np.uint8
    assert np.abs(ph[0].phero.astype(int) - F["phero"][-1][0].astype(np.uint8).astype(int)).max() <= 1


def test_snapshot_of_chosen_envs_from_the_batched_backend():
    import torch
    from antsrl_amd import config as cm, snapshot as S
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(6, 40, 48, 40, n_rocks=3, deposit_strength=256.0)
    benv = BatchedAntsEnv(cfg, torch.device("cuda", 0))
    init = synth_init(cfg, seed=3)
    benv.reset(init)
    rot, ph = random_actions(cfg, 5, seed=1)
    for t in range(5):
        benv.step_update(rot[t], ph[t], None)
    snaps = S.snapshot_batched(benv, [1, 4])
    xyt = benv.read_state(cm.S_ANTS_XYT).cpu().numpy()
    for s, e in zip(snaps, (1, 4)):
        o = {type(x).__name__: x for x in s.objects}
        assert s.timestep == 1 and s.sim_timestep == 6 and (s.w, s.h) == (48, 40)  # snapshot envs restart at 1 (environment.py:36-40)
        np.testing.assert_array_equal(o["AntsVisualization"].ants, xyt[e])
        assert (o["AnthillVisualization"].x, o["AnthillVisualization"].y, o["AnthillVisualization"].radius) == tuple(
            int(v) for v in init["anthill_xyr"][e])
        np.testing.assert_array_equal(o["CircleObstaclesVisualization"].radiuses, init["rocks"][e][:, 2])
        np.testing.assert_array_equal(o["CircleObstaclesVisualization"].weights, init["rocks"][e][:, 3])
        np.testing.assert_array_equal(o["Walls"].map, init["walls"][e].astype(bool))
    assert len(S.loads(S.dumps(snaps))) == 2


def test_outputs_to_host_matches_device_tensors():
    """BatchedAntsEnv.outputs_to_host (one packed device-to-host copy) returns fresh arrays equal to the
    device tensors, with numpy actions going up through the pinned staging buffer; a re-pointed output
    (RewardGather slot) falls back to per-tensor copies."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(3, 37, 48, 40, n_rocks=2, deposit_strength=256.0)
    init = synth_init(cfg, seed=4, n_food_discs=4, food_rmin=2, food_rmax=4)
    a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg)
    a.reset(init)
    b.reset(init)
    rot, ph = random_actions(cfg, 6, seed=3)
    kept = []
    for t in range(6):
        a.step_update(torch.from_numpy(rot[t]).cuda(), torch.from_numpy(ph[t]).cuda())   # device tensors
        b.step_update(rot[t].astype(np.int64), ph[t])                                       # numpy, any int dtype
        out = b.outputs_to_host()
        for got, ref in zip(out, (a.obs, a.agent_state, a.reward, a.done)):
            np.testing.assert_array_equal(got, ref.cpu().numpy())
        kept.append(out[0])
    assert all(k.base is None or k.flags.owndata for k in kept)          # fresh arrays,
    assert not np.array_equal(kept[0], kept[-1])                         # not views of one buffer
    small = b.outputs_to_host(want_obs=False)
    assert small[0] is None and np.array_equal(small[2], a.reward.cpu().numpy())
    b.reward = torch.zeros_like(b.reward)                                 # re-pointed output
    b.step_update(rot[0], ph[0])
    a.step_update(rot[0], ph[0])
    np.testing.assert_array_equal(b.outputs_to_host()[2], a.reward.cpu().numpy())


def test_host_views_read_only_what_they_show():
    """Pheromone.phero reads ONE channel (ANTSRL_S_PHERO_C<i>), equal to that slice of the all-channel read-out;
    Environment.timestep is the library's host mirror (no device round trip) and follows the device counter;
    fractional actions are refused, not truncated."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(3, 20, 48, 40, deposit_strength=256.0)
    env = BatchedAntsEnv(cfg)
    env.reset(synth_init(cfg, seed=4, n_food_discs=4, food_rmin=2, food_rmax=5))
    rot, ph = random_actions(cfg, 4, seed=8)
    for t in range(4):
        env.step_update(rot[t], ph[t], None)
    allc = env.read_state(cm.S_PHERO)
    assert float(allc.abs().sum()) > 0
    for c in range(2):
        assert torch.equal(env.read_state(cm.S_PHERO_C0 + c), allc[:, c])
    with pytest.raises(Exception):
        env.read_state(cm.S_PHERO_C2)
    assert env.query(cm.Q_TIMESTEP) == 5 and (env.read_state(cm.S_TIMESTEP).cpu().numpy() == 5).all()
    with pytest.raises(ValueError):
        env.step(rot[0].astype(np.float64) * 0.5, ph[0])
    with pytest.raises(ValueError):  # torch tensors follow the numpy rule: fractional values are refused ...
        env.step(torch.full((3, 20), 0.5, device=env.device), torch.zeros((3, 20), dtype=torch.int8, device=env.device))
    env.step(torch.zeros((3, 20), device=env.device), torch.zeros((3, 20), dtype=torch.int8, device=env.device))  # ... whole ones taken
    env.step(rot[0].astype(np.float64), ph[0])  # whole-number floats are fine (numpy promotes argmax - 1 to int64 anyway)


def test_caller_added_env_objects_are_updated_in_update_step_order():
    """environment.py:42-47: Environment.update calls every object's update() in stable update_step() order.  Host objects
    the caller added run where their step puts them among the world's objects (tests/test_gpu_update_phases.py), each
    group in stable sorted order."""
    from antsrl_amd.rl_api import EnvObject
    api, env, F, meta = build("s02_walls")
    calls = []

    class Probe(EnvObject):
        def __init__(self, environment, name, step):
            self.name, self.step = name, step
            super().__init__(environment)

        def update_step(self):
            return self.step

        def update(self):
            calls.append((self.name, int(np.asarray(self.environment.timestep).reshape(-1)[0])))

    Probe(env, "late", 2000)
    Probe(env, "early_b", -5)
    Probe(env, "zero_a", 0)
    Probe(env, "early_a", -5)   # same step as early_b, added later: stays behind it (stable sort)
    Probe(env, "zero_b", 0)
    api.observation()
    n = api.ants.n_ants
    api.step(np.zeros(n, dtype=np.int64), np.ones(n, dtype=np.int64))
    t0 = int(np.asarray(env.timestep).reshape(-1)[0])
    env.update()
    assert [c[0] for c in calls] == ["early_b", "early_a", "zero_a", "zero_b", "late"]
    # environment.py:45: timestep += 1 BEFORE any object's update — the early objects see the new value too (the device
    # counter itself moves inside the device update)
    assert [c[1] for c in calls] == [t0 + 1] * 5
    assert int(np.asarray(env.timestep).reshape(-1)[0]) == t0 + 1
    # the golden replay is unaffected by host objects (a second, plain env gives the same state)
    api2, env2, _, _ = build("s02_walls")
    api2.observation()
    api2.step(np.zeros(n, dtype=np.int64), np.ones(n, dtype=np.int64))
    env2.update()
    np.testing.assert_array_equal(api.ants.ants, api2.ants.ants)
    snap = env.save_state()  # host objects without a visualisation copy add nothing
    assert len(snap.objects) == len(env2.save_state().objects)


def test_overridden_view_update_is_called_between_the_phases():
    """a caller's subclass of a device view with its OWN update() is a host object (its update() is called); a
    plain view is not (the kernels update it); a host object whose step falls between the world's objects runs between the
    device update's phases (round 5: antsrl_update_phase; tests/test_gpu_update_phases.py pins WHAT it sees there)."""
    import warnings
    from antsrl_amd.rl_api import EnvObject, Walls
    api, env, F, meta = build("s02_walls")
    calls = []

    class MovingWalls(Walls):
        def update(self):
            calls.append("moving_walls")

    class Mid(EnvObject):
        def update_step(self):
            return 500

        def update(self):
            calls.append("mid")

    MovingWalls(env)
    Mid(env)
    api.observation()
    n = api.ants.n_ants
    api.step(np.zeros(n, dtype=np.int64), np.ones(n, dtype=np.int64))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env.update()
        env.update()
    assert calls == ["moving_walls", "mid", "moving_walls", "mid"]  # Walls.update_step() is -1: right behind the world's Walls
    assert not [x for x in w if "update_step()" in str(x.message)]  # (no "cannot interleave" warning any more)


def test_action_arrays_must_be_whole_numbers_in_int8_range():
    """BatchedAntsEnv._integral: numpy and torch inputs follow ONE rule — whole numbers of any dtype are accepted,
    fractional values and values outside int8 are refused (never truncated or wrapped)."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import synth_init
    cfg = cm.make_cfg(2, 8, 32, 32)
    env = BatchedAntsEnv(cfg)
    env.reset(synth_init(cfg, seed=1, n_food_discs=2, food_rmin=2, food_rmax=3))
    ok = [np.ones((2, 8)), np.ones((2, 8), dtype=np.int64), torch.ones((2, 8)), torch.ones((2, 8), dtype=torch.int64),
          torch.ones((2, 8), dtype=torch.float64, device="cuda"), torch.ones((2, 8), dtype=torch.int8, device="cuda")]
    outs = []
    for a in ok:
        env.reset(synth_init(cfg, seed=1, n_food_discs=2, food_rmin=2, food_rmax=3))
        outs.append(env.step(a, a)[0].clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    for bad in (np.full((2, 8), 0.5), torch.full((2, 8), 0.5), torch.full((2, 8), 0.5, device="cuda")):
        with pytest.raises(ValueError):
            env.step(bad, None)
    for bad in (np.full((2, 8), 300), torch.full((2, 8), 300, dtype=torch.int64), torch.full((2, 8), -200, dtype=torch.int32, device="cuda")):
        with pytest.raises(ValueError):
            env.step(bad, None)
    # validate_actions = False: no check, no host synchronisation (the caller vouches for its actions)
    env.validate_actions = False
    env.reset(synth_init(cfg, seed=1, n_food_discs=2, food_rmin=2, food_rmax=3))
    assert torch.equal(env.step(torch.ones((2, 8), dtype=torch.int64, device="cuda"), torch.ones((2, 8), dtype=torch.int64, device="cuda"))[0], outs[0])
