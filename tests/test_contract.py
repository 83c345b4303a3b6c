"""Rows (f)3 / (f)4 of SURVEY.md §8 and the agent contract, PINNED to the reference: the fixtures under
tests/golden/contract/ were produced by tests/golden/make_contract_golden.py, which drives the reference's own
`ReplayMemory` (agents/replay_memory.py:60-114), `Environment.save_state` (environment.py:36-40 + the
*Visualization classes), and `CollectAgentMemory` through the loop of main.py:86-105.

CPU tests: the replay memory (on a CPU device), the snapshot layout against the reference's own pickle, the
recorded agent contract against the shim's surface.  GPU tests: the same replay on the device, and the
reference agent's recorded run replayed through antsrl_amd.RLApi."""
import json
import os
import pickle
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CONTRACT = os.path.join(HERE, "golden", "contract")


def _replay_into(mem, R):
    for b in range(int(R["n_batches"])):
        ph = R["b%d_ph" % b] if bool(R["b%d_has_ph" % b]) else None
        mem.extend(R["b%d_states" % b], R["b%d_agent_states" % b], (R["b%d_rot" % b], ph), R["b%d_rewards" % b],
                   R["b%d_new_states" % b], R["b%d_new_agent_states" % b], bool(R["b%d_done" % b]))
        assert (mem.head, mem.fill, len(mem)) == (int(R["b%d_head" % b]), int(R["b%d_fill" % b]), int(R["b%d_len" % b]))


def _check_replay(mem, R):
    for name in ("states", "agent_states", "actions", "rewards", "new_states", "new_agent_states", "dones"):
        got = getattr(mem, name).cpu().numpy()
        assert got.dtype == R[name].dtype, name
        np.testing.assert_array_equal(got, R[name], err_msg=name)
    got = mem[R["getitem_idx"].tolist()]
    for k, name in enumerate(("states", "agent_states", "actions", "rewards", "new_states", "new_agent_states", "dones")):
        np.testing.assert_array_equal(got[k].cpu().numpy(), R["getitem_" + name], err_msg="getitem " + name)


def test_replay_memory_matches_the_reference_arrays_cpu():
    """DeviceReplayMemory fed the batches the reference's ReplayMemory was fed: identical arrays (values AND
    dtypes: float32 / int64 / bool, replay_memory.py:18-24), head, fill, len and __getitem__ results."""
    from antsrl_amd.replay import DeviceReplayMemory
    R = dict(np.load(os.path.join(CONTRACT, "replay_ref.npz")))
    assert str(R["actions_dtype"]) == "torch.int64"
    mem = DeviceReplayMemory(int(R["max_len"]), tuple(R["obs_space"]), tuple(R["agent_space"]), tuple(R["action_space"]),
                             device="cpu")
    _replay_into(mem, R)
    _check_replay(mem, R)
    # per-env `done` of shape [E] folded over the ant axis (batched envs)
    mem2 = DeviceReplayMemory(12, (2,), (1,), (2,), device="cpu")
    E, N = 3, 4
    mem2.extend(np.zeros((E * N, 2), np.float32), np.zeros((E * N, 1), np.float32), (np.zeros(E * N, np.int64), None),
                np.zeros(E * N, np.float32), np.zeros((E * N, 2), np.float32), np.zeros((E * N, 1), np.float32),
                np.array([True, False, True]))
    np.testing.assert_array_equal(mem2.dones.numpy(), np.repeat([True, False, True], N))


@pytest.mark.gpu
def test_replay_memory_matches_the_reference_arrays_gpu():
    from antsrl_amd.replay import DeviceReplayMemory
    R = dict(np.load(os.path.join(CONTRACT, "replay_ref.npz")))
    mem = DeviceReplayMemory(int(R["max_len"]), tuple(R["obs_space"]), tuple(R["agent_space"]), tuple(R["action_space"]))
    assert mem.states.is_cuda
    _replay_into(mem, R)
    _check_replay(mem, R)


def test_snapshot_layout_matches_the_reference_pickle():
    """(1) The reference's OWN pickle (main.py:143-144) loads into antsrl_amd.snapshot's classes and equals, field by
    field, the snapshot rebuilt from the same state arrays.  (2) The generating script unpickled OUR pickle with the
    reference's real environment.* classes importable and recorded equality with the reference's objects."""
    from antsrl_amd import snapshot as S
    chk = json.load(open(os.path.join(CONTRACT, "snapshot_check.json")))
    assert chk["ok"] is True and all(v is True for v in chk["field_equal"].values()), chk
    assert chk["reference_classes"] == {
        "AnthillVisualization": "environment.anthill", "Walls": "environment.walls", "FoodVisualization": "environment.food",
        "CircleObstaclesVisualization": "environment.circle_obstacles", "AntsVisualization": "environment.ants",
        "PheromoneVisualization": "environment.pheromone", "RLVisualization": "environment.RL_api"}
    A = dict(np.load(os.path.join(CONTRACT, "snapshot_ref.npz")))
    with open(os.path.join(CONTRACT, "snapshot_ref.pkl"), "rb") as f:
        ref_states = S.load(f)
    assert len(ref_states) == int(A["n"])
    for i, ref in enumerate(ref_states):
        g = lambda k: A["s%d_%s" % (i, k)]  # noqa: E731
        ours = S.snapshot_from_arrays(
            int(A["w"]), int(A["h"]), int(A["max_time"]), int(g("timestep")), ants_xyt=g("ants_xyt"), mandibles=g("mandibles"),
            holding=g("holding"), reward_state=g("reward_state"), phero=g("phero"), phero_colors=A["phero_colors"].tolist(),
            phero_max_val=float(A["phero_max_val"]), food=g("food"), walls=g("walls"), anthill_xyr=g("anthill_xyr"),
            anthill_food=float(g("anthill_food")), rock_centers=g("rock_centers"), rock_radiuses=g("rock_radiuses"),
            rock_weights=g("rock_weights"), heatmap=None)
        assert type(ref) is S.Environment and (ref.w, ref.h, ref.max_time, ref.timestep) == (ours.w, ours.h, ours.max_time, ours.timestep)
        # the reference lists every visualisation copy twice (environment.py:8,39), Walls once; ours lists each once
        seen, ref_objs = set(), []
        for o in ref.objects:
            if id(o) not in seen and not any(o is p for p in ref_objs):
                ref_objs.append(o)
        by_kind = lambda objs, kind: [o for o in objs if type(o).__name__ == kind]  # noqa: E731
        for kind in ("AnthillVisualization", "Walls", "FoodVisualization", "CircleObstaclesVisualization", "AntsVisualization",
                     "PheromoneVisualization", "RLVisualization"):
            r = by_kind(ref.objects, kind)
            r = r[::2] if kind != "Walls" else r
            o = by_kind(ours.objects, kind)
            assert len(r) == len(o) and len(o) >= 1, kind
            for ro, oo in zip(r, o):
                for k, v in vars(ro).items():
                    if k == "environment" or (kind == "RLVisualization" and k == "heatmap"):
                        continue
                    w = getattr(oo, k)
                    if isinstance(v, np.ndarray):
                        assert np.asarray(w).dtype == v.dtype and np.asarray(w).shape == v.shape, (kind, k)
                        np.testing.assert_array_equal(w, v, err_msg="%s.%s" % (kind, k))
                    else:
                        assert w == v, (kind, k, w, v)


def test_agent_contract_surface():
    """Everything the reference's agent touched on `rl_api` (recorded by tracing CollectAgentMemory.setup /
    initialize / get_action against the reference) exists on antsrl_amd.RLApi with the same names."""
    C = json.load(open(os.path.join(CONTRACT, "agent_contract.json")))
    assert C["rl_api_attributes_read"] == ["rl_api.ants", "rl_api.ants.activate_all_pheromones", "rl_api.ants.n_ants",
                                           "rl_api.perceived_objects", "rl_api.perception_coords"]
    assert C["observation_space"] == [7, 7, 6] and C["agent_space"] == [2] and C["action_space"] == [2]
    assert [a["dtype"] for a in C["action"][:2]] == ["int64", "int64"] and C["action"][0]["shape"] == [C["n_ants"]]
    assert C["step_returns"]["done"] == "bool" and C["episode_reward_dtype"] == "float64"
    from antsrl_amd import rl_api
    api = rl_api.RLApi(rl_api.Reward(), 1, 1, 40 / 180 * np.pi, 0.05, 0.5)  # host-only until generate()
    for name in ("ants", "perceived_objects", "perception_coords", "save_perceptive_field", "step", "observation",
                 "setup_perception", "register_ants"):
        assert hasattr(api, name), name
    assert callable(getattr(rl_api.Ants, "activate_all_pheromones"))


class _BernoulliWalls:  # the stand-in of make_contract_golden.py
    def __init__(self, density, rng):
        self.density, self.rng = density, rng

    def generate(self, w, h):
        return self.rng.random((w, h)) < self.density


@pytest.mark.gpu
def test_reference_agent_run_replays_through_the_shim():
    """The run main.py:86-131 made with the reference's CollectAgentMemory (fixed seeds): same episode seed and
    generators -> the shim draws the same initial state; the agent's recorded int64 actions go into api.step exactly
    as `api.step(*action[:2])`; observations, agent_state, reward and done come back equal to the reference's
    (float32 instead of float64 — what the agent's torch.Tensor(state) makes of them anyway, collect_agent_memory.py:194)
    and accumulate into a float64 episode reward like main.py:100."""
    from antsrl_amd.generator import CirclesGenerator, EnvironmentGenerator
    from antsrl_amd.rl_api import All_Rewards, Pheromone, RLApi
    C = json.load(open(os.path.join(CONTRACT, "agent_contract.json")))
    F = dict(np.load(os.path.join(CONTRACT, "agent_contract.npz")))
    n = C["n_ants"]
    api = RLApi(reward=All_Rewards(fct_explore=1, fct_food=2, fct_anthill=10, fct_explore_holding=1, fct_headinganthill=3),
                reward_threshold=1, max_speed=1, max_rot_speed=40 / 180 * np.pi, carry_speed_reduction=0.05,
                backward_speed_reduction=0.5)
    env = EnvironmentGenerator(64, 64, n, 2, 0, CirclesGenerator(6, 3, 6), _BernoulliWalls(0.04, np.random.default_rng(1021)),
                               int(F["max_time"]), seed=21).generate(api)
    np.testing.assert_array_equal(api.ants.ants, F["init_ants_xyt"])
    # Agent.setup (agents/agent.py:22-25) and CollectAgentMemory.initialize (collect_agent_memory.py:129-131)
    assert (api.perception_coords.shape[0], api.perception_coords.shape[1], len(api.perceived_objects)) == tuple(C["observation_space"])
    assert api.ants.n_ants == n and [type(o).__name__ for o in api.perceived_objects] == C["perceived_objects"]
    n_ph = len([o for o in api.perceived_objects if isinstance(o, Pheromone)])
    api.ants.activate_all_pheromones(np.ones((n, n_ph)) * 10)
    obs, agent_state, state = api.observation()  # main.py:88
    assert list(obs.shape) == C["first_observation"]["obs"]["shape"] and list(agent_state.shape) == C["first_observation"]["agent_state"]["shape"]
    ints = [0, 3, 4, 5]
    np.testing.assert_array_equal(np.asarray(obs)[..., ints], F["obs0"][..., ints])
    assert np.abs(np.asarray(obs)[..., 1:3] - F["obs0"][..., 1:3]).max() < 2e-5
    episode_reward = np.zeros(n)
    for s in range(len(F["rot"])):
        rot, ph = F["rot"][s], F["ph"][s]
        assert rot.dtype == np.int64 and ph.dtype == np.int64
        new_state, new_agent_state, reward, done = api.step(rot, ph)  # main.py:98
        assert isinstance(done, bool) and done == bool(F["done"][s])
        assert list(np.asarray(new_state).shape) == C["step_returns"]["obs"]["shape"]
        np.testing.assert_array_equal(np.asarray(reward), F["reward"][s].astype(np.float32))
        np.testing.assert_array_equal(np.asarray(new_agent_state), F["agent_state"][s].astype(np.float32))
        np.testing.assert_array_equal(np.asarray(new_state)[..., ints], F["obs"][s][..., ints])
        assert np.abs(np.asarray(new_state)[..., 1:3] - F["obs"][s][..., 1:3]).max() < 2e-5
        episode_reward += reward  # main.py:100
        env.update(F["jitter"][s][None])  # main.py:131, with the draws the reference consumed
    assert episode_reward.dtype == np.float64
    np.testing.assert_allclose(episode_reward, F["reward"].sum(axis=0), rtol=0, atol=1e-4)
