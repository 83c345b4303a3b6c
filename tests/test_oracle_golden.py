"""Pins the CPU oracle (oracle/antsrl_oracle.c) against golden vectors recorded from the real
reference (tests/golden/make_golden.py).  float64 vs float64: everything bit-exact, except the
pheromone grid under a patched diffusion filter (scipy's summation order is not restated)."""
import numpy as np
import pytest

from helpers import OP_OBSERVE, OP_STEP, OP_UPDATE, fixture_names, load_fixture
from oracle.oracle import Oracle


def replay(name, n_envs=1, n_threads=1):
    cfg, init, F, meta = load_fixture(name, n_envs)
    o = Oracle(cfg, init, n_threads=n_threads)
    np.testing.assert_array_equal(o.anthill_area[0], F["init_anthill_area"].astype(np.uint8))
    if meta["deposit_strength"] != 1.0:
        o.set_activation(np.broadcast_to(F["init_activation"], o.activation.shape))
    exact_phero = np.count_nonzero(np.array(meta["filter"])) <= 1
    for t, op in enumerate(F["ops"]):
        ctx = "%s op %d kind %d" % (name, t, op)
        if op == OP_STEP:
            rot = np.broadcast_to(F["rot"][t], (n_envs, cfg.n_ants)) if F["has_rot"][t] else None
            ph = np.broadcast_to(F["ph"][t], (n_envs, cfg.n_ants)) if F["has_ph"][t] else None
            obs, ast, rew, done = o.step(rot, ph)
            assert (done == F["done"][t]).all(), ctx
        elif op == OP_OBSERVE:
            obs, ast, rew = o.observe()
        else:
            hits = o.update(np.broadcast_to(F["jitter"][t], (n_envs, cfg.n_ants)))
            assert (hits == F["hits"][t]).all(), ctx
        for e in {0, n_envs - 1}:
            if op != OP_UPDATE:
                np.testing.assert_array_equal(obs[e], F["obs"][t], err_msg=ctx)
                np.testing.assert_array_equal(ast[e], F["agent_state"][t], err_msg=ctx)
                np.testing.assert_array_equal(rew[e], F["reward"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.ants_xyt[e], F["ants"][t], err_msg=ctx)
            np.testing.assert_array_equal(np.stack([o.prev_x[e], o.prev_y[e]], -1), F["prev"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.holding[e], F["holding"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.mandibles[e], F["mandibles"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.activation[e], F["activation"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.reward_state[e], F["reward_state"][t], err_msg=ctx)
            np.testing.assert_array_equal(o.food[e], F["food"][t].astype(np.float64), err_msg=ctx)
            np.testing.assert_array_equal(o.timestep[e], F["timestep"][t], err_msg=ctx)
            assert o.anthill_food[e] == F["anthill_food"][t], ctx
            if meta["reward"] in ("exploration", "all"):
                np.testing.assert_array_equal(o.explored[e], F["explored"][t], err_msg=ctx)
            if cfg.n_rocks:
                np.testing.assert_array_equal(o.rock_centers[e], F["rock_centers"][t], err_msg=ctx)
            if F["stored"][t]:
                if exact_phero:
                    np.testing.assert_array_equal(o.phero[e], F["phero"][t], err_msg=ctx)
                else:
                    np.testing.assert_allclose(o.phero[e], F["phero"][t], rtol=1e-12, atol=1e-300, err_msg=ctx)


@pytest.mark.parametrize("name", fixture_names())
def test_oracle_matches_reference_golden(name):
    replay(name)


def test_oracle_batched_threads_match():
    # env-major batching + OpenMP must not change any env's result
    replay("s03_walls_rocks", n_envs=5, n_threads=3)
