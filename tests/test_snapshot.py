"""State snapshots in the reference's pickled layout (antsrl_amd/snapshot.py).
The CPU tests use stand-in modules under the reference's import paths to play the viewer's
process: `pickle.load` must find `environment.ants.AntsVisualization` etc. and nothing else."""
import pickle
import sys
import types

import numpy as np
import pytest

from antsrl_amd import snapshot as S

REF_CLASSES = {
    "environment.environment": ["Environment"], "environment.ants": ["AntsVisualization"],
    "environment.pheromone": ["PheromoneVisualization"], "environment.food": ["FoodVisualization"],
    "environment.anthill": ["AnthillVisualization"], "environment.circle_obstacles": ["CircleObstaclesVisualization"],
    "environment.RL_api": ["RLVisualization"], "environment.walls": ["Walls"],
}


def _arrays(seed=0, W=12, H=9, N=5, R=2):
    r = np.random.default_rng(seed)
    return dict(
        ants_xyt=r.random((N, 3)) * [W, H, 6.28], mandibles=r.integers(0, 2, N), holding=r.random(N) * 5,
        reward_state=r.integers(0, 256, N), phero=r.random((2, W, H)).astype(np.float32) * 255,
        phero_colors=S.PHERO_COLORS, phero_max_val=255.0, food=r.integers(0, 9, (W, H)).astype(np.float32),
        walls=r.integers(0, 2, (W, H)), anthill_xyr=(4, 3, 2), anthill_food=17.0,
        rock_centers=r.random((R, 2)) * 8, rock_radiuses=np.array([1.5, 2.0]), rock_weights=np.array([3.0, 9.0]),
        heatmap=r.integers(0, 2, (W, H)).astype(bool))


@pytest.fixture
def viewer_modules():
    """Empty classes under the reference's module paths, as the viewer's process would have them."""
    saved = {k: sys.modules.get(k) for k in ["environment"] + list(REF_CLASSES)}
    pkg = types.ModuleType("environment")
    pkg.__path__ = []
    sys.modules["environment"] = pkg
    made = {}
    for mod, names in REF_CLASSES.items():
        m = types.ModuleType(mod)
        for n in names:
            made[(mod, n)] = type(n, (object,), {"__module__": mod})
            setattr(m, n, made[(mod, n)])
        sys.modules[mod] = m
    yield made
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_layout_and_order():
    a = _arrays()
    env = S.snapshot_from_arrays(12, 9, 500, 37, **a)
    kinds = [type(o).__name__ for o in env.objects]
    assert kinds == ["AnthillVisualization", "Walls", "FoodVisualization", "CircleObstaclesVisualization",
                     "AntsVisualization", "PheromoneVisualization", "PheromoneVisualization", "RLVisualization"]
    # (save_state builds a fresh Environment: its own timestep is always 1, environment.py:36-40 — pinned by
    # tests/golden/contract/snapshot_ref.pkl; the simulation's step rides along as sim_timestep)
    assert (env.w, env.h, env.max_time, env.timestep, env.sim_timestep) == (12, 9, 500, 1, 37)
    ph = env.objects[5]
    assert ph.phero.dtype == np.uint8 and ph.phero.shape == (12, 9) and ph.max_val == 255.0  # pheromone.py:17
    np.testing.assert_array_equal(ph.phero, a["phero"][0].astype(np.uint8))
    assert env.objects[2].qte.dtype == np.uint8                                               # food.py:10
    assert env.objects[1].map.dtype == bool
    assert all(o.environment is env for o in env.objects)
    with pytest.raises(TypeError):
        S.FoodVisualization(None, qty=1)


def test_pickle_names_the_reference_classes_and_loads_in_a_viewer_process(viewer_modules):
    states = [S.snapshot_from_arrays(12, 9, 500, t, **_arrays(t)) for t in (1, 2, 3)]
    blob = S.dumps(states)
    assert b"antsrl_amd" not in blob
    for mod, names in REF_CLASSES.items():
        for n in names:
            assert (mod + "\n" + n + "\n").encode() in blob
    loaded = pickle.loads(blob)  # plain pickle: only the viewer-side classes exist for it
    assert len(loaded) == 3 and type(loaded[0]) is viewer_modules[("environment.environment", "Environment")]
    for t, env in zip((1, 2, 3), loaded):
        a = _arrays(t)
        assert env.timestep == 1 and env.sim_timestep == t and env.w == 12 and env.h == 9
        ants = [o for o in env.objects if type(o) is viewer_modules[("environment.ants", "AntsVisualization")]]
        assert len(ants) == 1 and ants[0].environment is env
        np.testing.assert_array_equal(ants[0].ants, a["ants_xyt"])
        np.testing.assert_array_equal(ants[0].holding, a["holding"])
        hill = [o for o in env.objects if type(o).__name__ == "AnthillVisualization"][0]
        assert (hill.x, hill.y, hill.radius, hill.food) == (4, 3, 2, 17.0)
        rocks = [o for o in env.objects if type(o).__name__ == "CircleObstaclesVisualization"][0]
        np.testing.assert_array_equal(rocks.weights, [3.0, 9.0])
        rl = [o for o in env.objects if type(o).__name__ == "RLVisualization"][0]
        np.testing.assert_array_equal(rl.heatmap, a["heatmap"])


def test_round_trip_without_any_reference_module(tmp_path):
    assert "environment" not in sys.modules or getattr(sys.modules["environment"], "__antsrl_alias__", False)
    states = [S.snapshot_from_arrays(12, 9, 500, 5, **_arrays(5))]
    f = tmp_path / "saved.arl"
    with open(f, "wb") as fh:
        S.dump(states, fh)
    with open(f, "rb") as fh:
        back = S.load(fh)
    assert type(back[0]) is S.Environment and type(back[0].objects[4]) is S.AntsVisualization
    np.testing.assert_array_equal(back[0].objects[4].ants, states[0].objects[4].ants)
    no_rocks = dict(_arrays(1), rock_centers=None)
    env = S.snapshot_from_arrays(12, 9, 500, 1, **no_rocks)
    assert "CircleObstaclesVisualization" not in [type(o).__name__ for o in env.objects]
