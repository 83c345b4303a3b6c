"""State snapshots in the reference's pickled layout.

The reference's replay viewer (`gui/visualize.py:125`) unpickles a list of `Environment` snapshots
whose objects are told apart with `isinstance(obj, AntsVisualization)` etc. (`gui/visualize.py:
78-89,190-246`).  A pickle names classes by `module.qualname`, so a file written here loads in the
reference's process as ITS classes if the names and the attribute layout match:

    environment.environment.Environment        w, h, objects, max_time, timestep     (environment.py:21-40)
    environment.ants.AntsVisualization          ants[N,3], mandibles, holding, reward_state (ants.py:8-14)
    environment.pheromone.PheromoneVisualization color, max_val, phero uint8[W,H]     (pheromone.py:12-17)
    environment.food.FoodVisualization          qte uint8[W,H]                        (food.py:7-10)
    environment.anthill.AnthillVisualization    x, y, radius, food                    (anthill.py:8-14)
    environment.circle_obstacles.CircleObstaclesVisualization  centers, radiuses, weights (circle_obstacles.py:8-13)
    environment.RL_api.RLVisualization          heatmap (or None)                     (RL_api.py:17-20)
    environment.walls.Walls                     w, h, map bool[W,H]                   (walls.py:9-17: snapshots keep the object itself)

The classes below carry those names; `dump` / `dumps` write them with a pickler that emits the
reference's module paths whether or not such modules exist in this process (nothing of the
reference is imported).  `load` / `loads` read such a file back into these classes (round trips,
tests), again without the reference.
"""
from __future__ import annotations

import io
import pickle
from typing import Iterable, List, Optional

import numpy as np


def _ref_class(name: str, module: str, fields: tuple):
    def __init__(self, env, **kw):
        missing = [f for f in fields if f not in kw]
        if missing or len(kw) != len(fields):
            raise TypeError("%s takes exactly the fields %s" % (name, ", ".join(fields)))
        self.environment = env
        self.__dict__.update(kw)
        if env is not None:
            env.add_object(self)

    cls = type(name, (object,), {"__init__": __init__, "__module__": module, "_fields": fields})
    cls.__qualname__ = name
    return cls


class Environment:
    """environment/environment.py:21-40 (the snapshot side: plain attributes only)."""
    __module__ = "environment.environment"

    def __init__(self, w, h, max_time, timestep=1):
        self.w, self.h = int(w), int(h)
        self.objects: List[object] = []
        self.max_time = max_time
        self.timestep = int(timestep)

    def add_object(self, obj):
        self.objects.append(obj)


AntsVisualization = _ref_class("AntsVisualization", "environment.ants", ("ants", "mandibles", "holding", "reward_state"))
PheromoneVisualization = _ref_class("PheromoneVisualization", "environment.pheromone", ("color", "max_val", "phero"))
FoodVisualization = _ref_class("FoodVisualization", "environment.food", ("qte",))
AnthillVisualization = _ref_class("AnthillVisualization", "environment.anthill", ("x", "y", "radius", "food"))
CircleObstaclesVisualization = _ref_class("CircleObstaclesVisualization", "environment.circle_obstacles",
                                          ("centers", "radiuses", "weights"))
RLVisualization = _ref_class("RLVisualization", "environment.RL_api", ("heatmap",))
Walls = _ref_class("Walls", "environment.walls", ("w", "h", "map"))

_CLASSES = (Environment, AntsVisualization, PheromoneVisualization, FoodVisualization, AnthillVisualization,
            CircleObstaclesVisualization, RLVisualization, Walls)
_BY_NAME = {(c.__module__, c.__qualname__): c for c in _CLASSES}


class _RefPickler(pickle._Pickler):  # the pure-Python pickler: save_global can be overridden
    def save_global(self, obj, name=None):
        if obj in _CLASSES:
            # GLOBAL <module>\n<name>\n — the reference's import path, no lookup in this process
            self.write(pickle.GLOBAL + obj.__module__.encode() + b"\n" + obj.__qualname__.encode() + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


_PLACEHOLDERS = {}


class _RefUnpickler(pickle.Unpickler):
    """Reads snapshot files without the reference: the snapshot classes resolve to the ones above; any OTHER
    `environment.*` class becomes an attribute-bag placeholder.  (The reference's own files contain them:
    Walls.visualize_copy returns the live Walls object, walls.py:16-17, whose `.environment` drags the whole
    live Environment — RLApi, Ants, Anthill, Food, Pheromone, the reward — into the pickle.)"""

    def find_class(self, module, name):
        cls = _BY_NAME.get((module, name))
        if cls is not None:
            return cls
        if module == "environment" or module.startswith("environment."):
            key = (module, name)
            if key not in _PLACEHOLDERS:
                _PLACEHOLDERS[key] = type(name, (object,), {"__module__": module})
            return _PLACEHOLDERS[key]
        return super().find_class(module, name)


def dumps(states) -> bytes:
    """Pickles a list of snapshots (what main.py:144 writes to saved/<name>)."""
    buf = io.BytesIO()
    _RefPickler(buf, protocol=2).dump(states)
    return buf.getvalue()


def dump(states, file) -> None:
    file.write(dumps(states))


def loads(data: bytes):
    return _RefUnpickler(io.BytesIO(data)).load()


def load(file):
    return _RefUnpickler(file).load()


def snapshot_from_arrays(w: int, h: int, max_time: int, timestep: int, *, ants_xyt, mandibles, holding,
                         reward_state, phero, phero_colors, phero_max_val, food, walls, anthill_xyr,
                         anthill_food, rock_centers=None, rock_radiuses=None, rock_weights=None,
                         heatmap=None, has_rl: bool = True) -> Environment:
    """One environment's state -> a snapshot in the reference's layout and object order
    (generator order, environment_generator.py:57-101: anthill, walls, food, rocks, ants,
    pheromones, RL api)."""
    # Environment.save_state (environment.py:36-40) builds a FRESH Environment(w, h, max_time): the snapshot's own
    # `timestep` is always 1 in the reference's files (pinned by tests/golden/contract/snapshot_ref.pkl); the
    # simulation's step rides along as an extra attribute the viewer ignores
    env = Environment(w, h, max_time, 1)
    env.sim_timestep = int(timestep)
    AnthillVisualization(env, x=int(anthill_xyr[0]), y=int(anthill_xyr[1]), radius=int(anthill_xyr[2]),
                         food=float(anthill_food))
    Walls(env, w=int(w), h=int(h), map=np.asarray(walls).astype(bool))
    FoodVisualization(env, qte=np.asarray(food).astype(np.uint8))  # food.py:10
    if rock_centers is not None and len(rock_centers):
        CircleObstaclesVisualization(env, centers=np.array(rock_centers, dtype=float),
                                     radiuses=np.array(rock_radiuses, dtype=float),
                                     weights=np.array(rock_weights, dtype=float))
    # (mandibles: int64 0/1, what Ants.mandibles holds after the first update_mandibles, ants.py:103-107)
    AntsVisualization(env, ants=np.array(ants_xyt, dtype=float), mandibles=np.array(mandibles).astype(np.int64),
                      holding=np.array(holding, dtype=float), reward_state=np.array(reward_state).astype(np.uint8))
    for c in range(len(phero)):
        PheromoneVisualization(env, color=tuple(phero_colors[c]), max_val=phero_max_val,
                               phero=np.asarray(phero[c]).astype(np.uint8))  # pheromone.py:17
    if has_rl:
        RLVisualization(env, heatmap=None if heatmap is None else np.array(heatmap))
    return env


PHERO_COLORS = ((255, 64, 0), (64, 64, 255), (100, 255, 100), (255, 255, 64))


def snapshot_batched(benv, env_indices: Optional[Iterable[int]] = None, phero_colors=PHERO_COLORS,
                     with_heatmap: bool = True) -> List[Environment]:
    """Snapshots of the chosen environments of a BatchedAntsEnv (default: env 0), read back through
    antsrl_read_state in one pass per state array."""
    from . import config as cm

    c = benv.cfg
    idx = [0] if env_indices is None else list(env_indices)
    rd = lambda which: benv.read_state(which).cpu().numpy()  # noqa: E731
    xyt, hold, mand, rst = rd(cm.S_ANTS_XYT), rd(cm.S_HOLDING), rd(cm.S_MANDIBLES), rd(cm.S_REWARD_STATE)
    ph, food, walls = rd(cm.S_PHERO), rd(cm.S_FOOD), rd(cm.S_WALLS)
    hill_food, ts = rd(cm.S_ANTHILL_FOOD), rd(cm.S_TIMESTEP)
    rocks = rd(cm.S_ROCK_CENTERS) if c.n_rocks else None
    expl = rd(cm.S_EXPLORED) if with_heatmap and c.reward_kind in (cm.REWARD_EXPLORATION, cm.REWARD_ALL) else None
    xyr = rd(cm.S_ANTHILL_XYR)
    rw = rd(cm.S_ROCK_RW) if c.n_rocks else None
    out = []
    for e in idx:
        out.append(snapshot_from_arrays(
            c.w, c.h, c.max_time, int(ts[e]), ants_xyt=xyt[e], mandibles=mand[e], holding=hold[e], reward_state=rst[e],
            phero=ph[e], phero_colors=phero_colors, phero_max_val=(c.phero_max_val if c.has_max_val else None),
            food=food[e], walls=walls[e], anthill_xyr=xyr[e], anthill_food=hill_food[e],
            rock_centers=None if rocks is None else rocks[e],
            rock_radiuses=None if rw is None else rw[e][:, 0],
            rock_weights=None if rw is None else rw[e][:, 1],
            heatmap=None if expl is None else expl[e].astype(bool)))
    return out
