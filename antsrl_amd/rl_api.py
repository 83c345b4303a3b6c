"""Host-side mirror of the reference's RLApi / Environment surface.

Same class names, constructor arguments, method names and return tuples as
environment/RL_api.py, environment/environment.py and the object classes the agents touch,
so the reference's driver loop (main.py:69-131) and its agents run unchanged:

    api = RLApi(reward=ExplorationReward(), reward_threshold=1, max_speed=1,
                max_rot_speed=40/180*np.pi, carry_speed_reduction=0.05, backward_speed_reduction=0.5)
    env = EnvironmentGenerator(w, h, n_ants, 2, 0, food_gen, walls_gen, max_steps, seed).generate(api)
    obs, agent_state, state = api.observation()
    obs, agent_state, reward, done = api.step(rotation, pheromone); env.update()

Underneath there is no numpy simulation: every call goes through the C-ABI to the HIP kernels
(antsrl_amd.batched.BatchedAntsEnv).  The object classes here are *views*: reading
`pheromone.phero`, `food.qte`, `ants.ants` ... copies that piece of state back from the GPU.

Batching: one RLApi can drive E environments (EnvironmentGenerator(..., n_envs=E)).  With
E == 1 every array has the reference's shape.  With E > 1 the env axis is folded into the ant
axis ([E*N, 7, 7, K], n_ants == E*N) so per-ant agents still work unchanged; `done` is then a
bool array [E].  `as_numpy=False` keeps outputs as torch tensors on the GPU (no PCIe copy).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import config as cm

AX = np.newaxis  # utils.py:5
DELTA = 1.1      # environment/RL_api.py:15


# ----------------------------------------------------------------------------- world objects
class EnvObject:  # environment/environment.py:5-18
    def __init__(self, environment: Optional["Environment"]):
        self.environment = environment
        if self.environment is not None:
            self.environment.add_object(self)

    def visualize_copy(self, newenv):
        return None  # nothing to draw (environment.py:11-12 copies an empty EnvObject)

    def update(self):
        pass

    def update_step(self):
        return 0


class Environment:  # environment/environment.py:21-47
    def __init__(self, w, h, max_time):
        self.w, self.h, self.max_time = w, h, max_time
        self.objects: List[EnvObject] = []
        self._backend = None

    def add_object(self, obj):
        self.objects.append(obj)

    def detach_object(self, obj):
        if obj in self.objects:
            self.objects.remove(obj)

    @property
    def timestep(self):
        if self._backend is None:
            return 1
        # every env of a batch steps in lockstep: the library mirrors the counter on the host (no device round trip)
        ts = self._backend.query(cm.Q_TIMESTEP) + getattr(self, "_host_ts_ahead", 0)
        n = self._backend.cfg.n_envs
        return ts if n == 1 else np.full((n,), ts, dtype=np.int32)

    def update(self, wall_jitter=None):
        """Environment.update (environment.py:42-47).  The world objects this module defines (Walls, Food,
        CircleObstacles, Pheromone, Ants, Anthill, RLApi) live on the device: their updates are ONE antsrl_update
        call, in the reference's order.  `wall_jitter` (float64 [E,N]) injects the np.random.random draws of
        Walls.update (walls.py:28); default is the library's counter-based generator.

        Objects the CALLER added (any other `EnvObject` with an `update()`) are host objects: they are called like
        the reference calls them — stable sort on `update_step()` (environment.py:43-47), BETWEEN the world's objects where
        their step puts them: the device update is then run one reference step at a time (antsrl_update_phase: Walls -1,
        Rocks / Pheromone 0, Ants 999, Anthill 1000) and a host object with step s runs before Walls (s < -1), after Walls
        (-1 <= s < 0), after the rocks and the pheromone update (0 <= s < 999), after the ants' (999 <= s < 1000) or after
        the anthill's (s >= 1000) — an object appended after the generator's keeps its place behind the world's objects
        of equal step, as list.sort(key=update_step) leaves it.  A host object that reads state through the views sees
        what the reference's would: the phases before it applied, the ones after it not yet.  Without such objects the
        update is ONE call (and may be deferred into the next step's launch)."""
        host = [(o.update_step(), i, o) for i, o in enumerate(self.objects) if not _device_updates(o)]
        host.sort(key=lambda t: (t[0], t[1]))  # stable in insertion order, like list.sort(key=update_step)
        b = self._backend

        def run(lo, hi):
            for step, _, o in host:
                if lo <= step < hi:
                    o.update()
        # (environment.py:45 increments timestep BEFORE any object's update: a host object called while the device update
        #  is outstanding sees the new value; the device counter advances with the last phase)
        self._host_ts_ahead = 1
        try:
            run(float("-inf"), -1)
            if any(-1 <= step < 1000 for step, _, _ in host):
                b.update_phase(cm.PHASE_WALLS, wall_jitter)
                run(-1, 0)
                b.update_phase(cm.PHASE_ROCKS_PHEROMONE)
                run(0, 999)
                b.update_phase(cm.PHASE_ANTS)
                run(999, 1000)
                b.update_phase(cm.PHASE_ANTHILL)
            else:
                b.update(wall_jitter)
        finally:
            self._host_ts_ahead = 0
        run(1000, float("inf"))

    def save_state(self):
        """Environment.save_state (environment.py:36-40): a host snapshot for the visualiser, in the
        reference's pickled layout (antsrl_amd.snapshot; write a list of them with snapshot.dump).
        Each object appears once (the reference's copies register themselves AND are added again by
        save_state, environment.py:8,39 — the viewer draws such an object twice to the same effect)."""
        from . import snapshot as S
        # a FRESH Environment like the reference's (its own timestep restarts at 1, environment.py:36-40); the
        # simulation's step rides along as sim_timestep
        snap = S.Environment(self.w, self.h, self.max_time, 1)
        snap.sim_timestep = int(np.asarray(self.timestep).reshape(-1)[0])
        for obj in self.objects:
            obj.visualize_copy(snap)  # the snapshot classes register themselves with `snap`
        return snap


def _device_updates(obj) -> bool:
    """True when the backend's kernels do this object's update: a device-backed view whose class has NOT overridden
    update().  A caller's subclass of a view with its own update() (say a Walls that also moves) is a host object as far
    as Environment.update is concerned — its update() is called."""
    return getattr(obj, "_device_backed", False) and type(obj).update is EnvObject.update


class _View(EnvObject):
    """Base of the device-backed object views."""
    _device_backed = True  # Environment.update: updated by the backend's kernels, not through update()

    def __init__(self, environment, env_index=0):
        super().__init__(environment)
        self._e = env_index

    @property
    def _b(self):
        return self.environment._backend

    def _read(self, which):
        return self._b.read_state(which).cpu().numpy()


class Pheromone(_View):  # environment/pheromone.py:20-45
    def __init__(self, environment, index, color=(64, 64, 64), max_val=None):
        super().__init__(environment)
        self.index, self.color, self.max_val = index, color, max_val
        self.w, self.h = environment.w, environment.h

    @property
    def phero(self):
        p = self._read(cm.S_PHERO_C0 + self.index)  # this object's channel only
        return p[0] if p.shape[0] == 1 else p

    def visualize_copy(self, newenv):
        from .snapshot import PheromoneVisualization
        return PheromoneVisualization(newenv, color=self.color, max_val=self.max_val,
                                      phero=np.asarray(self.phero).astype(np.uint8))  # pheromone.py:17


class Food(_View):  # environment/food.py:12-18
    @property
    def qte(self):
        q = self._read(cm.S_FOOD)
        return q[0] if q.shape[0] == 1 else q

    def visualize_copy(self, newenv):
        from .snapshot import FoodVisualization
        return FoodVisualization(newenv, qte=np.asarray(self.qte).astype(np.uint8))  # food.py:10


class Walls(_View):  # environment/walls.py:9-30
    def __init__(self, environment):
        super().__init__(environment)
        self.w, self.h = environment.w, environment.h
        self._map = None

    @property
    def map(self):
        if self._map is None:
            m = self._read(cm.S_WALLS).astype(bool)
            self._map = m[0] if m.shape[0] == 1 else m
        return self._map

    def update_step(self):
        return -1

    def visualize_copy(self, newenv):  # walls.py:16-17 keeps the object itself: same attributes
        from .snapshot import Walls as WallsSnapshot
        return WallsSnapshot(newenv, w=self.w, h=self.h, map=np.array(self.map))


class Anthill(_View):  # environment/anthill.py:16-46
    def __init__(self, environment, xyr):
        super().__init__(environment)
        self.w, self.h = environment.w, environment.h
        self._xyr = np.asarray(xyr)
        self._area = None

    @property
    def x(self):
        return int(self._xyr[0, 0]) if len(self._xyr) == 1 else self._xyr[:, 0]

    @property
    def y(self):
        return int(self._xyr[0, 1]) if len(self._xyr) == 1 else self._xyr[:, 1]

    @property
    def radius(self):
        return int(self._xyr[0, 2]) if len(self._xyr) == 1 else self._xyr[:, 2]

    @property
    def area(self):
        if self._area is None:
            a = self._read(cm.S_ANTHILL_AREA).astype(bool)
            self._area = a[0] if a.shape[0] == 1 else a
        return self._area

    @property
    def food(self):
        f = self._read(cm.S_ANTHILL_FOOD)
        return float(f[0]) if f.size == 1 else f

    def update_step(self):
        return 1000

    def visualize_copy(self, newenv):
        from .snapshot import AnthillVisualization
        return AnthillVisualization(newenv, x=self.x, y=self.y, radius=self.radius, food=self.food)


class CircleObstacles(_View):  # environment/circle_obstacles.py:15-61
    def __init__(self, environment, radiuses, weights):
        super().__init__(environment)
        self.w, self.h = environment.w, environment.h
        self.radiuses, self.weights = radiuses, weights
        self.n_obst = radiuses.shape[-1]

    @property
    def centers(self):
        c = self._read(cm.S_ROCK_CENTERS)
        return c[0] if c.shape[0] == 1 else c

    def visualize_copy(self, newenv):
        from .snapshot import CircleObstaclesVisualization
        return CircleObstaclesVisualization(newenv, centers=np.array(self.centers), radiuses=np.array(self.radiuses),
                                            weights=np.array(self.weights))


class Ants(_View):  # environment/ants.py:17-144
    def __init__(self, environment, n_ants_per_env, n_envs, max_hold):
        super().__init__(environment)
        self._n, self._E = n_ants_per_env, n_envs
        self.n_ants = n_ants_per_env * n_envs  # the env axis is folded into the ant axis
        self.max_hold = max_hold
        self.pheromones: List[Pheromone] = []

    def _flat(self, a):
        return a.reshape((self.n_ants,) + a.shape[2:])

    @property
    def ants(self):
        return self._flat(self._read(cm.S_ANTS_XYT))

    @property
    def prev_ants(self):
        return self._flat(self._read(cm.S_PREV_XY))

    @property
    def x(self):
        return self.ants[:, 0]

    @property
    def y(self):
        return self.ants[:, 1]

    @property
    def xy(self):
        return self.ants[:, 0:2]

    @property
    def theta(self):
        return self.ants[:, 2]

    @property
    def holding(self):
        return self._flat(self._read(cm.S_HOLDING))

    @property
    def mandibles(self):
        return self._flat(self._read(cm.S_MANDIBLES)).astype(bool)

    @property
    def seed(self):
        return self._flat(self._read(cm.S_SEED))

    @property
    def reward_state(self):
        return self._flat(self._read(cm.S_REWARD_STATE))

    @property
    def phero_activation(self):
        return self._flat(self._read(cm.S_ACTIVATION))

    def register_pheromone(self, pheromone):  # ants.py:82-84
        self.pheromones.append(pheromone)

    def activate_all_pheromones(self, new_activations):
        """ants.py:86-87.  The reference's activation matrix is bool until this call replaces it
        with a float array; from then on activate_pheromone's 256 deposits 256.0 instead of
        True == 1.0.  A float argument therefore also switches the deposit
        strength to 256."""
        a = np.asarray(new_activations)
        strength = 0.0 if a.dtype == np.bool_ else 256.0
        self._b.set_activation(a.astype(np.float32).reshape(self._E, self._n, -1), strength)

    def update_step(self):
        return 999

    def visualize_copy(self, newenv):
        from .snapshot import AntsVisualization
        return AntsVisualization(newenv, ants=self.ants, mandibles=self.mandibles, holding=self.holding,
                                 reward_state=self.reward_state)


# ----------------------------------------------------------------------------- rewards
class Reward:  # environment/rewards/reward.py:6-45
    kind = cm.REWARD_NONE

    def __init__(self):
        self.ants = None
        self.environment = None
        self.rewards = None

    def setup(self, ants):
        self.ants = ants
        self.environment = ants.environment
        self.rewards = np.zeros(ants.n_ants, dtype=float)

    def weights(self):
        return {}

    def visualization(self):
        return None


class _ExploredMixin:
    @property
    def explored_map(self):
        m = self.environment._backend.read_state(cm.S_EXPLORED).cpu().numpy().astype(bool)
        return m[0] if m.shape[0] == 1 else m

    def visualization(self):
        return self.explored_map.copy()


class ExplorationReward(_ExploredMixin, Reward):  # reward_custom.py:8-25
    kind = cm.REWARD_EXPLORATION


class Food_Reward(Reward):  # reward_custom.py:28-40
    kind = cm.REWARD_FOOD


class All_Rewards(_ExploredMixin, Reward):  # reward_custom.py:43-109
    kind = cm.REWARD_ALL

    def __init__(self, fct_explore=1, fct_food=1, fct_anthill=5, fct_explore_holding=0, fct_headinganthill=1):
        super().__init__()
        self.fct_explore, self.fct_food, self.fct_anthill = fct_explore, fct_food, fct_anthill
        self.fct_explore_holding, self.fct_headinganthill = fct_explore_holding, fct_headinganthill

    def weights(self):
        return dict(fct_explore=float(self.fct_explore), fct_food=float(self.fct_food),
                    fct_anthill=float(self.fct_anthill), fct_explore_holding=float(self.fct_explore_holding),
                    fct_headinganthill=float(self.fct_headinganthill))


# ----------------------------------------------------------------------------- RLApi
_KIND_OF = {Ants: cm.CH_ANTS, Pheromone: cm.CH_PHERO, Anthill: cm.CH_ANTHILL, Walls: cm.CH_WALLS,
            Food: cm.CH_FOOD, CircleObstacles: cm.CH_ROCKS}


class RLApi(EnvObject):  # environment/RL_api.py:22-204
    _device_backed = True  # (RLApi.update is the base class's no-op, RL_api.py:22,62)
    def __init__(self, reward: Reward, reward_threshold: float, max_speed: float, max_rot_speed: float,
                 carry_speed_reduction: float, backward_speed_reduction: float, as_numpy: bool = True):
        super().__init__(None)
        self.reward = reward
        self.reward_threshold = reward_threshold
        self.ants: Optional[Ants] = None
        self.original_ants_position = None
        self.perception_radius = 0
        self.perception_mask = None
        self.perceived_objects: List[EnvObject] = []
        self.perception_coords = None
        self.perception_fwd_delta = 0
        self.max_speed, self.max_rot_speed = max_speed, max_rot_speed
        self.carry_speed_reduction = carry_speed_reduction
        self.backward_speed_reduction = backward_speed_reduction
        self.save_perceptive_field = False  # (RL_api.py:49; main.py:51 sets it for the viewer) -> `perceptive_field` after
        self.perceptive_field = None        # every observation: bool [w, h] ([E, w, h] for a batch), RL_api.py:144-153
        self.as_numpy = as_numpy
        self._pending = None  # (base cfg kwargs, init arrays) from the generator
        self._backend = None

    # -- wiring ----------------------------------------------------------------
    def register_ants(self, new_ants: Ants):  # RL_api.py:57-66
        if self.environment is not None:
            self.environment.detach_object(self)
        self.ants = new_ants
        self.environment = new_ants.environment
        self.environment.add_object(self)
        self.perceived_objects = []
        self.reward.setup(self.ants)

    def setup_perception(self, radius: int, objects: List[EnvObject], mask=None, forward_delta=0):
        """RL_api.py:80-93.  (Re)creates the device batch for this perception set-up and loads the
        generator's initial state into it, so it must precede the first observation()/step()."""
        self.perception_radius = radius
        self.perception_mask = mask
        self.perceived_objects = objects
        self.perception_fwd_delta = forward_delta
        rng = np.arange(-radius, radius + 1)
        self.perception_coords = np.dstack([rng[AX, :].repeat(2 * radius + 1, 0),
                                            rng[:, AX].repeat(2 * radius + 1, 1)]).astype(float) * DELTA
        if self._pending is not None:
            self._materialize()

    def _channels(self):
        ch = []
        for obj in self.perceived_objects:
            kind = next((k for cls, k in _KIND_OF.items() if isinstance(obj, cls)), None)
            if kind is None:
                raise TypeError("cannot perceive %r" % (obj,))
            ch.append((kind, obj.index if kind == cm.CH_PHERO else 0))
        return ch

    def _materialize(self):
        from .batched import BatchedAntsEnv
        kw, init = self._pending
        cfg = cm.make_cfg(mask=self.perception_mask, perception_radius=self.perception_radius,
                          fwd_delta=float(self.perception_fwd_delta), delta=DELTA, channels=self._channels(),
                          max_speed=float(self.max_speed), max_rot_speed=float(self.max_rot_speed),
                          carry_speed_reduction=float(self.carry_speed_reduction),
                          backward_speed_reduction=float(self.backward_speed_reduction),
                          reward_kind=self.reward.kind, reward_threshold=float(self.reward_threshold),
                          **self.reward.weights(), **kw)
        self._backend = BatchedAntsEnv(cfg)
        if isinstance(init, tuple):  # ("device", AntsGen, episode_seed[, walls bitmaps]): generated on the GPU
            self._backend.generate(init[1], init[2], walls=init[3] if len(init) > 3 else None)
        else:
            self._backend.reset(init)
        self.environment._backend = self._backend

    # -- outputs ---------------------------------------------------------------
    def _out(self, t, fold=True):
        if fold:
            t = t.reshape((t.shape[0] * t.shape[1],) + tuple(t.shape[2:]))
        return t.cpu().numpy() if self.as_numpy else t

    def _acts(self, a):
        if a is None:
            return None
        c = self._backend.cfg
        if hasattr(a, "detach"):
            return a.reshape(c.n_envs, c.n_ants)
        return np.asarray(a).reshape(c.n_envs, c.n_ants)

    def observation(self):
        """RL_api.py:96-165 -> (perception [n,P,P,K], agent_state [n,2], state [n,2+C])."""
        b = self._backend
        obs, ast, rew = b.observe()
        self._field()
        self.reward.rewards = self._out(rew)
        state = self._state()
        return self._out(obs), self._out(ast), state

    def _field(self):  # RL_api.py:144-153
        if self.save_perceptive_field:
            f = self._backend.perceptive_field()
            f = f.cpu().numpy() if self.as_numpy else f
            self.perceptive_field = f[0] if self._backend.cfg.n_envs == 1 else f

    def _state(self):  # RL_api.py:155-158
        import torch
        b = self._backend
        m = b.read_state(cm.S_MANDIBLES).to(torch.float32)
        h = b.read_state(cm.S_HOLDING)
        a = (b.read_state(cm.S_ACTIVATION) > 0).to(torch.float32)
        return self._out(torch.cat([m[..., None], h[..., None], a], dim=-1))

    def step(self, rotation, on_off_pheromones):
        """RL_api.py:168-204 -> (perception, agent_state, reward, done)."""
        b = self._backend
        obs, ast, rew, done = b.step(self._acts(rotation), self._acts(on_off_pheromones))
        self._field()
        if self.as_numpy:  # the four outputs in one device-to-host copy
            obs, ast, rew, done = b.outputs_to_host()
            fold = lambda t: t.reshape((t.shape[0] * t.shape[1],) + t.shape[2:])
            self.reward.rewards = r = fold(rew)
            d = bool(done[0]) if b.cfg.n_envs == 1 else done.astype(bool)
            return fold(obs), fold(ast), r, d
        r = self._out(rew)
        self.reward.rewards = r
        if b.cfg.n_envs == 1:
            d = bool(done.cpu().numpy()[0])
        else:
            d = done.cpu().numpy().astype(bool) if self.as_numpy else done
        return self._out(obs), self._out(ast), r, d

    def visualize_copy(self, newenv):
        from .snapshot import RLVisualization
        return RLVisualization(newenv, heatmap=self.reward.visualization())
