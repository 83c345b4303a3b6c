"""Optional module aliases so code written against the reference's import paths
(`from environment.RL_api import RLApi`, `from environment.pheromone import Pheromone`,
`from generator.environment_generator import EnvironmentGenerator`, ...) picks up the
MI355X-backed classes without editing that code.  See INTEGRATION.md."""
import sys
import types

from . import generator as _gen
from . import rl_api as _api

_MAP = {
    "environment.RL_api": (_api, ["RLApi", "DELTA"]),
    "environment.environment": (_api, ["Environment", "EnvObject"]),
    "environment.ants": (_api, ["Ants"]),
    "environment.pheromone": (_api, ["Pheromone"]),
    "environment.food": (_api, ["Food"]),
    "environment.walls": (_api, ["Walls"]),
    "environment.anthill": (_api, ["Anthill"]),
    "environment.circle_obstacles": (_api, ["CircleObstacles"]),
    "environment.rewards.reward": (_api, ["Reward"]),
    "environment.rewards.reward_custom": (_api, ["ExplorationReward", "Food_Reward", "All_Rewards"]),
    "generator.environment_generator": (_gen, ["EnvironmentGenerator", "PHERO_COLORS"]),
    "generator.map_generators": (_gen, ["CirclesGenerator", "PerlinGenerator"]),
}


def install_reference_aliases(force: bool = False):
    """Registers `environment.*` / `generator.*` modules that re-export antsrl_amd classes.
    Refuses to shadow already-imported modules of those names unless force=True."""
    for name in ["environment", "environment.rewards", "generator"] + list(_MAP):
        if name in sys.modules and not force and not getattr(sys.modules[name], "__antsrl_alias__", False):
            raise RuntimeError("module %r is already imported; pass force=True to shadow it" % name)
    for pkg in ("environment", "environment.rewards", "generator"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        m.__antsrl_alias__ = True
        sys.modules[pkg] = m
    for name, (src, attrs) in _MAP.items():
        m = types.ModuleType(name)
        m.__antsrl_alias__ = True
        for a in attrs:
            setattr(m, a, getattr(src, a))
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        setattr(sys.modules[parent], leaf, m)
