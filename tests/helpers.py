"""Shared test helpers: fixture loading and the parity comparators."""
import glob
import json
import os

import numpy as np

from antsrl_amd import config as cfgmod

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OP_STEP, OP_UPDATE, OP_OBSERVE = 0, 1, 2

_KIND = {"Ants": cfgmod.CH_ANTS, "Pheromone": cfgmod.CH_PHERO, "Anthill": cfgmod.CH_ANTHILL,
         "Walls": cfgmod.CH_WALLS, "Food": cfgmod.CH_FOOD, "CircleObstacles": cfgmod.CH_ROCKS}
_REWARD = {"none": cfgmod.REWARD_NONE, "exploration": cfgmod.REWARD_EXPLORATION,
           "food": cfgmod.REWARD_FOOD, "all": cfgmod.REWARD_ALL}


def fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_fixture(name, n_envs=1):
    """-> (cfg, init dict of numpy arrays with leading env axis, F = npz dict, meta)."""
    F = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.loads(str(F["meta_json"]))
    channels, pi = [], 0
    for k in meta["perceived"]:
        kind = _KIND[k]
        if kind == cfgmod.CH_PHERO:
            channels.append((kind, pi))
            pi += 1
        else:
            channels.append((kind, 0))
    wts = meta["weights"] or {}
    cfg = cfgmod.make_cfg(
        n_envs, meta["n_ants"], meta["w"], meta["h"], n_phero=meta["n_phero"], n_rocks=meta["n_rocks"],
        max_time=meta["max_time"], mask=F["mask"].astype(np.uint8), fwd_delta=meta["fwd_delta"],
        channels=channels, max_speed=meta["max_speed"], max_rot_speed=meta["max_rot_speed"],
        carry_speed_reduction=meta["carry"], backward_speed_reduction=meta["backward"],
        max_hold=meta["max_hold"], phero_max_val=meta["max_val"], deposit_strength=meta["deposit_strength"],
        filt=np.array(meta["filter"]), reward_kind=_REWARD[meta["reward"]],
        reward_threshold=meta["reward_threshold"], **{k: float(v) for k, v in wts.items()})

    def rep(a):
        return np.ascontiguousarray(np.broadcast_to(a[None], (n_envs,) + a.shape))

    init = dict(ants_xyt=rep(F["init_ants_xyt"]), seed=rep(F["init_seed"]),
                walls=rep(F["init_walls"].astype(np.uint8)), food=rep(F["init_food"].astype(np.float32)),
                anthill_xyr=rep(F["init_anthill_xyr"].astype(np.int32)),
                rocks=rep(F["init_rocks"]) if meta["n_rocks"] else None)
    F["explored"] = np.unpackbits(F["explored"], axis=-1)
    return cfg, init, F, meta


def phero_close(got, want, threshold=0.01, rtol=1e-5, atol=1e-5):
    """Pheromone comparator for the fp32 device grid vs the float64 reference:
    |got - want| <= atol + rtol*|want|, except that a cell whose reference value lies within
    rtol of the `< threshold -> 0` cut (pheromone.py:45) may sit on either side of it."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    ok = np.abs(got - want) <= atol + rtol * np.abs(want)
    band = (np.abs(want - threshold) <= 4 * rtol * threshold) | \
           ((want == 0) & (np.abs(got - threshold) <= 4 * rtol * threshold))
    ok |= band & ((got == 0) | (np.abs(got - threshold) <= 4 * rtol * threshold))
    return ok
