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


#: Pheromone tolerance of the fp32 device grid against the float64 reference.  The bar (north_star) is "within 1e-5 fp32";
#: the comparator is set to 4 x the largest error the kernels actually make, measured over the whole GPU suite with
#: ANTSRL_ERR_LOG (round 5, profiles/r05/parity_errors.md): largest RELATIVE error 7.7e-7 (the fp32 stencils: radius-3 and
#: 3 x 3 diffusion fixtures; 6.5e-7 on the benched c4 shard), 1.5e-7 with scaled units over a 2000-step c3 episode; no cell
#: outside the cut's band was non-zero where the reference is zero.  (Until round 4: 1e-5 + 1e-5 |ref| — 170 fp32 ulps at a
#: clipped cell.)
PHERO_RTOL = 3.2e-6
PHERO_ATOL = 1e-7
#: half-width (relative to the threshold) of the band around the `< threshold -> 0` cut in which a cell may sit on either side
CUT_BAND_RTOL = 4e-5

_ERR_LOG = os.environ.get("ANTSRL_ERR_LOG")  # a jsonl file: every comparison's largest errors (measurement, not a gate)


def _log_err(kind, **kw):
    if _ERR_LOG:
        kw.update(kind=kind, test=os.environ.get("PYTEST_CURRENT_TEST", ""))
        with open(_ERR_LOG, "a") as f:
            f.write(json.dumps(kw) + "\n")


def phero_close(got, want, threshold=0.01, rtol=None, atol=None):
    """Pheromone comparator for the fp32 device grid vs the float64 reference:
    |got - want| <= atol + rtol*|want|, except that a cell whose reference value lies within
    CUT_BAND_RTOL of the `< threshold -> 0` cut (pheromone.py:45) may sit on either side of it."""
    rtol = PHERO_RTOL if rtol is None else rtol
    atol = PHERO_ATOL if atol is None else atol
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    ok = np.abs(got - want) <= atol + rtol * np.abs(want)
    bw = CUT_BAND_RTOL * threshold
    band = (np.abs(want - threshold) <= bw) | ((want == 0) & (np.abs(got - threshold) <= bw))
    ok |= band & ((got == 0) | (np.abs(got - threshold) <= bw))
    if _ERR_LOG:
        d = np.abs(got - want)
        nz = (want != 0) & ~band
        _log_err("phero", cells=int(want.size), nonzero=int(nz.sum()), max_abs=float(d[~band].max()) if (~band).any() else 0.0,
                 max_rel=float((d[nz] / np.abs(want[nz])).max()) if nz.any() else 0.0,
                 max_abs_where_ref_zero=float(d[(want == 0) & ~band].max()) if ((want == 0) & ~band).any() else 0.0,
                 in_band_flipped=int((band & ((got == 0) != (want == 0))).sum()), max_ref=float(want.max()) if want.size else 0.0)
    return ok


def assert_xy_close(got, want, atol, err_msg=""):
    """Float64 coordinates (ants, previous positions, rock centres) against the reference: absolute tolerance `atol`;
    the largest difference seen goes to ANTSRL_ERR_LOG."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if _ERR_LOG and got.shape == want.shape and got.size:
        _log_err("xy", max_abs=float(np.abs(got - want).max()), atol=float(atol), n=int(got.size), ctx=str(err_msg)[:120])
    np.testing.assert_allclose(got, want, rtol=0, atol=atol, err_msg=err_msg)
