#!/usr/bin/env python3
"""VERDICT r4 item 6, measured before anything is built: what would k_perceive gain if an environment's ants were taken in
SPATIAL order (neighbouring ants in one wave / workgroup share cell-record lines)?

Upper bound, through the public API only: an episode is aged 400 steps (ants spread over the grid), its state is read
back, and the SAME handle on the SAME buffers (placement constant) is reset twice from that state — once with the ants in
their own order, once with every environment's ants sorted along a Morton curve over their cells (4 x 4-cell blocks) —
and stepped 30 times each, alternating, k_perceive timed by the library's HIP events.  The sorted arm keeps the dense row
order (an ant's row lies at its index), i.e. it is BETTER than a permutation inside the kernel could be (that one
scatters the rows).  If the sorted arm is not clearly faster the lead is closed.

    python profiles/r05/spatial_order_probe.py [c2 c3 c4 c5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init


def morton(x, y):
    def spread(v):
        v = v.astype(np.uint32)
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def run(name):
    W_ = dict(bench.CONFIGS[name])
    E, N, W, H, R = W_["E"], W_["N"], W_["W"], W_["H"], W_["R"]
    extra = dict(n_rocks=R, deposit_strength=256.0, max_time=1 << 30)
    if W_["radius3"]:
        ax = np.arange(-3, 4)
        g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
        extra["filt"] = g / g.sum() * (1 - 0.001)
    cfg = cm.make_cfg(E, N, W, H, **extra)
    dev = torch.device("cuda", 0)
    bf16 = name == "c5"
    env = BatchedAntsEnv(cfg, dev, obs_dtype=torch.bfloat16 if bf16 else torch.float32)
    env.tune_placement()
    init = synth_init(cfg, seed=1234)
    env.reset(init)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
    for t in range(400):
        env.step_update(rot[t % 8], ph[t % 8], None)
    xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()
    aged = dict(init)
    aged["food"] = env.read_state(cm.S_FOOD).cpu().numpy()
    aged["phero"] = env.read_state(cm.S_PHERO).cpu().numpy()
    if R:
        rc = env.read_state(cm.S_ROCK_CENTERS).cpu().numpy()
        rocks = init["rocks"].copy()
        rocks[:, :, :2] = rc
        aged["rocks"] = rocks
    cx, cy = np.floor(xyt[..., 0]).astype(np.int64), np.floor(xyt[..., 1]).astype(np.int64)
    key = morton(cx >> 2, cy >> 2)
    order = np.argsort(key, axis=1, kind="stable")
    arms = {"own order": xyt, "sorted (Morton over 4x4-cell blocks)": np.take_along_axis(xyt, order[..., None], axis=1)}
    seeds = {"own order": init["seed"], "sorted (Morton over 4x4-cell blocks)": np.take_along_axis(init["seed"], order, axis=1)}
    NEV = cm.TIMING_EVENTS
    res = {k: [] for k in arms}
    for rep in range(3):
        for k in arms:
            st = dict(aged, ants_xyt=arms[k], seed=seeds[k])
            env.reset(st)
            for t in range(6):  # the first steps mark the (fresh) explored map: not timed
                env.step_update(rot[t % 8], ph[t % 8], None)
            evs = bench.HipEvents(NEV * 24)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(24):
                env.set_timing_events([evs.ev[NEV * t + i].value for i in range(NEV)])
                env.step_update(rot[t % 8], ph[t % 8], None)
            e1.record()
            e1.synchronize()
            prc = float(np.mean([evs.elapsed_ms(NEV * t + 2, NEV * t + 3) for t in range(24)]))
            um = float(np.mean([evs.elapsed_ms(NEV * t + 1, NEV * t + 2) for t in range(24)]))
            evs.destroy()
            res[k].append((prc, um))
    print("%s (%d envs x %d ants, %dx%d%s):" % (name, E, N, W, H, ", bf16 rows" if bf16 else ""))
    for k, v in res.items():
        print("   %-40s k_perceive %s ms   (kernel in front of it: %s)" % (k, " ".join("%.4f" % a for a, _ in v), " ".join("%.4f" % b for _, b in v)))
    sys.stdout.flush()
    del env


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["c2", "c3", "c4", "c5"]):
        run(c)
