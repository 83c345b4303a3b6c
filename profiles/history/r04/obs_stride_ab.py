#!/usr/bin/env python3
"""In-process A/B of the observation row stride: ONE handle, ONE workspace, ONE observation buffer — the
dense tensor is a prefix of the padded one — so the physical page placement that makes processes differ by +-10 %
(box_state_probe) is the same for both arms.  Blocks of steps alternate dense / padded; per block: wall ms per
step and k_perceive's mean duration by the library's HIP events.

    python profiles/r04/obs_stride_ab.py [--config c3|c2|c4] [--blocks 6] [--steps 100] [--bf16]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from antsrl_amd import _lib
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import CONFIGS, HipEvents

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--blocks", type=int, default=6)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--age", type=int, default=400)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--envs", type=int, default=0)
args = ap.parse_args()
W_ = CONFIGS[args.config]
E = args.envs or W_["E"]
extra = dict(n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
if W_["radius3"]:
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    extra["filt"] = g / g.sum() * (1 - 0.001)
cfg = cm.make_cfg(E, W_["N"], W_["W"], W_["H"], **extra)
dt = torch.bfloat16 if args.bf16 else torch.float32
env = BatchedAntsEnv(cfg, obs_dtype=dt, obs_row_stride="line")
env.reset(synth_init(cfg, seed=1234))
dev = env.device
row, pitch = cfg.pside ** 2 * cfg.n_channels, env.obs_row_pitch
padded_buf = env.obs_padded
dense_view = padded_buf.reshape(-1)[: E * cfg.n_ants * row].view(E, cfg.n_ants, cfg.pside, cfg.pside, cfg.n_channels)


def set_mode(padded):
    _lib.check(env.lib.antsrl_set_obs_row_stride(env._h, pitch if padded else 0), "stride")
    if padded:
        env.obs_padded = padded_buf
        env.obs = padded_buf[..., :row].unflatten(-1, (cfg.pside, cfg.pside, cfg.n_channels))
    else:
        env.obs_padded = None
        env.obs = dense_view


gen = torch.Generator(device=dev)
gen.manual_seed(99)
RING = 8
rot = torch.randint(-1, 2, (RING, E, cfg.n_ants), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (RING, E, cfg.n_ants), generator=gen, device=dev, dtype=torch.int8)
set_mode(False)
for t in range(args.age):
    env.step_update(rot[t % RING], ph[t % RING], None)
NEV = cm.TIMING_EVENTS
evs = HipEvents(NEV * (args.steps // 10 + 1))
res = {False: [], True: []}
print("# %s  E=%d N=%d row=%d pitch=%d %s  (one handle / workspace / buffer; blocks of %d steps)" % (
    args.config, E, cfg.n_ants, row, pitch, "bf16" if args.bf16 else "f32", args.steps))
for b in range(2 * args.blocks):
    padded = bool(b & 1)
    set_mode(padded)
    for t in range(10):
        env.step_update(rot[t % RING], ph[t % RING], None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    slots = []
    for t in range(args.steps):
        if t % 10 == 0:
            j = len(slots)
            env.set_timing_events([evs.ev[NEV * j + i].value for i in range(NEV)])
            slots.append(j)
        env.step_update(rot[t % RING], ph[t % RING], None)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    kp = float(np.mean([evs.elapsed_ms(NEV * j + 2, NEV * j + 3) for j in slots]))
    ku = float(np.mean([evs.elapsed_ms(NEV * j + 1, NEV * j + 2) for j in slots]))
    res[padded].append((ms, kp, ku))
    print("block %2d %-6s  %.4f ms/step   k_perceive %.4f   k_(update_)move %.4f" % (b, "padded" if padded else "dense", ms, kp, ku))
for padded in (False, True):
    a = np.array(res[padded])
    print("%-6s median: %.4f ms/step, k_perceive %.4f, k_(update_)move %.4f" % ("padded" if padded else "dense", *np.median(a, axis=0)))
