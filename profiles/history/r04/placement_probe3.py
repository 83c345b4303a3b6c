#!/usr/bin/env python3
"""Fresh process per arm: HOW the observation tensor is allocated (placement_probe2.py: every freshly allocated buffer was
~12 % faster than the env's own packed, zero-filled output buffer).
    python profiles/r04/placement_probe3.py own|empty_after|zeros_after|empty_before|zeros_before|empty_after_touch
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import HipEvents

mode = sys.argv[1]
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0)
nobs = E * N * 343
pre = None
if mode.endswith("before"):
    pre = (torch.zeros if mode.startswith("zeros") else torch.empty)(nobs, dtype=torch.float32, device=dev)
env = BatchedAntsEnv(cfg, dev)
if pre is not None:
    env.obs = pre.view(env.obs.shape)
elif mode.startswith("empty_after"):
    env.obs = torch.empty(nobs, dtype=torch.float32, device=dev).view(env.obs.shape)
    if mode.endswith("touch"):
        env.obs.zero_()
elif mode == "zeros_after":
    env.obs = torch.zeros(nobs, dtype=torch.float32, device=dev).view(env.obs.shape)
env.reset(synth_init(cfg, seed=1234))
gen = torch.Generator(device=dev); gen.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
for t in range(400):
    env.step_update(rot[t % 8], ph[t % 8], None)
NEV = cm.TIMING_EVENTS
STEPS = 60
evs = HipEvents(NEV * (STEPS // 5 + 1))
slots = []
for t in range(STEPS):
    if t % 5 == 0:
        env.set_timing_events([evs.ev[NEV * len(slots) + i].value for i in range(NEV)])
        slots.append(len(slots))
    env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
kp = float(np.mean([evs.elapsed_ms(NEV * j + 2, NEV * j + 3) for j in slots]))
ku = float(np.mean([evs.elapsed_ms(NEV * j + 1, NEV * j + 2) for j in slots]))
print("%-18s ws VA %x obs VA %x  k_perceive %.4f  k_update_move %.4f" % (mode, env._ws_ptr, env.obs.data_ptr(), kp, ku))
