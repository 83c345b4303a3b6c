#!/usr/bin/env python3
"""Fresh process per arm: a dummy allocation of D GiB made (and kept) BEFORE the env is created — does shifting where the
workspace and the output buffer land in physical memory change the state of a box on which nothing reaches the fast level?
    python profiles/r04/placement_probe6.py <dummy GiB> [torch|pieced]     (the output buffer's kind; workspace torch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import HipEvents

D = float(sys.argv[1])
kind = sys.argv[2] if len(sys.argv) > 2 else "pieced"
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0)
dummy = torch.empty(int(D * (1 << 30)), dtype=torch.uint8, device=dev) if D > 0 else None
env = BatchedAntsEnv(cfg, dev, pieced_memory=(kind == "pieced"))
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
print("dummy %5.1f GiB  out %-6s  ws VA %x  k_perceive %.4f" % (D, kind, env._ws_ptr, kp))
