#!/usr/bin/env python3
"""Follow-up to placement_probe.py: ONE workspace, many observation buffers — what makes one fast?  The buffers are kept
alive (every allocation is new memory); per buffer: k_perceive over 30 steps at the same episode age (state restored by
reset + ageing is skipped: the episode just keeps running, the drift over the probe is reported by the repeated baseline).
    series A  exact-size buffers allocated back to back
    series B  a spacer of S MiB allocated in front of each buffer
    series C  the buffer as a slice of a larger allocation (+64 MiB, +1 GiB)
    series D  the env's own packed output buffer layout (small outputs in front, as BatchedAntsEnv allocates it)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import HipEvents

E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
NEV = cm.TIMING_EVENTS
STEPS = 30
evs = HipEvents(NEV * (STEPS // 5 + 1))
nobs = E * N * 343
env = BatchedAntsEnv(cfg, dev)
env.reset(synth_init(cfg, seed=1234))
own = env.obs
for t in range(400):
    env.step_update(rot[t % 8], ph[t % 8], None)


def run(tag):
    slots = []
    for t in range(STEPS):
        if t % 5 == 0:
            env.set_timing_events([evs.ev[NEV * len(slots) + i].value for i in range(NEV)])
            slots.append(len(slots))
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    kp = float(np.mean([evs.elapsed_ms(NEV * j + 2, NEV * j + 3) for j in slots]))
    ku = float(np.mean([evs.elapsed_ms(NEV * j + 1, NEV * j + 2) for j in slots]))
    print("%-44s obs VA %x  k_perceive %.4f  k_update_move %.4f" % (tag, env.obs.data_ptr(), kp, ku), flush=True)
    return kp


print("workspace VA %x" % env._ws_ptr)
run("own obs (inside the packed output buffer)")
keep = []
for i in range(8):
    b = torch.empty(nobs, dtype=torch.float32, device=dev); keep.append(b)
    env.obs = b.view(own.shape)
    run("A%d exact size, back to back" % i)
env.obs = own
run("own obs again")
for s in (2, 16, 64, 128, 256, 300, 512, 700, 1024, 2048):
    keep.append(torch.empty(s << 20, dtype=torch.uint8, device=dev))
    b = torch.empty(nobs, dtype=torch.float32, device=dev); keep.append(b)
    env.obs = b.view(own.shape)
    run("B spacer %4d MiB" % s)
for extra in (64 << 20, 1 << 30):
    for off in (0, extra // 4 // 2, extra // 4):
        b = torch.empty(nobs + extra // 4, dtype=torch.float32, device=dev); keep.append(b)
        env.obs = b[off:off + nobs].view(own.shape)
        run("C slice of +%4d MiB at element offset %d" % (extra >> 20, off))
for i in range(4):
    b = torch.empty(nobs + (6 << 20) // 4 + 64, dtype=torch.float32, device=dev); keep.append(b)
    o = (6 << 20) // 4 + 64
    env.obs = b[o:o + nobs].view(own.shape)
    run("D%d packed-output layout (6 MiB + 256 B in front)" % i)
env.obs = own
run("own obs again")
