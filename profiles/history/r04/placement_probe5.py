#!/usr/bin/env python3
"""Fresh process per arm, any bench configuration: which of the two big buffers comes from pieced memory.
    python profiles/r04/placement_probe5.py <config> none|obs|ws|both [piece MiB for the workspace, default 16]
(obs pieced = the product's default, antsrl_mem_alloc's 16 MiB pieces; the workspace's pieces come from the probe's ctypes
allocator so that their size can be varied.)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import CONFIGS, HipEvents
from vmm_ctypes import ShuffledBuffer

name, mode = sys.argv[1], sys.argv[2]
piece = (int(sys.argv[3]) if len(sys.argv) > 3 else 16) << 20
W_ = CONFIGS[name]
E, N = W_["E"], W_["N"]
extra = dict(n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
if W_["radius3"]:
    ax = np.arange(-3, 4)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    extra["filt"] = g / g.sum() * (1 - 0.001)
cfg = cm.make_cfg(E, N, W_["W"], W_["H"], **extra)
dev = torch.device("cuda", 0)
if mode in ("ws", "both"):
    need = C.c_size_t()
    _lib.check(_lib.load().antsrl_workspace_bytes(C.byref(cfg), C.byref(need)), "workspace_bytes")
    wsbuf = ShuffledBuffer(need.value + 256, dev, shuffle=False, chunk_bytes=piece)
    _orig = torch.empty

    def _empty(*a, **k):  # (the first torch.empty of the constructor is the workspace)
        torch.empty = _orig
        return wsbuf.tensor
    torch.empty = _empty
env = BatchedAntsEnv(cfg, dev, pieced_memory=mode in ("obs", "both"))
env.reset(synth_init(cfg, seed=1234))
gen = torch.Generator(device=dev); gen.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
for t in range(300):
    env.step_update(rot[t % 8], ph[t % 8], None)
NEV = cm.TIMING_EVENTS
STEPS = 40
evs = HipEvents(NEV * (STEPS // 5 + 1))
slots = []
for t in range(STEPS):
    if t % 5 == 0:
        env.set_timing_events([evs.ev[NEV * len(slots) + i].value for i in range(NEV)])
        slots.append(len(slots))
    env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
ms = np.array([[evs.elapsed_ms(NEV * j + i, NEV * j + i + 1) for i in range(NEV - 1)] for j in slots]).mean(axis=0)
print("%s %-5s ws piece %3d MiB: sweep %.4f  move/update_move %.4f  perceive %.4f  update %.4f  sum %.4f" % (
    name, mode, piece >> 20, ms[0], ms[1], ms[2], ms[3], ms.sum()))
