#!/usr/bin/env python3
"""Fresh process per arm: the observation tensor from antsrl_amd.vmm.ShuffledBuffer (2 MiB physical pieces in shuffled /
sequential order) against the env's own allocation.
    python profiles/r04/placement_probe4.py own|product|vmm_seq|vmm_shuf|vmm_ws|vmm_both   (own: torch.empty memory; product: BatchedAntsEnv's default, antsrl_mem_alloc)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vmm_ctypes import ShuffledBuffer
from bench import HipEvents

mode = sys.argv[1]
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0)
nobs = E * N * 343
CH = int(os.environ.get("CHUNK_MB", "2")) << 20
wsbuf = None
if mode in ("vmm_ws", "vmm_both"):  # the workspace from 2 MiB pieces as well: BatchedAntsEnv allocates it with torch.empty
    import ctypes as C
    from antsrl_amd import _lib
    need = C.c_size_t()
    _lib.check(_lib.load().antsrl_workspace_bytes(C.byref(cfg), C.byref(need)), "workspace_bytes")
    wsbuf = ShuffledBuffer(need.value + 256, dev, shuffle=False, chunk_bytes=CH)
    _orig_empty = torch.empty

    def _empty(*a, **k):  # (the first torch.empty of the constructor is the workspace)
        torch.empty = _orig_empty
        return wsbuf.tensor
    torch.empty = _empty
env = BatchedAntsEnv(cfg, dev, pieced_memory=(mode == 'product'))
buf = None
if mode in ("vmm_seq", "vmm_shuf", "vmm_both"):
    buf = ShuffledBuffer(nobs * 4, dev, seed=int(os.environ.get("SEED", "0")), shuffle="shuf" in mode, chunk_bytes=CH)
    env.obs = buf.tensor.view(torch.float32).view(env.obs.shape)
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
chk = float(env.obs[E // 2, 7].sum())
print("%-10s chunk %4s MiB  ws VA %x obs VA %x  k_perceive %.4f  k_update_move %.4f  (checksum %.3f)" % (mode, os.environ.get("CHUNK_MB", "2"), env._ws_ptr, env.obs.data_ptr(), kp, ku, chk))
