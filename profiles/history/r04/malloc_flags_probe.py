#!/usr/bin/env python3
"""Does the CACHING ATTRIBUTE of the observation tensor's memory decide fast / slow?  The tensor from
hipExtMallocWithFlags(flag) — 0 default, 1 fine-grained, 3 uncached, 4 contiguous — against the env's own allocation; the
workspace optionally too.   python3 profiles/r04/malloc_flags_probe.py <obs flag | own> [ws flag]   (fresh process per arm)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from antsrl_amd import config as cm, _lib
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import HipEvents

class _Holder:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

def ext_alloc(nbytes, flag):
    hip = _lib.hip_runtime()
    hip.hipExtMallocWithFlags.restype = C.c_int
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag)
    if rc != 0: raise RuntimeError("hipExtMallocWithFlags(%d) -> %d" % (flag, rc))
    return torch.as_tensor(_Holder(p.value, nbytes), device="cuda:0")  # (leaks by design: a probe)

arm = sys.argv[1]
wsflag = sys.argv[2] if len(sys.argv) > 2 else None
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
if wsflag is not None:
    need = C.c_size_t()
    _lib.check(_lib.load().antsrl_workspace_bytes(C.byref(cfg), C.byref(need)), "workspace_bytes")
    wsbuf = ext_alloc(need.value + 256, int(wsflag))
    _orig = torch.empty
    def _empty(*a, **k):
        torch.empty = _orig
        return wsbuf
    torch.empty = _empty
env = BatchedAntsEnv(cfg, dev, pieced_memory=(arm == "product"))
if arm not in ("own", "product"):
    buf = ext_alloc(E * N * 343 * 4 + 256, int(arm))
    env.obs = buf[:E * N * 343 * 4].view(torch.float32).view(env.obs.shape)
env.reset(synth_init(cfg, seed=1234))
gen = torch.Generator(device=dev); gen.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
for t in range(400): env.step_update(rot[t % 8], ph[t % 8], None)
NEV = cm.TIMING_EVENTS; STEPS = 60
evs = HipEvents(NEV * (STEPS // 5 + 1)); slots = []
for t in range(STEPS):
    if t % 5 == 0:
        env.set_timing_events([evs.ev[NEV * len(slots) + i].value for i in range(NEV)]); slots.append(len(slots))
    env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
kp = float(np.mean([evs.elapsed_ms(NEV * j + 2, NEV * j + 3) for j in slots]))
ku = float(np.mean([evs.elapsed_ms(NEV * j + 1, NEV * j + 2) for j in slots]))
print("obs %-8s ws %-6s k_perceive %.4f  k_update_move %.4f  (checksum %.3f)" % (arm, wsflag, kp, ku, float(env.obs[E // 2, 7].sum())))
