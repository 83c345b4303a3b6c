#!/usr/bin/env python3
"""Does PHYSICALLY CONTIGUOUS device memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) for the workspace and the output
tensors give the fast state on a box whose ordinary allocations are in the slow one?  (follow-up to box_state_probe*.py)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from antsrl_amd import _lib, config as cm  # noqa: E402
from antsrl_amd import batched  # noqa: E402
from antsrl_amd.synth import synth_init  # noqa: E402

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
hip = _lib.hip_runtime()
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
held = []


class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def contiguous_u8(n):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), n, 0x4)  # hipDeviceMallocContiguous
    if rc != 0:
        raise RuntimeError("hipExtMallocWithFlags(contiguous, %d bytes) failed: %d" % (n, rc))
    raw = _Raw(p.value, n)
    held.append(raw)
    return torch.as_tensor(raw, device=dev)


E, N = 1024, 512
GRID = [256, 256]


def make():
    c = cm.make_cfg(E, N, GRID[0], GRID[1], n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
    return c, synth_init(c, seed=1234)


cfg, init = make()
g = torch.Generator(device=dev)
g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
_empty, _zeros = torch.empty, torch.zeros


def trial(tag, contiguous):
    if contiguous:  # BatchedAntsEnv allocates its workspace with torch.empty and its outputs with torch.zeros (uint8)
        def big(alloc):
            def f(*a, **k):
                n = a[0] if isinstance(a[0], int) else (a[0][0] if len(a[0]) == 1 else 0)
                if k.get("dtype") == torch.uint8 and str(k.get("device", "")).startswith("cuda") and n > (1 << 20):
                    t = contiguous_u8(n)
                    if alloc is _zeros:
                        t.zero_()
                    return t
                return alloc(*a, **k)
            return f
        batched.torch.empty, batched.torch.zeros = big(_empty), big(_zeros)
    try:
        env = batched.BatchedAntsEnv(cfg, dev)
    finally:
        batched.torch.empty, batched.torch.zeros = _empty, _zeros
    env.reset(init)
    for t in range(420):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(200):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    print("%-28s %dx%d ms/step %.4f  ws@%#x obs@%#x" % (tag, GRID[0], GRID[1], (time.perf_counter() - t0) / 200 * 1e3, env._ws_ptr, env.obs.data_ptr()), flush=True)
    del env
    torch.cuda.empty_cache()


for wh in ((256, 256), (256, 252), (256, 260), (254, 256), (256, 256)):
    GRID[:] = wh
    cfg, init = make()
    trial("ordinary allocations", False)
    trial("contiguous allocations", True)
