#!/usr/bin/env python3
"""Stress of antsrl_mem_alloc / antsrl_mem_free: allocate, fill, check, free, again (the virtual range is usually handed out
again): does a kernel ever see a stale mapping?  Also two live buffers written alternately, sizes that change."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import time
import torch
from antsrl_amd import vmm
WAIT = float(os.environ.get('WAIT_S', '0'))

dev = "cuda:0"
bad = 0
seen = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    n = ((it % 7) * 37 + 70) << 20
    t = vmm.pieced_u8(n, dev)
    seen[t.data_ptr()] = seen.get(t.data_ptr(), 0) + 1
    v = t.view(torch.int32)
    if WAIT:
        torch.cuda.synchronize(); time.sleep(WAIT)
    v.fill_(it + 1)
    other = torch.empty(n // 4, dtype=torch.int32, device=dev).fill_(-(it + 1))   # a torch allocation beside it
    u = vmm.pieced_u8(n, dev).view(torch.int32)
    if WAIT:
        torch.cuda.synchronize(); time.sleep(WAIT)
    u.copy_(v)
    u += 1000000
    bv, bu, bo = (v != it + 1), (u != it + 1 + 1000000), (other != -(it + 1))
    if bool(bv.any()) or bool(bu.any()) or bool(bo.any()):
        bad += 1
        def where(b):
            idx = b.nonzero().flatten()
            return "none" if idx.numel() == 0 else "%d elems, MiB %.1f..%.1f" % (idx.numel(), float(idx[0]) * 4 / 2**20, float(idx[-1]) * 4 / 2**20)
        print("iteration %d (%d MiB): v[%s] u[%s] other[%s]; wrong v value %s" % (it, n >> 20, where(bv), where(bu), where(bo),
              v[bv][:3].tolist() if bool(bv.any()) else ""), flush=True)
    del t, v, u, other
    if it % 3 == 0:
        gc.collect()
print("iterations done, mismatches: %d; distinct base addresses %d (re-used up to %d times)" % (bad, len(seen), max(seen.values())))
