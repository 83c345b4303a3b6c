#!/usr/bin/env python3
"""Does the plain device copy (antsrl_bench_copy, 16 B per lane) care where its two buffers lie?  src / dst from torch.empty
(hipMalloc) or from antsrl_mem_alloc pieces, 1 GiB and 2 GiB each."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import _lib, vmm

lib = _lib.load()
dev = torch.device("cuda", 0)


def alloc(kind, n):
    return vmm.pieced_u8(n, dev) if kind == "pieced" else torch.empty(n, dtype=torch.uint8, device=dev)


def rate(src, dst, n, reps=20):
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        _lib.check(lib.antsrl_bench_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, st), "copy")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.antsrl_bench_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, st), "copy")
    e1.record(); e1.synchronize()
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


for gib in (1, 2):
    n = gib << 30
    for ks, kd in (("torch", "torch"), ("torch", "pieced"), ("pieced", "torch"), ("pieced", "pieced"), ("torch", "torch")):
        s, d = alloc(ks, n), alloc(kd, n)
        s.view(torch.int32).fill_(7)
        print("%d GiB  src %-6s dst %-6s  %.0f GB/s (read + written)" % (gib, ks, kd, rate(s, d, n)), flush=True)
        del s, d
    same = torch.empty(2 * n, dtype=torch.uint8, device=dev)
    print("%d GiB  both halves of ONE torch allocation  %.0f GB/s" % (gib, rate(same[:n], same[n:], n)), flush=True)
    del same
