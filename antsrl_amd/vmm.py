"""torch tensors over antsrl_mem_alloc (include/antsrl.h): device memory in physical pieces of at most 16 MiB.

On MI355X the observation kernel runs 15 % slower when the observation tensor (a streaming write) and the workspace's cell
records (scattered gathers) lie in the same ZONE of the device's memory (profiles/r05/two_colour.txt).  In a
fresh process hipMalloc (what torch.empty ends in) and the HIP virtual-memory allocator draw from different zones: the
observation tensor on antsrl_mem_alloc pieces + the workspace on torch.empty is a fast pair (k_perceive 0.167 against 0.197 ms
at c3), two torch.empty buffers the slow one; BatchedAntsEnv.tune_placement measures the device at hand.
`pieced_u8(nbytes, device)` is a uint8 tensor over such memory.  PyTorch stays plumbing: the tensor wraps the pointer through
`__cuda_array_interface__`."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


SMALL_BYTES = 64 << 20  # buffers below this come from torch.empty


class _Holder:
    def __init__(self, ptr, nbytes, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner  # the tensor references this object: the mapping lives as long as any view of it


class _Mapping:
    """One antsrl_mem_alloc range; handed back to the library's pool when the last tensor over it has gone (the pool keeps
    it mapped and gives it to the next request of the same size: antsrl_mem.hip; `trim()` returns pooled memory to the device)."""

    def __init__(self, nbytes: int, dev: torch.device):
        self.lib = _lib.load()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            torch.cuda.current_stream(dev)  # (the device's context exists)
            p = C.c_void_p()
            _lib.check(self.lib.antsrl_mem_alloc(int(nbytes), idx, C.byref(p)), "mem_alloc")
        self.ptr = p.value

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p and C is not None and torch is not None:  # (at interpreter shutdown the modules may be gone: the process's memory goes with it)
            try:  # (antsrl_mem_free waits for the BLOCK's device — nothing enqueued may still touch it — and parks it in the pool)
                self.lib.antsrl_mem_free(C.c_void_p(p))
            except Exception:
                pass


def pieced_u8(nbytes: int, device) -> torch.Tensor:
    """A uint8 device tensor of `nbytes` over antsrl_mem_alloc memory.  The tensor's storage references the mapping (through
    the `__cuda_array_interface__` holder): no cycle, the range is released when the last view dies."""
    dev = torch.device(device)
    m = _Mapping(nbytes, dev)
    t = torch.as_tensor(_Holder(m.ptr, int(nbytes), m), device=dev)
    assert t.data_ptr() == m.ptr, "the tensor must alias the mapping, not copy it"
    return t


def empty_u8(nbytes: int, device) -> torch.Tensor:
    """pieced_u8, falling back to torch.empty (with a warning) only where the HIP runtime lacks the virtual-memory API: an
    allocation strategy, not a compute path."""
    if nbytes < SMALL_BYTES:  # (nothing to alias with: a single piece would do, and hipMalloc's sub-allocation is cheaper)
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    try:
        return pieced_u8(nbytes, device)
    except _lib.AntsrlError as e:
        import warnings
        warnings.warn("antsrl_mem_alloc failed (%s): falling back to torch.empty — physically contiguous memory can cost the "
                      "observation kernel 15 %% on MI355X" % e)
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def trim() -> None:
    """Returns every pooled (freed) block's physical memory to the device (antsrl_mem_trim)."""
    _lib.check(_lib.load().antsrl_mem_trim(), "mem_trim")
