"""Probe-only: a device buffer from hipMemCreate pieces of a CHOSEN size (ctypes on the HIP runtime torch loaded), mapped in
sequential or shuffled order.  The product's allocator is antsrl_mem_alloc (antsrl_amd/csrc/antsrl_mem.hip, fixed piece size)."""
import ctypes as C

import numpy as np
import torch

from antsrl_amd import _lib


class _Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class _AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class _Prop(C.Structure):  # hipMemAllocationProp
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", _Loc),
                ("win32HandleMetaData", C.c_void_p), ("allocFlags", _AllocFlags)]


class _Access(C.Structure):  # hipMemAccessDesc
    _fields_ = [("location", _Loc), ("flags", C.c_int)]


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


class _Holder:
    def __init__(self, ptr, nbytes, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


class ShuffledBuffer:
    def __init__(self, nbytes, device, seed=0, shuffle=True, chunk_bytes=2 << 20):
        self.hip = hip = _lib.hip_runtime()
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        for name in ("hipMemAddressReserve", "hipMemCreate", "hipMemMap", "hipMemSetAccess", "hipMemGetAllocationGranularity"):
            getattr(hip, name).restype = C.c_int
        hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(_Prop), C.c_ulonglong]
        hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Access), C.c_size_t]
        hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(_Prop), C.c_int]
        with torch.cuda.device(dev):
            torch.cuda.current_stream(dev)
            prop = _Prop(1, 0, _Loc(1, idx), None, _AllocFlags(0, 0, 0))
            gran = C.c_size_t()
            _ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")
            chunk = (max(int(chunk_bytes), int(gran.value)) + gran.value - 1) // gran.value * gran.value
            n = (nbytes + chunk - 1) // chunk
            self.ptr = C.c_void_p()
            _ck(hip.hipMemAddressReserve(C.byref(self.ptr), n * chunk, chunk, None, 0), "reserve")
            hs = []
            for _ in range(n):
                h = C.c_void_p()
                _ck(hip.hipMemCreate(C.byref(h), chunk, C.byref(prop), 0), "create")
                hs.append(h)
            order = np.random.default_rng(seed).permutation(n) if shuffle else np.arange(n)
            for i in range(n):
                _ck(hip.hipMemMap(C.c_void_p(self.ptr.value + i * chunk), chunk, 0, hs[int(order[i])], 0), "map")
            acc = _Access(_Loc(1, idx), 3)
            _ck(hip.hipMemSetAccess(self.ptr, n * chunk, C.byref(acc), 1), "access")
            self.tensor = torch.as_tensor(_Holder(self.ptr.value, n * chunk, self), device=dev)[:nbytes]  # (leaks by design: a probe)
