"""ctypes binding of libantsrl_hip.so (include/antsrl.h).  Fails loudly: there is no CPU
fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

from .config import AntsCfg, AntsGen, AntsInit

LIB_PATH = os.environ.get("ANTSRL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                        "libantsrl_hip.so")  # ANTSRL_LIB: A/B builds

#: every symbol include/antsrl.h declares
EXPORTS = ("antsrl_abi_version", "antsrl_cfg_size", "antsrl_last_error", "antsrl_workspace_bytes", "antsrl_create",
           "antsrl_destroy", "antsrl_reset", "antsrl_generate", "antsrl_step", "antsrl_observe", "antsrl_update", "antsrl_flush",
           "antsrl_step_update", "antsrl_set_timing_events", "antsrl_set_activation", "antsrl_policy_mlp", "antsrl_read_state", "antsrl_state_bytes",
           "antsrl_set_obs_format", "antsrl_query", "antsrl_bench_copy", "antsrl_set_inloop_policy", "antsrl_set_obs_row_stride", "antsrl_mem_alloc", "antsrl_mem_free",
           "antsrl_mem_trim", "antsrl_mem_stats", "antsrl_update_phase",
           "antsrl_perceptive_field")

_lib = None


class AntsrlError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AntsrlError(
            "libantsrl_hip.so is missing (%s): build it with `python -m antsrl_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm).
    # A process must hold ONE HIP runtime: import torch first so the NEEDED entries of
    # libantsrl_hip.so bind to the runtime torch's streams and tensors live in.  (Loaded the
    # other way round, launches fail with "no ROCm-capable device is detected".)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    lib.antsrl_abi_version.restype = i32
    lib.antsrl_last_error.restype = C.c_char_p
    lib.antsrl_workspace_bytes.argtypes = [C.POINTER(AntsCfg), C.POINTER(C.c_size_t)]
    lib.antsrl_create.argtypes = [C.POINTER(AntsCfg), vp, C.c_size_t, C.POINTER(vp)]
    lib.antsrl_destroy.argtypes = [vp]
    lib.antsrl_destroy.restype = None
    lib.antsrl_reset.argtypes = [vp, C.POINTER(AntsInit), vp]
    lib.antsrl_generate.argtypes = [vp, C.POINTER(AntsGen), C.c_uint64, vp]
    lib.antsrl_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.antsrl_observe.argtypes = [vp, vp, vp, vp, vp]
    lib.antsrl_update.argtypes = [vp, vp, vp]
    lib.antsrl_flush.argtypes = [vp, vp]
    lib.antsrl_update_phase.argtypes = [vp, i32, vp, vp]
    lib.antsrl_perceptive_field.argtypes = [vp, vp, vp]
    lib.antsrl_step_update.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.antsrl_set_timing_events.argtypes = [vp, C.POINTER(vp)]
    lib.antsrl_set_activation.argtypes = [vp, vp, C.c_double, vp]
    lib.antsrl_policy_mlp.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.antsrl_read_state.argtypes = [vp, i32, vp, vp]
    lib.antsrl_state_bytes.argtypes = [vp, i32, C.POINTER(C.c_size_t)]
    lib.antsrl_query.argtypes = [vp, i32, C.POINTER(C.c_longlong)]
    lib.antsrl_set_inloop_policy.argtypes = [vp, i32] + [vp] * 9
    lib.antsrl_bench_copy.argtypes = [vp, vp, C.c_size_t, vp]
    lib.antsrl_set_obs_row_stride.argtypes = [vp, i32]
    lib.antsrl_mem_alloc.argtypes = [C.c_size_t, i32, C.POINTER(vp)]
    lib.antsrl_mem_free.argtypes = [vp]
    lib.antsrl_mem_trim.argtypes = []
    lib.antsrl_mem_stats.argtypes = [C.POINTER(C.c_size_t)] * 4
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the build lost a symbol
    lib.antsrl_cfg_size.restype = C.c_size_t
    from .config import ABI_VERSION
    if lib.antsrl_abi_version() != ABI_VERSION:
        raise AntsrlError("ABI version mismatch: library %d, binding %d" % (lib.antsrl_abi_version(), ABI_VERSION))
    if lib.antsrl_cfg_size() != C.sizeof(AntsCfg):
        raise AntsrlError("AntsCfg layout mismatch: library %d bytes, binding %d" % (
            lib.antsrl_cfg_size(), C.sizeof(AntsCfg)))
    _lib = lib
    return lib


def hip_runtime() -> C.CDLL:
    """The libamdhip64 this process already uses (torch's), for raw hipEvent_t handling."""
    load()
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if "libamdhip64" in path:
            return C.CDLL(path)
    raise AntsrlError("no libamdhip64 mapped in this process")


def check(rc: int, what: str = "antsrl") -> None:
    if rc != 0:
        msg = load().antsrl_last_error().decode("utf-8", "replace")
        raise AntsrlError("%s failed (%d): %s" % (what, rc, msg))


class HipEvents:
    """Raw hipEvent_t handles (the ABI hook records them on the launch stream)."""

    def __init__(self, n):
        self.hip = hip_runtime()  # the runtime torch already loaded, not a second copy
        self.ev = []
        for _ in range(n):
            e = C.c_void_p()
            rc = self.hip.hipEventCreate(C.byref(e))
            assert rc == 0, "hipEventCreate failed: %d" % rc
            self.ev.append(e)

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        rc = self.hip.hipEventElapsedTime(C.byref(ms), self.ev[a], self.ev[b])
        assert rc == 0, "hipEventElapsedTime failed: %d" % rc
        return ms.value

    def destroy(self):
        for e in self.ev:
            self.hip.hipEventDestroy(e)
