"""Builds libantsrl_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m antsrl_amd.build          # or antsrl_amd.build.build_hip()

-ffp-contract=off: the reference (numpy) rounds every product before adding; fused
multiply-add would change ant coordinates in the last bit.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libantsrl_hip.so")
#: the same sources with -DANTSRL_PROFILING: A/B selectors, ablation flags and the k_act phase trace exist in this
#: library only (profiles/*.sh, tests/alt_paths.sh load it through ANTSRL_LIB); the product library has none
PROF_LIB_PATH = os.path.join(LIB_DIR, "libantsrl_hip_prof.so")
SOURCES = ["antsrl_act.hip", "antsrl_perceive.hip", "antsrl_update.hip", "antsrl_sweep.hip", "antsrl_state.hip",
           "antsrl_capi.hip", "antsrl_policy.hip", "antsrl_mem.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("antsrl_device.h", "antsrl_util.h", "antsrl_update_env.h",
                                           "antsrl_update_one.h", "antsrl_flush.h", "antsrl_layout.h")] + [
    os.path.join(HERE, "..", "include", "antsrl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needs the ROCm toolchain)")


#: host-only translation units (no kernel in them): a change there moves no byte on the device
HOST_ONLY_SOURCES = ("antsrl_capi.hip", "antsrl_mem.hip")


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the KERNEL sources (every translation unit with device code + the shared headers +
    the C-ABI header): what a measurement was taken ON (profiles/traffic_<config>.json records it; bench.py flags PMC
    figures that pre-date the kernels it runs)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES if s not in HOST_ONLY_SOURCES] + [os.path.normpath(x) for x in HEADERS]):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def is_stale(path: str = LIB_PATH) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(out: str, extra, verbose: bool) -> None:
    """One hipcc process per source (they are independent translation units), then one link."""
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(os.path.dirname(out), "obj_" + os.path.splitext(os.path.basename(out))[0])  # (object files never travel: .gpurunignore)
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in FLAGS if f != "-shared"] + list(extra)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), obj, cmd))
    objs = []
    for pr, obj, cmd in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        objs.append(obj)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    _compile(LIB_PATH, [], verbose)
    return LIB_PATH


def build_prof(force: bool = False, verbose: bool = False) -> str:
    """libantsrl_hip_prof.so: the profiling build (A/B selectors, ablations, phase trace)."""
    if not force and not is_stale(PROF_LIB_PATH):
        return PROF_LIB_PATH
    _compile(PROF_LIB_PATH, ["-DANTSRL_PROFILING"], verbose)
    return PROF_LIB_PATH


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """lib/variants/<name>.so: the profiling build plus extra -D switches (compile-time ablations and A/B
    variants for profiles/*.sh; loaded through ANTSRL_LIB)."""
    out = os.path.join(LIB_DIR, "variants", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    _compile(out, ["-DANTSRL_PROFILING"] + list(defines), verbose)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m antsrl_amd.build --variant NAME -DFOO -DBAR=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")], verbose=True))
        sys.exit(0)
    print(build_hip(force="--force" in sys.argv, verbose=True))
    if "--prof" in sys.argv:
        print(build_prof(force="--force" in sys.argv, verbose=True))
