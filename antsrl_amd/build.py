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
SOURCES = ["antsrl_act.hip", "antsrl_update.hip", "antsrl_sweep.hip", "antsrl_state.hip", "antsrl_capi.hip",
           "antsrl_policy.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("antsrl_device.h", "antsrl_util.h", "antsrl_update_env.h")] + [
    os.path.join(HERE, "..", "include", "antsrl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needs the ROCm toolchain)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
