"""Bounds of the observation copy-out (antsrl_amd/csrc/antsrl_flush.h), checked on the host with the SAME
code the kernels run: tests/native/flush_plan_check.cpp includes the header, g++ compiles it, and every lane
of every (row length, one/two rows, misalignment, line phase) combination is enumerated.  Invariant: a wave
only ever stores inside its own rows, and stores all of them.  (the regression guard for the
round-1 abort / run-to-run difference on the c3 batch.)"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_flush_plan_stays_inside_its_rows_and_covers_them(tmp_path):
    so = str(tmp_path / "flush_plan_check.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
                           os.path.join(HERE, "native", "flush_plan_check.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.flush_plan_violations.restype = C.c_long
    assert lib.flush_plan_violations(1) == 0
