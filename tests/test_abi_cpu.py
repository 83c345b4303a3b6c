"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol
include/antsrl.h declares, its AntsCfg layout matches the ctypes mirror, and its host-only
entry points validate their arguments.  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

from antsrl_amd import _lib
from antsrl_amd import build as buildmod
from antsrl_amd.config import AntsCfg, make_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    buildmod.build_hip()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "antsrl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(antsrl_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), "libantsrl_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names


def test_layout_and_version(lib):
    assert lib.antsrl_abi_version() == 5
    assert lib.antsrl_cfg_size() == C.sizeof(AntsCfg)


def test_workspace_bytes_and_validation(lib):
    cfg = make_cfg(1024, 512, 256, 256, n_rocks=8)
    n = C.c_size_t()
    assert lib.antsrl_workspace_bytes(C.byref(cfg), C.byref(n)) == 0
    # scaled pheromone units, two channels: ONE array of 16-byte {p0, p1, food, pad} cell records
    # (1 GiB) + bitmaps + ant SoA
    cells = 1024 * 256 * 256
    assert 16 * cells < n.value < 16 * cells + 0.1 * 2 ** 30
    from antsrl_amd.config import PHERO_EXPLICIT_SWEEP
    cfg2 = make_cfg(1024, 512, 256, 256, n_rocks=8, phero_mode=PHERO_EXPLICIT_SWEEP)
    n2 = C.c_size_t()
    assert lib.antsrl_workspace_bytes(C.byref(cfg2), C.byref(n2)) == 0
    # explicit sweep: two [cell][2] float32 pheromone buffers (ping-pong) + a separate array of {food, meta}
    # records (the cell-meta path: wall / anthill / presence bits and the explored stamp beside the food value)
    assert n2.value - n.value == cells * (2 * 8 + 8) - cells * 16
    bad = cfg.copy()
    bad.n_phero = 9
    assert lib.antsrl_workspace_bytes(C.byref(bad), C.byref(n)) == -1
    assert b"n_phero" in lib.antsrl_last_error()
    big = make_cfg(70000, 4, 16, 16)
    assert lib.antsrl_workspace_bytes(C.byref(big), C.byref(n)) == -1
    assert b"65535" in lib.antsrl_last_error()
    bad = cfg.copy()
    bad.abi_version = 7
    assert lib.antsrl_workspace_bytes(C.byref(bad), C.byref(n)) == -1


def test_env_identity_fields_are_validated(lib):
    """AntsCfg.env_id_base / n_envs_total (ABI 5): a negative base, a base past 31 bits and a total smaller than the
    shard's own end are refused by the host-side validation."""
    n = C.c_size_t()
    ok = make_cfg(8, 16, 32, 32, env_id_base=1016, n_envs_total=1024)
    assert lib.antsrl_workspace_bytes(C.byref(ok), C.byref(n)) == 0
    for base, total, msg in ((-1, 0, b"env_id_base"), (0x7fffffff - 3, 0, b"env_id_base"), (1016, 1023, b"n_envs_total")):
        bad = make_cfg(8, 16, 32, 32, env_id_base=base, n_envs_total=total)
        assert lib.antsrl_workspace_bytes(C.byref(bad), C.byref(n)) == -1
        assert msg in lib.antsrl_last_error()


def test_create_rejects_bad_workspace(lib):
    cfg = make_cfg(2, 8, 16, 16)
    h = C.c_void_p()
    assert lib.antsrl_create(C.byref(cfg), None, 0, C.byref(h)) == -1
    # fake (never dereferenced on the host) aligned pointer, but too small
    assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096), 16, C.byref(h)) == -2
    assert b"too small" in lib.antsrl_last_error()
    n = C.c_size_t()
    lib.antsrl_workspace_bytes(C.byref(cfg), C.byref(n))
    assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096 + 8), n.value, C.byref(h)) == -1  # misaligned
    assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096), n.value, C.byref(h)) == 0
    # not reset yet: every state-touching call refuses, nothing is launched
    assert lib.antsrl_update(h, None, None) == -1
    assert b"antsrl_reset" in lib.antsrl_last_error()
    lib.antsrl_destroy(h)


def test_lds_budget_is_checked(lib):
    """Grids of any size are accepted: the reference's perception shapes take the cell-meta path (no per-env
    bit map in LDS at all); other channel lists run k_act, whose per-env bit maps move from LDS to HBM scratch
    past ~600k cells (the workspace then includes them).  An ant count whose per-ant state cannot live in
    160 KiB of LDS is refused at create time."""
    from antsrl_amd.config import CH_ANTS, CH_FOOD, CH_PHERO, CH_WALLS
    n = C.c_size_t()
    n_small = C.c_size_t()
    cfg = make_cfg(1, 64, 2048, 2048)
    assert lib.antsrl_workspace_bytes(C.byref(cfg), C.byref(n)) == 0
    h = C.c_void_p()
    assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096), n.value, C.byref(h)) == 0
    v = C.c_longlong()
    assert lib.antsrl_query(h, 0, C.byref(v)) == 0 and v.value == 1  # ANTSRL_Q_CELL_META
    lib.antsrl_destroy(h)
    words = 2048 * 2048 // 32
    assert 16 * 2048 * 2048 + 3 * 4 * words <= n.value < 16 * 2048 * 2048 + 4 * 4 * words
    # a channel list the cell-meta path does not take: k_act with its maps in HBM scratch at this size
    odd = [(CH_FOOD, 0), (CH_ANTS, 0), (CH_PHERO, 0), (CH_PHERO, 1), (CH_WALLS, 0)]
    cfg_o = make_cfg(1, 64, 2048, 2048, channels=odd)
    assert lib.antsrl_workspace_bytes(C.byref(cfg_o), C.byref(n)) == 0
    assert lib.antsrl_create(C.byref(cfg_o), C.c_void_p(4096), n.value, C.byref(h)) == 0
    assert lib.antsrl_query(h, 0, C.byref(v)) == 0 and v.value == 0
    lib.antsrl_destroy(h)
    assert n.value >= 16 * 2048 * 2048 + 5 * 4 * words
    cfg_s = make_cfg(1, 64, 512, 512, channels=odd)  # bit maps in LDS: no scratch maps in the workspace
    assert lib.antsrl_workspace_bytes(C.byref(cfg_s), C.byref(n_small)) == 0
    assert n_small.value < 16 * 512 * 512 + 4 * 4 * (512 * 512 // 32) + 65536
    cfg = make_cfg(1, 6000, 64, 64)
    lib.antsrl_workspace_bytes(C.byref(cfg), C.byref(n))
    assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096), n.value, C.byref(h)) == -4
    assert b"LDS" in lib.antsrl_last_error()


def test_reference_seed_limit_is_checked_without_wrapping(lib):
    """np.random.seed(seed * 5) takes 32 bits (environment_generator.py:55): (episode_seed + env_id_base + n_envs) * 5 must
    stay below 2^32.  A base past 0xFFFFFFFF / 5 used to make the limit's subtraction wrap: the call was accepted
    and the seeds of the last environments replayed earlier episodes' streams.  Refused before anything is launched."""
    from antsrl_amd.config import make_gen
    n = C.c_size_t()
    for base, seed in ((900000000, 1), (858993459 - 8, 1), (0, 858993452), (0, 2 ** 33), (0x7fffffff - 8, 0)):
        cfg = make_cfg(8, 16, 32, 32, env_id_base=base)
        assert lib.antsrl_workspace_bytes(C.byref(cfg), C.byref(n)) == 0
        h = C.c_void_p()
        assert lib.antsrl_create(C.byref(cfg), C.c_void_p(4096), n.value, C.byref(h)) == 0
        gen = make_gen(wall_density=0.0, rng="reference")
        rc = lib.antsrl_generate(h, C.byref(gen), C.c_uint64(seed), None)  # (refused: nothing touches the fake workspace)
        assert rc == -1 and b"np.random.seed" in lib.antsrl_last_error(), (base, seed, lib.antsrl_last_error())
        lib.antsrl_destroy(h)
    # (seeds right below the limit are accepted: tests/test_gpu_generate.py runs them on the device)


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "antsrl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "liboracle" not in src, f


def test_batched_env_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from antsrl_amd.batched import BatchedAntsEnv
    with pytest.raises(_lib.AntsrlError):
        BatchedAntsEnv(make_cfg(1, 4, 16, 16))


def test_placement_tuner_stopping_rule():
    """BatchedAntsEnv.tune_placement walks further into the device's memory until its trials have shown both placement levels
    (profiles/r05/two_colour.txt): the rule on recorded trial sets."""
    from antsrl_amd.batched import placement_levels_seen
    fresh = [0.20761, 0.23375, 0.22291, 0.23091, 0.2008, 0.20152, 0.20208, 0.20301]      # a fresh process on a fast device
    one_level = [0.23279, 0.23461, 0.23397, 0.23541, 0.23309, 0.23433, 0.23434, 0.23468]  # device 0xb6d3a6e8...: every trial slow
    assert placement_levels_seen(fresh)
    assert not placement_levels_seen(one_level)
    assert not placement_levels_seen(one_level + [0.32247])          # an outlier is not a level
    assert placement_levels_seen(one_level + [0.32247, 0.2012])      # ... a buffer of another zone is
    assert not placement_levels_seen([0.0358, 0.0352, 0.0357, 0.0351])  # c2: no zone effect at all
