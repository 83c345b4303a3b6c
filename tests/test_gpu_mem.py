"""antsrl_mem_alloc / antsrl_mem_free (include/antsrl.h) and antsrl_amd.vmm: device memory in physical pieces — usable by
kernels and torch alike, aligned, released when the last tensor over it dies, and what BatchedAntsEnv steps on by default."""
import ctypes as C
import gc

import pytest

pytestmark = pytest.mark.gpu


def test_pieced_memory_round_trip_and_lifetime():
    import torch
    from antsrl_amd import _lib, vmm
    lib = _lib.load()
    free0 = torch.cuda.mem_get_info()[0]
    n = (200 << 20) + 12345
    t = vmm.pieced_u8(n, "cuda:0")
    assert t.numel() == n and t.dtype == torch.uint8 and t.data_ptr() % (2 << 20) == 0
    v = t[: 64 << 20].view(torch.float32)
    v.copy_(torch.arange(v.numel(), device="cuda:0", dtype=torch.float32))
    t[-1] = 7
    assert float(v[12345]) == 12345.0 and int(t[-1]) == 7 and float(v.sum(dtype=torch.float64)) == (v.numel() - 1) * v.numel() / 2
    w = t[5 << 20:]          # a view keeps the mapping alive ...
    del t, v
    gc.collect()
    assert int(w[-1]) == 7
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info()[0]
    del w                    # ... the last one releases it: the physical pieces go back to the device
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] - held >= (190 << 20), "the 200 MiB were not returned"
    del free0
    # the C-ABI directly: bad arguments, double free
    p = C.c_void_p()
    assert lib.antsrl_mem_alloc(0, 0, C.byref(p)) == -1
    assert lib.antsrl_mem_alloc(1 << 20, 0, C.byref(p)) == 0 and p.value % (2 << 20) == 0
    assert lib.antsrl_mem_free(p) == 0 and lib.antsrl_mem_free(p) == -1 and lib.antsrl_mem_free(None) == 0


def test_env_on_pieced_memory_equals_env_on_torch_memory():
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(128, 384, 256, 256, n_rocks=4, deposit_strength=256.0)   # outputs 67 MiB: pieced (past vmm.SMALL_BYTES)
    init = synth_init(cfg, seed=3)
    a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg, pieced_memory=False)
    assert a._out_flat.data_ptr() % (2 << 20) < 256 and a.obs.data_ptr() != b.obs.data_ptr()
    a.reset(init)
    b.reset(init)
    rot, ph = random_actions(cfg, 5, seed=2)
    for t in range(5):
        for x, y in zip(a.step_update(rot[t], ph[t], None), b.step_update(rot[t], ph[t], None)):
            assert torch.equal(x, y)
    for which in (cm.S_ANTS_XYT, cm.S_PHERO, cm.S_FOOD, cm.S_EXPLORED):
        assert torch.equal(a.read_state(which), b.read_state(which))


def test_freed_ranges_never_alias_live_buffers():
    """antsrl_mem_free retires the virtual range (antsrl_mem.hip): on ROCm 7.2 a re-used range could still be translated to
    its OLD physical pieces, so that two live buffers aliased each other — seen in 95 of 300 rounds of exactly this loop
    before the fix (profiles/r04/vmm_stress.py).  Allocate two buffers, fill, cross-copy, check, free; sizes vary."""
    import torch
    from antsrl_amd import vmm
    seen = set()
    for it in range(60):
        n = ((it % 7) * 37 + 70) << 20
        v = vmm.pieced_u8(n, "cuda:0").view(torch.int32)
        assert v.data_ptr() not in seen, "a virtual range was handed out twice"
        seen.add(v.data_ptr())
        v.fill_(it + 1)
        other = torch.empty(n // 4, dtype=torch.int32, device="cuda:0").fill_(-(it + 1))
        u = vmm.pieced_u8(n, "cuda:0").view(torch.int32)
        u.copy_(v)
        u += 1000000
        assert bool((v == it + 1).all()) and bool((u == it + 1 + 1000000).all()) and bool((other == -(it + 1)).all()), it
        del v, u, other
        if it % 3 == 0:
            gc.collect()


def test_tune_placement_leaves_a_clean_handle():
    """BatchedAntsEnv.tune_placement() steps scratch episodes on the four {torch, pieced} x {torch, pieced} (workspace, outputs)
    pairs and keeps the fastest (possibly a re-created handle on another workspace); the handle is then loaded as usual —
    results equal those of an env that never tuned, whatever pair won."""
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(128, 384, 256, 256, n_rocks=4, deposit_strength=256.0)
    init = synth_init(cfg, seed=3)
    a, b = BatchedAntsEnv(cfg), BatchedAntsEnv(cfg)
    times = a.tune_placement(age=20, steps=8)
    assert times is not None and len(times) == 8 and a.placement_trials["chosen"] in range(8)  # (four pairs + four more draws)
    assert len(BatchedAntsEnv(cfg).tune_placement(age=10, steps=4, extra_outputs=0)) == 4
    assert BatchedAntsEnv(cm.make_cfg(2, 8, 32, 32)).tune_placement() is None  # (small batches: nothing to alias)
    a.reset(init)
    b.reset(init)
    rot, ph = random_actions(cfg, 5, seed=2)
    for t in range(5):
        for x, y in zip(a.step_update(rot[t], ph[t], None), b.step_update(rot[t], ph[t], None)):
            assert torch.equal(x, y)
    ha, hb = a.outputs_to_host(), b.outputs_to_host()
    for x, y in zip(ha, hb):
        assert (x == y).all()
    for which in (cm.S_ANTS_XYT, cm.S_PHERO, cm.S_FOOD, cm.S_EXPLORED, cm.S_TIMESTEP):
        assert torch.equal(a.read_state(which), b.read_state(which))
