"""antsrl_set_obs_row_stride (opt-in): observation rows a whole number of 128-byte lines apart.  The values are those of
the dense tensor, bit for bit, in the same [E][N][P][P][K] positions of a strided view; the padding is zeros; nothing
outside the padded buffer is written; the dense layout stays the default and the in-loop policy refuses the stride."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CANARY = -12345.5


@pytest.mark.parametrize("E,N,W,H,rocks,bf16,explicit", [
    (1024, 509, 256, 256, 8, False, False),   # c3's shape with an odd ant count: partial runs, one-row tails (K = 7: 343 -> 352)
    (40, 130, 64, 64, 0, False, False),       # K = 6: 294 -> 320 elements
    (24, 511, 128, 128, 3, True, False),      # bfloat16, K = 7: 343 -> 384 elements
    (9, 77, 64, 48, 0, True, True),           # bfloat16, K = 6 (294 -> 320), explicit-sweep records (two gathers)
    (3, 1500, 128, 128, 2, False, False),     # more than 1024 ants per env
])
def test_padded_rows_equal_the_dense_tensor(E, N, W, H, rocks, bf16, explicit):
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init
    cfg = cm.make_cfg(E, N, W, H, n_rocks=rocks, deposit_strength=256.0,
                      phero_mode=cm.PHERO_EXPLICIT_SWEEP if explicit else cm.PHERO_AUTO)
    init = synth_init(cfg, seed=6, n_food_discs=6, food_rmin=2, food_rmax=5)
    dt = torch.bfloat16 if bf16 else torch.float32
    dense, padded = BatchedAntsEnv(cfg, obs_dtype=dt), BatchedAntsEnv(cfg, obs_dtype=dt, obs_row_stride="line")
    row = cfg.pside ** 2 * cfg.n_channels
    pitch = padded.obs_row_pitch
    assert pitch > row and (pitch * (2 if bf16 else 4)) % 128 == 0 and padded.obs_padded.shape == (E, N, pitch)
    assert padded.obs.shape == dense.obs.shape and padded.obs.data_ptr() == padded.obs_padded.data_ptr()
    # guard bands around the padded buffer, canaries inside it
    band = 4096
    buf = torch.full((E * N * pitch + 2 * band,), CANARY, dtype=dt, device=padded.device)
    off = (-(buf.data_ptr() + band * buf.element_size()) % 128) // buf.element_size()  # (the buffer must start on a line)
    padded.obs_padded = buf[band + off:band + off + E * N * pitch].view(E, N, pitch)
    padded.obs = padded.obs_padded[..., :row].unflatten(-1, tuple(dense.obs.shape[2:]))
    front, back = buf[:band + off], buf[band + off + E * N * pitch:]
    dense.reset(init)
    padded.reset(init)
    rot, ph = random_actions(cfg, 4, seed=5)
    for t in range(4):
        od = dense.step_update(rot[t], ph[t], None)
        op = padded.step_update(rot[t], ph[t], None)
        for name, a, b in zip(("obs", "agent_state", "reward", "done"), od, op):
            assert torch.equal(a, b), "step %d: %s" % (t, name)
        assert bool((padded.obs_padded[..., row:] == 0).all()), "the padding is zeros"
        assert bool((front == CANARY).all()) and bool((back == CANARY).all()), "store outside the padded buffer"
    od, op = dense.observe(), padded.observe()
    assert torch.equal(od[0], op[0])
    if not bf16:
        hd, hp = dense.outputs_to_host(), padded.outputs_to_host()  # (re-pointed obs: the per-tensor copy path)
        np.testing.assert_array_equal(hd[0], hp[0])


def test_row_stride_is_validated_and_excludes_the_inloop_policy():
    import ctypes as C
    import torch
    from antsrl_amd import _lib
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.policy import LinearPolicy
    from antsrl_amd.synth import synth_init
    cfg = cm.make_cfg(4, 64, 64, 64, n_rocks=2)
    env = BatchedAntsEnv(cfg)
    lib = env.lib
    for bad in (100, 343 + 1, 352 + 32, 10000):
        assert lib.antsrl_set_obs_row_stride(env._h, bad) == -1, bad
    assert lib.antsrl_set_obs_row_stride(env._h, 352) == 0 and lib.antsrl_set_obs_row_stride(env._h, 0) == 0
    assert lib.antsrl_set_obs_row_stride(env._h, 343) == 0
    kact = BatchedAntsEnv(cm.make_cfg(4, 64, 64, 64, n_rocks=2, act_path=cm.ACT_SINGLE_KERNEL))
    assert lib.antsrl_set_obs_row_stride(kact._h, 352) == -4  # cell-meta path only
    p16 = BatchedAntsEnv(cfg, obs_dtype=torch.bfloat16, obs_row_stride="line")
    if p16.query(cm.Q_PERCEIVE_RUN) * 4 > 32:
        pytest.skip("the in-loop policy needs a 32-ant tile per workgroup (tests/alt_paths.sh: ANTSRL_PRC_RUN)")
    p16.reset(synth_init(cfg, seed=1, n_food_discs=3, food_rmin=2, food_rmax=4))
    # refused when it is CONFIGURED, in either order — not in the middle of a step whose move has already been enqueued
    pol = LinearPolicy(cfg.pside ** 2 * cfg.n_channels, p16.device, seed=1)
    with pytest.raises(_lib.AntsrlError, match="in-loop policy"):
        pol.attach(p16)
    p16.observe()  # (no policy was attached: the padded env still observes)
    d16 = BatchedAntsEnv(cfg, obs_dtype=torch.bfloat16)
    d16.reset(synth_init(cfg, seed=1, n_food_discs=3, food_rmin=2, food_rmax=4))
    pol.attach(d16)
    assert lib.antsrl_set_obs_row_stride(d16._h, 384) == -4 and b"in-loop policy" in lib.antsrl_last_error()
    d16.observe()
    pol.detach(d16)
    assert lib.antsrl_set_obs_row_stride(d16._h, 384) == 0
    del C
