"""Device replay memory against a straightforward host ring buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rolling_extend_and_access():
    import torch
    from antsrl_amd.replay import DeviceReplayMemory
    rng = np.random.default_rng(0)
    L, obs_sp, ag_sp = 50, (7, 7, 6), (2,)
    mem = DeviceReplayMemory(L, obs_sp, ag_sp, (2,))
    ring = {k: [None] * L for k in "s a act r ns na d".split()}
    head = fill = 0
    for it, n in enumerate([12, 12, 12, 20, 7, 64, 3]):  # 4th batch wraps, 6th exceeds max_len
        s, ns = rng.random((n,) + obs_sp, np.float32), rng.random((n,) + obs_sp, np.float32)
        a, na = rng.random((n,) + ag_sp, np.float32), rng.random((n,) + ag_sp, np.float32)
        rot, ph = rng.integers(-1, 2, n), (rng.integers(0, 3, n) if it % 2 == 0 else None)
        r, done = rng.random(n, np.float32), bool(it % 3 == 0)
        if it % 2:  # tensors already on the device
            mem.extend(torch.from_numpy(s).cuda(), torch.from_numpy(a).cuda(), (torch.from_numpy(rot).cuda(), ph),
                       torch.from_numpy(r).cuda(), torch.from_numpy(ns).cuda(), torch.from_numpy(na).cuda(), done)
        else:
            mem.extend(s, a, (rot, ph), r, ns, na, done)
        for j in range(n):
            ring["s"][head], ring["a"][head], ring["r"][head] = s[j], a[j], r[j]
            ring["ns"][head], ring["na"][head], ring["d"][head] = ns[j], na[j], done
            ring["act"][head] = (rot[j], 1 if ph is None else ph[j])
            head = (head + 1) % L
            fill = min(L, fill + 1)
        assert len(mem) == fill and mem.head == head
        got = mem[list(range(fill))]
        for k, t in zip("s a act r ns na d".split(), got):
            want = np.array([ring[k][i] for i in range(fill)])
            np.testing.assert_array_equal(t.cpu().numpy(), want.astype(t.cpu().numpy().dtype))
    batch = mem.random_access(16)
    assert batch[0].shape == (16, 7, 7, 6) and batch[2].shape == (16, 2) and batch[0].is_cuda


def test_feeds_from_the_batched_env_without_leaving_the_device():
    import torch
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.policy import LinearPolicy
    from antsrl_amd.replay import DeviceReplayMemory
    from antsrl_amd.synth import synth_init
    cfg = cm.make_cfg(3, 40, 64, 64, deposit_strength=256.0, max_time=4)
    env = BatchedAntsEnv(cfg)
    env.reset(synth_init(cfg, seed=2, n_food_discs=5, food_rmin=3, food_rmax=6))
    pol = LinearPolicy(49 * 6, env.device)
    mem = DeviceReplayMemory(1000, (7, 7, 6), (2,), (2,))
    obs, ast, _ = env.observe()
    obs, ast = obs.clone(), ast.clone()
    for t in range(5):  # main.py:92-131 with every array staying on the GPU
        rot, ph = pol.act(obs, ast)
        nobs, nast, rew, done = env.step_update(rot, ph)
        mem.extend(obs, ast, (rot, ph), rew, nobs, nast, done.repeat_interleave(cfg.n_ants))
        obs, ast = nobs.clone(), nast.clone()
    assert len(mem) == 5 * 3 * 40
    s, a, act, r, ns, na, d = mem.random_access(32)
    assert s.is_cuda and act.dtype == torch.int64 and int(act[:, 0].min()) >= -1
    assert int(mem.dones.sum()) == 3 * 40  # exactly the step at timestep == max_time
