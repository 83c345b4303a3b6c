"""N > 1 path on CPU: world_size-2 (and 3, ragged) gloo processes shard the env batch and
all-gather reward/done exactly as bench.py / a multi-GPU driver does over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from antsrl_amd.config import make_cfg
from antsrl_amd.dist import RewardGather, shard_range
from antsrl_amd.synth import synth_init


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(E, rank, world)
        # each rank draws ITS block of the global batch: same arrays as the single-process batch
        cfg = make_cfg(hi - lo, N, 32, 32)
        init = synth_init(cfg, seed=50, env_offset=lo, n_food_discs=3, food_rmin=2, food_rmax=4)
        g = RewardGather(E, N, "cpu")
        out = []
        for step in range(3):
            rew = torch.full((hi - lo, N), float(step)) + torch.arange(lo, hi, dtype=torch.float32)[:, None]
            done = torch.tensor([(e + step) % 2 for e in range(lo, hi)], dtype=torch.uint8)
            if step % 2 == 0:
                r, d = g(rew, done)  # blocking form
            else:  # overlapped form: the caller's buffers may be overwritten right after start()
                g.start(rew, done)
                rew.fill_(-1.0)
                done.fill_(9)
                r, d = g.finish()
            out.append((r.clone().numpy(), d.clone().numpy()))
        # zero-copy form: the "kernels" write into the send slots, alternating
        zc = []
        for step in range(3):
            rew, done = g.outputs(step % 2)
            rew.copy_(torch.full((hi - lo, N), 10.0 + step) + torch.arange(lo, hi, dtype=torch.float32)[:, None])
            done.copy_(torch.tensor([(e + step + 1) % 2 for e in range(lo, hi)], dtype=torch.uint8))
            g.start_slot(step % 2)
            if step > 0:
                r, d = g.finish_slot((step - 1) % 2)
                zc.append((r.clone().numpy(), d.clone().numpy()))
        r, d = g.finish_slot(2 % 2)
        zc.append((r.clone().numpy(), d.clone().numpy()))
        q.put((rank, init["ants_xyt"], out, zc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,E", [(2, 8), (3, 7)])
def test_sharded_envs_and_reward_gather(world, E):
    N = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = synth_init(make_cfg(E, N, 32, 32), seed=50, n_food_discs=3, food_rmin=2, food_rmax=4)["ants_xyt"]
    np.testing.assert_array_equal(np.concatenate([r[1] for r in res]), full)  # shards tile the batch
    for step in range(3):
        want_r = np.full((E, N), float(step)) + np.arange(E, dtype=np.float32)[:, None]
        want_d = np.array([(e + step) % 2 for e in range(E)], np.uint8)
        for _, _, out, zc in res:  # every rank sees the whole batch
            np.testing.assert_array_equal(out[step][0], want_r)
            np.testing.assert_array_equal(out[step][1], want_d)
            np.testing.assert_array_equal(zc[step][0], want_r + 10.0)
            np.testing.assert_array_equal(zc[step][1], 1 - want_d)


class _FakeEnv:
    """Stand-in for BatchedAntsEnv on CPU: `step_update` writes this rank's reward / done into whatever tensors
    env.reward / env.done currently point at — exactly what the kernels do with the pointers they are handed."""

    def __init__(self, lo, hi, n):
        self.lo, self.hi, self.n = lo, hi, n
        self.reward = torch.zeros((hi - lo, n), dtype=torch.float32)
        self.done = torch.zeros((hi - lo,), dtype=torch.uint8)

    def step_update(self, t):
        self.reward.copy_(torch.full((self.hi - self.lo, self.n), 100.0 + t) + torch.arange(self.lo, self.hi, dtype=torch.float32)[:, None])
        self.done.copy_(torch.tensor([(e + t) % 2 for e in range(self.lo, self.hi)], dtype=torch.uint8))


def _stepper_worker(rank, world, port, E, N, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from antsrl_amd.dist import ShardedStepper
        lo, hi = shard_range(E, rank, world)
        env = _FakeEnv(lo, hi, N)
        mode, _, algo = mode.partition("/")
        st = ShardedStepper(env, RewardGather(E, N, "cpu", algo=algo or "collective"), mode)
        seen = []
        for t in range(5):  # bench.py's loop: step, gather left in flight under the next step
            st.step(t, lambda: env.step_update(t))
        r, d = st.drain()
        seen.append((r.clone().numpy(), d.clone().numpy()))
        st.step(5, lambda: env.step_update(5))  # the loop keeps working after a drain (bench: one region after another)
        r, d = st.drain()
        seen.append((r.clone().numpy(), d.clone().numpy()))
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["staged", "zero_copy", "inline", "staged/direct", "zero_copy/direct"])
def test_bench_step_loop_two_ranks(mode):
    """bench.py's N > 1 sequence (antsrl_amd.dist.ShardedStepper: env.reward / env.done re-pointed at the gather's
    slots or snapshotted, one async all-gather per step, drain at the region's end) with two gloo ranks and a
    stand-in step writer: every rank ends up with the whole batch of the LAST step."""
    world, E, N = 2, 6, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stepper_worker, args=(r, world, port, E, N, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, seen in res:
        for (r, d), t in zip(seen, (4, 5)):
            np.testing.assert_array_equal(r, np.full((E, N), 100.0 + t) + np.arange(E, dtype=np.float32)[:, None])
            np.testing.assert_array_equal(d, np.array([(e + t) % 2 for e in range(E)], np.uint8))


class _OracleEnv:
    """Stand-in for BatchedAntsEnv on CPU with REAL arithmetic: the oracle (test infrastructure) steps this rank's block
    with the library's specification of the built-in wall jitter, keyed on the GLOBAL environment id the shard's AntsCfg
    carries (env_id_base) — what the HIP kernels do on a GPU (tests/test_gpu_shard_identity.py)."""

    def __init__(self, cfg, init):
        from oracle.oracle import Oracle
        self.orc = Oracle(cfg, init)
        self.reward = torch.zeros((cfg.n_envs, cfg.n_ants), dtype=torch.float32)
        self.done = torch.zeros((cfg.n_envs,), dtype=torch.uint8)

    def step_update(self, rot, ph):
        _, _, rew, done = self.orc.step(rot, ph, want_obs=False)
        self.orc.update(None)  # built-in jitter
        self.reward.copy_(torch.from_numpy(rew.astype(np.float32)))
        self.done.copy_(torch.from_numpy(done))


def _identity_worker(rank, world, port, E, N, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from antsrl_amd.dist import ShardedStepper, shard_cfg
        from antsrl_amd.synth import random_actions
        cfg, lo, hi = shard_cfg(E, rank, world, N, 32, 32, deposit_strength=256.0, max_time=4)
        assert cfg.env_id_base == lo and cfg.n_envs == hi - lo and cfg.n_envs_total == E
        init = synth_init(cfg, seed=50, env_offset=lo, wall_density=0.2, n_food_discs=3, food_rmin=2, food_rmax=4)
        rot, ph = random_actions(make_cfg(E, N, 32, 32), 6, seed=2)  # the whole batch's actions, sliced per rank
        env = _OracleEnv(cfg, init)
        mode, _, algo = mode.partition("/")
        st = ShardedStepper(env, RewardGather(E, N, "cpu", algo=algo or "collective"), mode)
        seen = []
        for t in range(6):
            st.step(t, lambda: env.step_update(rot[t][lo:hi], ph[t][lo:hi]))
            r, d = st.drain()
            seen.append((r.clone().numpy(), d.clone().numpy()))
        q.put((rank, seen, env.orc.theta.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,E,mode", [(2, 5, "staged"), (3, 7, "zero_copy"), (2, 6, "inline"), (3, 7, "staged/direct")])
def test_sharded_run_equals_the_single_process_batch(world, E, mode):
    """world gloo ranks step their blocks of one batch (shard_cfg: env_id_base = the block's first global id) with the
    library's OWN wall jitter and all-gather reward / done every step: every rank sees exactly what one process stepping
    the whole batch computes — sharding does not change a result (north_star: identical seeds, identical results)."""
    from antsrl_amd.synth import random_actions
    from oracle.oracle import Oracle
    N = 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_identity_worker, args=(r, world, port, E, N, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = make_cfg(E, N, 32, 32, deposit_strength=256.0, max_time=4)
    whole = Oracle(cfg, synth_init(cfg, seed=50, wall_density=0.2, n_food_discs=3, food_rmin=2, food_rmax=4))
    theta0 = whole.theta.copy()
    rot, ph = random_actions(cfg, 6, seed=2)
    for t in range(6):
        _, _, rew, done = whole.step(rot[t], ph[t], want_obs=False)
        whole.update(None)
        for _, seen, _ in res:
            np.testing.assert_array_equal(seen[t][0], rew.astype(np.float32), err_msg="step %d" % t)
            np.testing.assert_array_equal(seen[t][1], done, err_msg="step %d" % t)
        assert done.all() == (t == 3)  # RL_api.py:200: done exactly when timestep == max_time
    np.testing.assert_array_equal(np.concatenate([r[2] for r in res]), whole.theta)  # the jitter itself, bit for bit
    assert (np.abs(whole.theta - theta0) > 0).any()


def test_shard_range_partitions():
    for E in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            r = [shard_range(E, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == E
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
