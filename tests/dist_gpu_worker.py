"""Worker of tests/test_a_gpu_dist.py: one rank per GPU under torch.distributed.run (backend "nccl" = RCCL).

Every rank steps ITS contiguous block of a global batch with `ShardedStepper` (the loop bench.py --gpus N runs), in
`staged` and in `zero_copy` mode, and also steps the WHOLE batch alone on its own GPU: the gathered reward / done of every
step must equal the single-rank result bit for bit (environments are independent and every shard carries its global
environment ids — AntsCfg.env_id_base — so sharding must not change a value, the library's own wall jitter included).
Prints "DIST_GPU_OK <rank>" and exits 0 on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # ANTSRL_DIST_ONE_GPU=1 (a one-GPU box's rehearsal of world > 1): every rank on device 0, the collective over gloo —
    # RCCL refuses two ranks on one device.  Everything but the transport is the real thing: the HIP kernels, one process
    # per rank, shard_cfg's global env ids, ShardedStepper in both modes on device tensors.
    one_gpu = os.environ.get("ANTSRL_DIST_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if one_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.dist import RewardGather, ShardedStepper, shard_cfg, shard_range
    from antsrl_amd.synth import synth_init
    assert dist.get_world_size() == world
    E, N, steps = 6 * world + 1, 96, 8  # (ragged: the blocks differ by one environment)
    lo, hi = shard_range(E, rank, world)
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    rot = torch.randint(-1, 2, (steps, E, N), generator=g, dtype=torch.int8).to(dev)
    ph = torch.randint(0, 3, (steps, E, N), generator=g, dtype=torch.int8).to(dev)
    # the whole batch on this GPU alone: the reference every sharded run must reproduce
    cfg_all = cm.make_cfg(E, N, 64, 64, n_rocks=2, deposit_strength=256.0)
    ref = BatchedAntsEnv(cfg_all, dev)
    ref.reset(synth_init(cfg_all, seed=9, wall_density=0.08, n_food_discs=4, food_rmin=2, food_rmax=4))
    want = []
    for t in range(steps):
        ref.step_update(rot[t], ph[t], None)
        want.append((ref.reward.clone(), ref.done.clone()))
    for mode in ("staged", "zero_copy"):
        # the shard knows its place in the batch: env_id_base = lo keys the library's wall jitter on the GLOBAL env id
        cfg, lo_c, hi_c = shard_cfg(E, rank, world, N, 64, 64, n_rocks=2, deposit_strength=256.0)
        assert (lo_c, hi_c) == (lo, hi) and cfg.env_id_base == lo and cfg.n_envs_total == E
        env = BatchedAntsEnv(cfg, dev)
        env.reset(synth_init(cfg, seed=9, wall_density=0.08, env_offset=lo, n_food_discs=4, food_rmin=2, food_rmax=4))
        gather = RewardGather(E, N, dev)
        stepper = ShardedStepper(env, gather, mode)
        for t in range(steps):
            stepper.step(t, lambda: env.step_update(rot[t, lo:hi].contiguous(), ph[t, lo:hi].contiguous(), None))
            if t % 3 == 2 or t == steps - 1:  # drain now and then: every gathered batch is checked at least at these steps
                rew_all, done_all = stepper.drain()
                assert rew_all.shape == (E, N) and done_all.shape == (E,)
                assert torch.equal(rew_all, want[t][0]), "%s: reward of step %d differs from the single-rank run" % (mode, t)
                assert torch.equal(done_all.to(torch.uint8), want[t][1]), "%s: done of step %d" % (mode, t)
        del env, gather, stepper
    dist.barrier()
    print("DIST_GPU_OK %d of %d" % (rank, world), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
