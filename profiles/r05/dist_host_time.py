#!/usr/bin/env python3
"""Where does the HOST's time per step go in the N > 1 loop (one RCCL rank on one GPU)?  The c3 step takes 0.1975 ms on the
device; the loop with the per-step reward / done all-gather runs at 0.230 ms/step — host-bound?  Times (perf_counter, no
synchronisation inside the loop) of: antsrl_step_update through BatchedAntsEnv, RewardGather.start's parts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
sys.stdout.flush(); real = os.dup(1); os.dup2(2, 1)
import numpy as np, torch, torch.distributed as dist
import bench
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.dist import RewardGather, ShardedStepper
from antsrl_amd.synth import synth_init
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
W_ = bench.CONFIGS["c3"]; E, N = W_["E"], W_["N"]
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
env = BatchedAntsEnv(cfg, dev); env.tune_placement(); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8); ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
gather = RewardGather(E, N, dev)
for t in range(400): env.step_update(rot[t % 8], ph[t % 8], None)
def timed(fn, K=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(K): fn(t)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6
out = []
out.append(("step_update only", timed(lambda t: env.step_update(rot[t % 8], ph[t % 8], None))))
st = ShardedStepper(env, gather, "staged")
out.append(("step + staged gather", timed(lambda t: st.step(t, lambda: env.step_update(rot[t % 8], ph[t % 8], None))))); st.drain()
zc = ShardedStepper(env, gather, "zero_copy")
out.append(("step + zero-copy gather", timed(lambda t: zc.step(t, lambda: env.step_update(rot[t % 8], ph[t % 8], None))))); zc.drain()
send = torch.zeros((E, N + 1), dtype=torch.float32, device=dev); recv = torch.empty_like(send)
out.append(("all_gather_into_tensor(async) + wait alone", timed(lambda t: dist.all_gather_into_tensor(recv, send, async_op=True).wait())))
out.append(("two copy_ kernels alone", timed(lambda t: (send[:, :N].copy_(env.reward), send[:, N].copy_(env.done)))))
txt = "\n".join("%-46s host enqueue %7.1f us/step   wall %7.1f us/step" % (k, a, b) for k, (a, b) in out)
os.write(real, (txt + "\n").encode())
dist.destroy_process_group()
