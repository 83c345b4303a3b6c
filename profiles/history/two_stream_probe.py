"""Does splitting the c3 batch into two half-batches on two HIP streams help (the latency / random-access bound
k_update_move of one half under the streaming k_perceive of the other)?  Two BatchedAntsEnv handles of 512 envs on two
streams against one handle of 1024 envs.  usage: python profiles/two_stream_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

dev = torch.device("cuda:0")
N, W, H, R = 512, 256, 256, 8


def make(E, off):
    cfg = cm.make_cfg(E, N, W, H, n_rocks=R, deposit_strength=256.0, max_time=1 << 30)
    env = BatchedAntsEnv(cfg, dev)
    env.reset(synth_init(cfg, seed=1234, env_offset=off))
    g = torch.Generator(device=dev); g.manual_seed(99 + off)
    rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    return env, rot, ph


def run(parts, streams, steps, warm):
    for t in range(warm + steps):
        if t == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for (env, rot, ph), st in zip(parts, streams):
            with torch.cuda.stream(st):
                env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = [make(1024, 0)]
ms1 = run(one, [torch.cuda.current_stream(dev)], 400, 420)
del one; torch.cuda.empty_cache()
two = [make(512, 0), make(512, 512)]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ms2 = run(two, [sa, sb], 400, 420)
ms2s = run(two, [sa, sa], 400, 0)
print("one handle x 1024 envs: %.4f ms/step;  two x 512 on two streams: %.4f;  two x 512 on one stream: %.4f" % (ms1, ms2, ms2s))
four = None
