#!/usr/bin/env python3
"""How often does an observation buffer of each KIND land on the slow side, with the workspace on torch.empty (the product's
pair)?  One process, c3, ten buffers of each kind alive at once (distinct physical memory), each stepped 30 times at the
same point of the same scratch episode:
   torch.empty | 16 MiB pieces in allocation order (antsrl_mem_alloc, the product) | 16 MiB pieces mapped in SHUFFLED order |
   2 MiB pieces shuffled | 64 MiB pieces in order
(ctypes on the HIP virtual-memory API for the non-product kinds: profiles/history/r04/vmm_ctypes.py.)

    python profiles/r05/draw_distribution_probe.py [draws]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles", "history", "r04"))
import numpy as np
import torch

import bench
from antsrl_amd import config as cm, vmm
from antsrl_amd.batched import BatchedAntsEnv
from vmm_ctypes import ShuffledBuffer


def main(draws):
    W_ = bench.CONFIGS["c3"]
    E, N = W_["E"], W_["N"]
    cfg = cm.make_cfg(E, N, W_["W"], W_["H"], n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
    dev = torch.device("cuda", 0)
    env = BatchedAntsEnv(cfg, dev, pieced_memory=False)
    n_out = env._out_total + 256
    keep = []

    def shuffled(piece, shuffle, seed):
        b = ShuffledBuffer(n_out, dev, seed=seed, shuffle=shuffle, chunk_bytes=piece)
        keep.append(b)
        return b.tensor
    kinds = {
        "torch.empty": lambda i: torch.empty(n_out, dtype=torch.uint8, device=dev),
        "16 MiB in order (product)": lambda i: vmm.pieced_u8(n_out, dev),
        "16 MiB shuffled": lambda i: shuffled(16 << 20, True, 100 + i),
        "2 MiB shuffled": lambda i: shuffled(2 << 20, True, 200 + i),
        "64 MiB in order": lambda i: shuffled(64 << 20, False, 0),
    }
    bufs = {k: [] for k in kinds}
    for i in range(draws):  # interleaved allocation: every kind draws from the same moments of the allocator
        for k, mk in kinds.items():
            bufs[k].append(mk(i).zero_())
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(steps=30):
        # a FRESH scratch episode for every buffer: a step costs more the older the episode is (+10 % from step 150 to step
        # 1850: this probe's first version, draw_distribution_confounded.txt) — every buffer is measured at the same age
        env.generate(cm.make_gen(), episode_seed=0x7A11)
        for t in range(150):
            env.step_update(rot[t % 4], ph[t % 4], None)
        for t in range(4):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e0.record()
        for t in range(steps):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps
    res = {k: [] for k in kinds}
    for i in range(draws):
        for k in kinds:
            env._bind_outputs(bufs[k][i])
            res[k].append(measure())
    print("c3, workspace torch.empty, %d output buffers of each kind, ms per step:" % draws)
    for k, v in res.items():
        v = np.array(v)
        print("  %-28s min %.4f  median %.4f  max %.4f   %s" % (k, v.min(), np.median(v), v.max(), " ".join("%.4f" % x for x in v)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
