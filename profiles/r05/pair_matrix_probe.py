#!/usr/bin/env python3
"""Is a slow placement a property of ONE buffer or of the PAIR (workspace, observation tensor)?

One process, c3: four workspaces (two torch.empty, two antsrl_mem_alloc) x six output buffers (three of each kind), every
pair stepped 30 times at the same point of the same scratch episode (the machinery of BatchedAntsEnv.tune_placement),
ms per step; the whole matrix twice (the second pass in reverse order).  Virtual addresses printed beside the labels.

    python profiles/r05/pair_matrix_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from antsrl_amd import config as cm, vmm
from antsrl_amd.batched import BatchedAntsEnv


def main():
    W_ = bench.CONFIGS["c3"]
    E, N = W_["E"], W_["N"]
    cfg = cm.make_cfg(E, N, W_["W"], W_["H"], n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
    dev = torch.device("cuda", 0)
    env = BatchedAntsEnv(cfg, dev, pieced_memory=False)
    n_ws, n_out = env.workspace_bytes + 256, env._out_total + 256

    def tor(n):
        return torch.empty(n, dtype=torch.uint8, device=dev)
    wss = [("ws torch A", env._ws), ("ws pieced A", vmm.pieced_u8(n_ws, dev)), ("ws torch B", tor(n_ws)), ("ws pieced B", vmm.pieced_u8(n_ws, dev))]
    outs = [("out torch A", env._out_flat), ("out pieced A", vmm.pieced_u8(n_out, dev).zero_()), ("out torch B", tor(n_out).zero_()),
            ("out pieced B", vmm.pieced_u8(n_out, dev).zero_()), ("out torch C", tor(n_out).zero_()), ("out pieced C", vmm.pieced_u8(n_out, dev).zero_())]
    for k, b in wss + outs:
        print("%-14s virtual 0x%012x  %6.0f MiB" % (k, b.data_ptr(), b.numel() / 2 ** 20))
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(steps=30):
        for t in range(4):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e0.record()
        for t in range(steps):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    for pas in range(2):
        mat = np.zeros((len(wss), len(outs)))
        order_w = list(range(len(wss)))[::-1 if pas else 1]
        for wi in order_w:
            env._make_handle(wss[wi][1])
            env.generate(cm.make_gen(), episode_seed=0x7A11)
            for t in range(150):
                env.step_update(rot[t % 4], ph[t % 4], None)
            for oi in list(range(len(outs)))[::-1 if pas else 1]:
                env._bind_outputs(outs[oi][1])
                mat[wi, oi] = measure()
        print("pass %d: ms per step (rows: workspace, columns: output buffer)" % (pas + 1))
        print("%-14s" % "" + "".join("%14s" % k for k, _ in outs))
        for wi, (k, _) in enumerate(wss):
            print("%-14s" % k + "".join("%14.4f" % v for v in mat[wi]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
