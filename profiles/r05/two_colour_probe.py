#!/usr/bin/env python3
"""Is "slow" a property of the observation buffer, of the workspace, or of the PAIR — and if of the pair, does a two-colouring
explain it (slow iff both buffers have the same colour)?

One process, c3.  Workspaces and output buffers of both kinds (torch.empty / antsrl_mem_alloc) allocated at different
depths of a walk through the device's memory (5 GiB spacers), then EVERY workspace x EVERY output buffer stepped at the
same age of a fresh scratch episode.  Prints the matrix (F < 0.208 ms/step, S > 0.225, m between) and the best
two-colouring's misfit.

    python profiles/r05/two_colour_probe.py"""
import itertools
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
    n_out, n_ws = env._out_total + 256, env.workspace_bytes + 256

    def tor(n):
        return torch.empty(n, dtype=torch.uint8, device=dev)
    wss, outs, spacers = [("wsT@0", env._ws)], [("oT@0", env._out_flat)], []
    depth = 0
    for stage in range(6):
        outs.append(("oP@%d" % depth, vmm.pieced_u8(n_out, dev)))
        outs.append(("oT@%d" % depth, tor(n_out)))
        outs.append(("oP'@%d" % depth, vmm.pieced_u8(n_out, dev)))
        if stage in (0, 2, 5):
            wss.append(("wsP@%d" % depth, vmm.pieced_u8(n_ws, dev)))
        if stage in (2, 5):
            wss.append(("wsT@%d" % depth, tor(n_ws)))
        for _ in range(7):
            if torch.cuda.mem_get_info()[0] > 20 * 2 ** 30:
                spacers.append(tor(5 << 30))
                depth += 5
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(steps=24):
        env.generate(cm.make_gen(), episode_seed=0x7A11)
        for t in range(154):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e0.record()
        for t in range(steps):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps
    mat = np.zeros((len(wss), len(outs)))
    for wi, (_, ws) in enumerate(wss):
        env._make_handle(ws)
        for oi, (_, o) in enumerate(outs):
            env._bind_outputs(o)
            mat[wi, oi] = measure()
    print("ms per step; rows: workspace (kind @ GiB allocated before it), columns: output buffer")
    print("%-10s" % "" + "".join("%9s" % k for k, _ in outs))
    for wi, (k, _) in enumerate(wss):
        print("%-10s" % k + "".join("%9.4f" % v for v in mat[wi]))
    print()
    for wi, (k, _) in enumerate(wss):
        print("%-10s" % k + "".join("%9s" % ("F" if v < 0.208 else "S" if v > 0.225 else "m") for v in mat[wi]))
    # best two-colouring: slow (1) iff colour(ws) == colour(out); misfit counted over clear cells only
    clear = (mat < 0.208) | (mat > 0.225)
    slow = mat > 0.225
    best = None
    for cw in itertools.product((0, 1), repeat=len(wss)):
        co = []
        miss = 0
        for oi in range(len(outs)):
            m0 = sum(1 for wi in range(len(wss)) if clear[wi, oi] and slow[wi, oi] != (cw[wi] == 0))
            m1 = sum(1 for wi in range(len(wss)) if clear[wi, oi] and slow[wi, oi] != (cw[wi] == 1))
            co.append(0 if m0 <= m1 else 1)
            miss += min(m0, m1)
        if best is None or miss < best[0]:
            best = (miss, cw, tuple(co))
    print("\nbest two-colouring (slow iff same colour): %d of %d clear cells misfit; workspaces %s, outputs %s" %
          (best[0], int(clear.sum()), best[1], best[2]))
    # and the simpler model: slow is a property of the output buffer alone
    col_miss = sum(min(int((clear[:, oi] & slow[:, oi]).sum()), int((clear[:, oi] & ~slow[:, oi]).sum())) for oi in range(len(outs)))
    print("model 'a property of the output buffer alone': %d of %d clear cells misfit" % (col_miss, int(clear.sum())))


if __name__ == "__main__":
    main()
