#!/usr/bin/env python3
"""Where in the device's memory is an observation buffer fast?  One process, c3, workspace on torch.empty (allocated first):
a series of output buffers (antsrl_mem_alloc, 16 MiB pieces) with SPACERS of a few GB between them, so that the series
walks through the device's physical memory; every buffer is stepped at the same age of a fresh scratch episode.  Then the
same series of buffers against a SECOND workspace allocated at the end of the walk: is "slow" a property of the region
(absolute) or of the buffer pair (relative)?

    python profiles/r05/region_map_probe.py [buffers] [spacer_GiB]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from antsrl_amd import config as cm, vmm
from antsrl_amd.batched import BatchedAntsEnv


def main(nbuf, spacer_gib):
    W_ = bench.CONFIGS["c3"]
    E, N = W_["E"], W_["N"]
    cfg = cm.make_cfg(E, N, W_["W"], W_["H"], n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
    dev = torch.device("cuda", 0)
    env = BatchedAntsEnv(cfg, dev, pieced_memory=False)
    ws_a = env._ws
    n_out, n_ws = env._out_total + 256, env.workspace_bytes + 256
    free0 = torch.cuda.mem_get_info()[0]
    bufs, spacers = [], []
    for i in range(nbuf):
        bufs.append(vmm.pieced_u8(n_out, dev))
        if spacer_gib > 0 and torch.cuda.mem_get_info()[0] > (spacer_gib + 8) * 2 ** 30:
            spacers.append(torch.empty(int(spacer_gib * 2 ** 30), dtype=torch.uint8, device=dev))
    ws_b = torch.empty(n_ws, dtype=torch.uint8, device=dev)
    print("free at start %.1f GiB, after the walk %.1f GiB; %d buffers, %d spacers of %.1f GiB" %
          (free0 / 2 ** 30, torch.cuda.mem_get_info()[0] / 2 ** 30, len(bufs), len(spacers), spacer_gib))
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(steps=30):
        env.generate(cm.make_gen(), episode_seed=0x7A11)
        for t in range(154):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e0.record()
        for t in range(steps):
            env.step_update(rot[t % 4], ph[t % 4], None)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps
    for name, ws in (("workspace A (allocated first)", ws_a), ("workspace B (allocated last)", ws_b), ("workspace A again", ws_a)):
        env._make_handle(ws)
        t = []
        for b in bufs:
            env._bind_outputs(b)
            t.append(measure())
        print("%-32s %s" % (name, " ".join("%.3f" % x for x in t)))
        print("%-32s %s" % ("", "".join("F" if x < 0.208 else ("S" if x > 0.225 else "m") for x in t)))
        sys.stdout.flush()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, float(sys.argv[2]) if len(sys.argv) > 2 else 5.0)
