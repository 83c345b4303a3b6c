#!/usr/bin/env python3
"""Zones, one step further: k_perceive is slow when the observation tensor (a streaming write) lies in the zone of the
workspace — but the workspace holds two different things, the CELL RECORDS (scattered gathers, scattered read-modify-writes)
and the ants' STATE arrays (streaming reads and writes of k_update_move).  Does k_update_move gain when its streams and its
scattered accesses lie in different zones, as k_perceive does?  Profiling library: antsrl_debug_set_cells_base puts the
interleaved cell records at a caller-supplied address.  One process, c3 / c2 / c5-shaped, fresh scratch episode per cell;
k_update_move and k_perceive by the library's HIP events.

    ANTSRL_LIB=antsrl_amd/lib/libantsrl_hip_prof.so python profiles/r05/split_workspace_probe.py [c3 c2]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ANTSRL_LIB", os.path.join(ROOT, "antsrl_amd", "lib", "libantsrl_hip_prof.so"))
import numpy as np
import torch

import bench
from antsrl_amd import _lib, config as cm, vmm
from antsrl_amd.batched import BatchedAntsEnv


def main(name):
    W_ = bench.CONFIGS[name]
    E, N, W, H = W_["E"], W_["N"], W_["W"], W_["H"]
    cfg = cm.make_cfg(E, N, W, H, n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
    dev = torch.device("cuda", 0)
    env = BatchedAntsEnv(cfg, dev, pieced_memory=False)
    lib = env.lib
    lib.antsrl_debug_set_cells_base.argtypes = [C.c_void_p, C.c_void_p]
    n_out, n_ws, n_cells = env._out_total + 256, env.workspace_bytes + 256, 16 * E * W * H + 256

    def tor(n):
        return torch.empty(n, dtype=torch.uint8, device=dev)

    def pie(n):
        return vmm.pieced_u8(n, dev) if n >= vmm.SMALL_BYTES else tor(n)
    ws = {"T": env._ws, "P": pie(n_ws)}
    cells = {"in ws": None, "T": tor(n_cells), "P": pie(n_cells)}
    outs = {"T": env._out_flat, "P": pie(n_out).zero_()}
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
    NEV, K = cm.TIMING_EVENTS, 24
    print("%s: ms — step | k_update_move | k_perceive;  state = the workspace's kind, cells = where the records lie, out = outputs" % name)
    for wk, w in ws.items():
        for ck, cbuf in cells.items():
            for ok, o in outs.items():
                env._make_handle(w)
                if cbuf is not None:
                    base = cbuf.data_ptr() + (-cbuf.data_ptr()) % 256
                    _lib.check(lib.antsrl_debug_set_cells_base(env._h, C.c_void_p(base)), "set_cells_base")
                env._bind_outputs(o)
                env.generate(cm.make_gen(), episode_seed=0x7A11)
                for t in range(154):
                    env.step_update(rot[t % 4], ph[t % 4], None)
                evs = bench.HipEvents(NEV * K)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(K):
                    env.set_timing_events([evs.ev[NEV * i + j].value for j in range(NEV)])
                    env.step_update(rot[i % 4], ph[i % 4], None)
                e1.record()
                e1.synchronize()
                um = float(np.mean([evs.elapsed_ms(NEV * i + 1, NEV * i + 2) for i in range(K)]))
                pr = float(np.mean([evs.elapsed_ms(NEV * i + 2, NEV * i + 3) for i in range(K)]))
                evs.destroy()
                print("   state %s  cells %-5s  out %s :  %.4f | %.4f | %.4f" % (wk, ck, ok, e0.elapsed_time(e1) / K, um, pr), flush=True)


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["c3", "c2"]):
        main(c)
