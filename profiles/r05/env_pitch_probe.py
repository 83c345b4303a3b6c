#!/usr/bin/env python3
"""VERDICT r4 item 3 with the REAL kernel: what is it about the observation tensor's placement that k_perceive feels?

Physical addresses of device memory are not visible from user space (no HIP call returns them, pagemap does not cover
VRAM), so the high address bits of a buffer cannot be tabulated — only drawn (round 4: torch.empty / pieces of 2 .. 1024 MiB /
shuffled mapping order / offsets inside an allocation).  What CAN be set is the stream's own geometry.  The write stream
advances 43 904 bytes per workgroup and 702 464 bytes per environment, and eight consecutive workgroups belong to eight
consecutive environments (one per XCD); the records they gather lie 1 MiB per environment apart.  This probe pads the
environment pitch of the observation tensor (profiling library: antsrl_debug_set_obs_env_pad; the caller's buffer is
E * (N * row + pad) bytes) and times k_perceive by the library's HIP events, on a torch.empty buffer (physically contiguous
in pieces of hundreds of MiB: the usual SLOW placement) and on antsrl_mem_alloc pieces of 16 MiB (the usual fast one), same
handle, same workspace, same episode, alternating.

    ANTSRL_LIB=antsrl_amd/lib/libantsrl_hip_prof.so python profiles/r05/env_pitch_probe.py [c3|c2|c4]"""
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
from antsrl_amd.synth import synth_init


def main(name):
    W_ = dict(bench.CONFIGS[name])
    E, N, W, H, R = W_["E"], W_["N"], W_["W"], W_["H"], W_["R"]
    extra = dict(n_rocks=R, deposit_strength=256.0, max_time=1 << 30)
    if W_["radius3"]:
        ax = np.arange(-3, 4)
        g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
        extra["filt"] = g / g.sum() * (1 - 0.001)
    cfg = cm.make_cfg(E, N, W, H, **extra)
    dev = torch.device("cuda", 0)
    env = BatchedAntsEnv(cfg, dev, pieced_memory=False)  # (workspace and the small outputs: torch.empty)
    lib = env.lib
    lib.antsrl_debug_set_obs_env_pad.argtypes = [C.c_uint32]
    env.reset(synth_init(cfg, seed=1234))
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
    for t in range(400):
        env.step_update(rot[t % 8], ph[t % 8], None)
    row_bytes = cfg.pside * cfg.pside * cfg.n_channels * 4
    env_bytes = N * row_bytes
    # pads: none; small ones; pitches rounded up to 4 KiB / 64 KiB / 1 MiB / 2 MiB multiples, and those + one 128-byte line
    def up(v, m):
        return (v + m - 1) // m * m
    pads = [0, 128, 256, 1024, 4096 - env_bytes % 4096, 65536 - env_bytes % 65536, up(env_bytes, 1 << 20) - env_bytes,
            up(env_bytes, 1 << 20) - env_bytes + 128, up(env_bytes, 1 << 20) - env_bytes + 4096 + 128, up(env_bytes, 2 << 20) - env_bytes]
    max_pad = max(pads)
    nbytes = E * (env_bytes + max_pad) + 4096
    bufs = {"torch.empty": torch.empty(nbytes, dtype=torch.uint8, device=dev),
            "16 MiB pieces": vmm.pieced_u8(nbytes, dev),
            "torch.empty (2nd)": torch.empty(nbytes, dtype=torch.uint8, device=dev),
            "16 MiB pieces (2nd)": vmm.pieced_u8(nbytes, dev)}
    NEV = cm.TIMING_EVENTS
    K = 30
    print("%s: %d envs x %d ants, row %d B, env pitch %d B (%% 4096 = %d, %% 65536 = %d, %% 1 MiB = %d); k_perceive ms, mean of %d launches" %
          (name, E, N, row_bytes, env_bytes, env_bytes % 4096, env_bytes % 65536, env_bytes % (1 << 20), K))
    print("%-22s" % "pad (pitch)" + "".join("%22s" % k for k in bufs))
    t = 400
    for pad in pads:
        _lib.check(lib.antsrl_debug_set_obs_env_pad(pad), "set_obs_env_pad")
        line = "%-22s" % ("%d (%d)" % (pad, env_bytes + pad))
        for k, b in bufs.items():
            base = (-b.data_ptr()) % 256
            env.obs = b[base:base + E * (env_bytes + pad)].view(torch.float32)
            for _ in range(4):
                env.step_update(rot[t % 8], ph[t % 8], None)
                t += 1
            evs = bench.HipEvents(NEV * K)
            for i in range(K):
                env.set_timing_events([evs.ev[NEV * i + j].value for j in range(NEV)])
                env.step_update(rot[t % 8], ph[t % 8], None)
                t += 1
            torch.cuda.synchronize(dev)
            line += "%22.4f" % float(np.mean([evs.elapsed_ms(NEV * i + 2, NEV * i + 3) for i in range(K)]))
            evs.destroy()
        print(line, flush=True)
    _lib.check(lib.antsrl_debug_set_obs_env_pad(0), "set_obs_env_pad")


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["c3"]):
        main(c)
