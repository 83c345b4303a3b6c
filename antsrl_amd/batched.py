"""BatchedAntsEnv — E independent AntsRL environments stepped on one MI355X.

Thin host-side plumbing over the C-ABI (include/antsrl.h): PyTorch-ROCm supplies device memory
(one workspace tensor + the observation / reward / done output tensors) and the stream; every
piece of arithmetic happens in the HIP kernels of libantsrl_hip.so.  There is no CPU fallback.

Call order mirrors the reference driver (main.py:88-131):
    env.reset(init); obs, agent_state, _ = env.observe()
    loop:  obs, agent_state, reward, done = env.step(rotation, pheromone); env.update()
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from . import config as cfgmod
from . import vmm
from .config import AntsCfg, AntsInit

_STATE_DTYPES = {
    cfgmod.S_ANTS_XYT: torch.float64, cfgmod.S_PREV_XY: torch.float64, cfgmod.S_HOLDING: torch.float32,
    cfgmod.S_MANDIBLES: torch.uint8, cfgmod.S_ACTIVATION: torch.float32, cfgmod.S_PHERO: torch.float32,
    cfgmod.S_FOOD: torch.float32, cfgmod.S_EXPLORED: torch.uint8, cfgmod.S_ANTHILL_FOOD: torch.float64,
    cfgmod.S_ROCK_CENTERS: torch.float64, cfgmod.S_TIMESTEP: torch.int32, cfgmod.S_REWARD_STATE: torch.uint8,
    cfgmod.S_WALLS: torch.uint8, cfgmod.S_ANTHILL_AREA: torch.uint8, cfgmod.S_SEED: torch.float32,
    cfgmod.S_ANTHILL_XYR: torch.int32, cfgmod.S_ROCK_RW: torch.float64,
    cfgmod.S_PHERO_C0: torch.float32, cfgmod.S_PHERO_C1: torch.float32, cfgmod.S_PHERO_C2: torch.float32,
    cfgmod.S_PHERO_C3: torch.float32,
}


import contextlib

_NULL_CTX = contextlib.nullcontext()
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def placement_levels_seen(times) -> bool:
    """tune_placement's stopping rule: have the trials — the observation kernel's times — shown BOTH placement levels (15 %
    apart at c3)?  A spread of 6 % or more among the trials — not counting a trial far above everything else, which is not a level (a
    first touch, a profiler's hiccup: 0.32 ms among 0.233s on a device whose every buffer sat on the slow level)."""
    ok = [t for t in times if t < 1.25 * min(times)]
    return max(ok) >= 1.06 * min(ok)  # (k_perceive: 15-18 % apart at c3 / c4 / c5, 3-4 % at c2-sized batches — which then walk)


class BatchedAntsEnv:
    def __init__(self, cfg: AntsCfg, device: Optional[torch.device] = None, obs_dtype: torch.dtype = torch.float32,
                 obs_row_stride=None, pieced_memory: bool = True):
        """obs_dtype: torch.float32 (the reference's values) or torch.bfloat16 (the same values rounded
        to nearest even: half the bytes per step; what the bf16 policy rounds its input to anyway —
        antsrl_set_obs_format).
        obs_row_stride: None (dense, the default) or "line": every ant's row of P*P*K values starts on a 128-byte line
        (antsrl_set_obs_row_stride; float32 7x7x7: 352 elements per row instead of 343).  `self.obs` keeps the reference's
        shape [E, N, P, P, K] — then as a strided VIEW of `self.obs_padded` [E, N, stride], whose padding is zeros.
        pieced_memory: the output buffer (the observation tensor) comes from antsrl_mem_alloc (physical pieces of 16 MiB)
        instead of torch.empty; the workspace stays a torch allocation.  On MI355X the observation kernel runs 15 % apart
        depending on the physical layout of these two buffers; this combination is the fast one on most boxes, two torch
        allocations are mostly slow (antsrl_amd/vmm.py, profiles/history/r04/placement_probe*.txt).  tune_placement() measures the
        four combinations on the box at hand and keeps the fastest."""
        if obs_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("obs_dtype must be torch.float32 or torch.bfloat16")
        if not torch.cuda.is_available():
            raise _lib.AntsrlError("BatchedAntsEnv needs an MI355X (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
        self.lib = _lib.load()
        self.cfg = cfg.copy()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        need = C.c_size_t()
        _lib.check(self.lib.antsrl_workspace_bytes(C.byref(self.cfg), C.byref(need)), "workspace_bytes")
        self.workspace_bytes = need.value
        with torch.cuda.device(self.device):
            big = (lambda n: vmm.empty_u8(n, self.device)) if pieced_memory else (lambda n: torch.empty(n, dtype=torch.uint8, device=self.device))
            self._pieced = bool(pieced_memory)
            self._h = None
            self._make_handle(torch.empty(need.value + 256, dtype=torch.uint8, device=self.device))
            E, N, P, K = cfg.n_envs, cfg.n_ants, cfg.pside, cfg.n_channels
            esz = 4 if obs_dtype == torch.float32 else 2
            row = P * P * K
            if obs_row_stride not in (None, "line"):
                raise ValueError("obs_row_stride must be None or 'line'")
            pitch = row if obs_row_stride is None else (row * esz + 127) // 128 * 128 // esz
            sizes = [("agent_state", E * N * 2 * 4), ("reward", E * N * 4), ("done", E), ("obs", E * N * pitch * esz)]
            offs, total = {}, 0
            for name, nbytes in sizes:
                offs[name] = total
                total += (nbytes + 255) // 256 * 256
            self._out_sizes, self._out_offs, self._out_total, self._small_bytes = sizes, offs, total, offs["obs"]
            self._obs_dtype, self._obs_pitch_row = obs_dtype, (pitch, row)
            if obs_dtype == torch.bfloat16:
                _lib.check(self.lib.antsrl_set_obs_format(self._h, 1), "set_obs_format")
            if pitch != row:
                _lib.check(self.lib.antsrl_set_obs_row_stride(self._h, pitch), "set_obs_row_stride")
            self.obs_row_pitch = pitch
            self._bind_outputs(big(total + 256).zero_())
        self._keep = None
        self._loaded = None     # what has been done to the handle since construction (tune_placement refuses to run then)
        self._host_out = None   # pinned mirror of _out_flat (outputs_to_host)
        self._host_act = None   # pinned staging of numpy actions + the event of its last upload
        self._act_event = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.antsrl_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ helpers
    def _bind_outputs(self, flat: torch.Tensor) -> None:
        """The four step outputs as views of ONE device buffer `flat` (256-byte aligned pieces, the small ones first):
        outputs_to_host() brings them over in a single copy."""
        c = self.cfg
        E, N, P, K = c.n_envs, c.n_ants, c.pside, c.n_channels
        sizes, offs, total = self._out_sizes, self._out_offs, self._out_total
        pitch, row = self._obs_pitch_row
        base = (-flat.data_ptr()) % 256
        self._out_flat = flat[base:base + total]

        def piece(name, nbytes, dtype, shape):
            return self._out_flat[offs[name]:offs[name] + nbytes].view(dtype).view(shape)
        if pitch == row:
            self.obs = piece("obs", sizes[3][1], self._obs_dtype, (E, N, P, P, K))
            self.obs_padded = None
        else:  # rows a whole number of 128-byte lines apart; the reference's shape is a view of the padded buffer
            self.obs_padded = piece("obs", sizes[3][1], self._obs_dtype, (E, N, pitch))
            self.obs = self.obs_padded[..., :row].unflatten(-1, (P, P, K))
        self.agent_state = piece("agent_state", sizes[0][1], torch.float32, (E, N, 2))
        self.reward = piece("reward", sizes[1][1], torch.float32, (E, N))
        self.done = piece("done", sizes[2][1], torch.uint8, (E,))
        self._host_out = None

    def _make_handle(self, ws: torch.Tensor) -> None:
        """(Re)creates the handle over workspace tensor `ws` with this env's configuration and observation format."""
        if getattr(self, "_h", None):
            self.lib.antsrl_destroy(self._h)
        self._ws = ws
        self._ws_ptr = ws.data_ptr() + (-ws.data_ptr()) % 256
        self._h = C.c_void_p()
        _lib.check(self.lib.antsrl_create(C.byref(self.cfg), C.c_void_p(self._ws_ptr), self.workspace_bytes, C.byref(self._h)), "create")
        if getattr(self, "_obs_dtype", torch.float32) == torch.bfloat16:
            _lib.check(self.lib.antsrl_set_obs_format(self._h, 1), "set_obs_format")
        pitch, row = getattr(self, "_obs_pitch_row", (0, 0))
        if pitch != row:
            _lib.check(self.lib.antsrl_set_obs_row_stride(self._h, pitch), "set_obs_row_stride")

    def tune_placement(self, age: int = 150, steps: int = 30, verbose: bool = False, extra_outputs: int = 4,
                       walk_spacers: int = 3, spacer_gib: float = 24.0, force_walk: bool = False):
        """Pick the (workspace, output buffer) pair whose PHYSICAL placement steps fastest.  Call it right after construction,
        BEFORE reset() / generate() and before anything is attached to the handle: it runs scratch episodes (device
        generator + uniform random actions), may re-create the handle on another workspace, and leaves it to be reset.

        Why (profiles/r05/two_colour.txt, region_map.txt): the device's memory falls into a few large
        ZONES (tens of GB each), and the observation kernel runs 15 % slower when the observation tensor — a streaming write
        — and the workspace's cell records — scattered gathers — lie in the SAME zone (two levels, 0.169 / 0.197 ms at c3;
        every buffer keeps its level against every workspace of one zone and has the opposite level against a workspace of
        another zone).  Zones cannot be seen (no call returns a physical address) but they can be measured.  In a fresh
        process hipMalloc (torch.empty) and antsrl_mem_alloc draw from different zones, so the env's default pair
        (workspace torch.empty, outputs pieced) is a fast one and two torch.empty buffers the slow one — until other
        allocations have moved either allocator into the other's zone.  So: the four {torch.empty, pieced} x {torch.empty,
        pieced} pairs and `extra_outputs` more draws of the default kind are each stepped `steps` times at the same point of
        a scratch episode; if they show BOTH levels (a spread of 6 % or more) the fastest pair is on the fast one and is
        kept.  If every pair sits on ONE level (all fast — or all in one zone), the tuner walks: up to `walk_spacers` times
        it takes a `spacer_gib` GiB spacer from each allocator (so that the next buffers come from further into the device's
        memory), draws one more output buffer of each kind and one more workspace and measures; it stops as soon as both
        levels have been seen (`force_walk`: walk regardless — tests).  Everything but
        the kept pair is freed at the end (pieced buffers go back to the library's pool, spacers to the driver).
        One-off cost ~0.4 s at c3 (plus ~0.1 s per walk step).  Returns the ms/step figures, the four pairs first (None for
        small batches, where there is nothing to alias)."""
        if self._out_total < vmm.SMALL_BYTES:
            return None
        if self._loaded:
            # (it re-creates the handle: an episode, an activation matrix, an attached policy or timing events would be
            #  dropped silently, and a scratch episode would be left in their place — ADVICE r4)
            raise _lib.AntsrlError("tune_placement() must run right after construction: this handle has already been %s" % self._loaded)
        c = self.cfg
        E, N = c.n_envs, c.n_ants
        dev = self.device

        def torch_u8(n):
            return torch.empty(n, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            g = torch.Generator(device=dev)
            g.manual_seed(7)
            rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
            ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8) if c.n_phero == 2 else [None] * 4
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def scratch_episode():
                self.generate(cfgmod.make_gen(), episode_seed=0x7A11)
                for t in range(age):
                    self.step_update(rot[t % 4], ph[t % 4], None)

            NEV, PSTEPS = cfgmod.TIMING_EVENTS, 8
            evs = _lib.HipEvents(NEV * PSTEPS)
            prc_times = []  # the observation kernel alone, per trial: what the zones act on (the levels are read off THIS)

            def measure():
                for t in range(4):
                    self.step_update(rot[t % 4], ph[t % 4], None)
                e0.record()
                for t in range(steps):
                    self.step_update(rot[t % 4], ph[t % 4], None)
                e1.record()
                for t in range(PSTEPS):  # (then a few steps with the library's timing hook: k_perceive by HIP events)
                    self.set_timing_events([evs.ev[NEV * t + i].value for i in range(NEV)])
                    self.step_update(rot[t % 4], ph[t % 4], None)
                torch.cuda.current_stream(dev).synchronize()
                prc_times.append(float(np.mean([evs.elapsed_ms(NEV * t + 2, NEV * t + 3) for t in range(PSTEPS)])))
                return e0.elapsed_time(e1) / steps
            # the env's own pair first (workspace torch.empty, outputs pieced unless pieced_memory=False), then the other three
            own_ws, own_out = self._ws, self._out_flat
            n_ws, n_out = self.workspace_bytes + 256, self._out_total + 256
            pairs, times = [], []
            for ws_kind in ("own", "other"):
                if ws_kind == "other":
                    self._make_handle(vmm.empty_u8(n_ws, dev))  # (the env's own workspace is torch.empty memory)
                scratch_episode()
                for out_kind in ("own", "other"):
                    if out_kind == "other":
                        self._bind_outputs((torch_u8(n_out) if self._pieced else vmm.empty_u8(n_out, dev)).zero_())
                    elif ws_kind == "other":
                        self._bind_outputs(own_out)
                    pairs.append((self._ws, self._out_flat))
                    times.append(measure())
            labels = ["ws torch / out %s" % ("pieced" if self._pieced else "torch"),
                      "ws torch / out %s" % ("torch" if self._pieced else "pieced"),
                      "ws pieced / out %s" % ("pieced" if self._pieced else "torch"),
                      "ws pieced / out %s" % ("torch" if self._pieced else "pieced")]
            if extra_outputs > 0 and self._pieced:
                self._make_handle(own_ws)
                scratch_episode()
                for k in range(int(extra_outputs)):
                    self._bind_outputs(vmm.empty_u8(n_out, dev).zero_())
                    pairs.append((self._ws, self._out_flat))
                    times.append(measure())
                    labels.append("ws torch / out pieced (draw %d)" % (k + 2))
            # Both levels seen?  If not, walk further into the device's memory until a buffer of another zone turns up.
            spacers = []
            walked = 0
            both_levels = placement_levels_seen
            while self._pieced and walked < int(walk_spacers) and (force_walk or not both_levels(prc_times)):
                # one step of the walk: a spacer from EACH allocator (hipMalloc and the virtual-memory one draw from
                # different ends of the device's memory: profiles/r05/two_colour.txt), then one more output buffer of each
                # kind against the env's own workspace, and the env's own output buffer against one more torch workspace
                free_b = torch.cuda.mem_get_info(dev)[0]
                want = int(min(spacer_gib * 2 ** 30, free_b / 5))
                if want < (4 << 30):
                    break
                try:
                    spacers.append(torch_u8(want))
                    spacers.append(vmm.empty_u8(want, dev))
                except (RuntimeError, _lib.AntsrlError):
                    break
                walked += 1
                depth = sum(x.numel() for x in spacers) / 2 ** 30
                if self._ws.data_ptr() != own_ws.data_ptr():
                    self._make_handle(own_ws)
                scratch_episode()
                for kind, mk in (("pieced", lambda: vmm.empty_u8(n_out, dev)), ("torch", lambda: torch_u8(n_out))):
                    self._bind_outputs(mk().zero_())
                    pairs.append((self._ws, self._out_flat))
                    times.append(measure())
                    labels.append("ws torch / out %s (walk %d: +%.0f GiB)" % (kind, walked, depth))
                self._make_handle(torch_u8(n_ws))
                scratch_episode()
                self._bind_outputs(own_out)
                pairs.append((self._ws, self._out_flat))
                times.append(measure())
                labels.append("ws torch (walk %d: +%.0f GiB) / out %s" % (walked, depth, "pieced" if self._pieced else "torch"))
            # the pair with the fastest observation kernel (the step time around it carries the timing hook's own stalls)
            best = min(range(len(times)), key=prc_times.__getitem__)
            evs.destroy()
            if verbose:
                print("tune_placement: ms/step per (workspace, outputs) pair %s -> %d" % (["%.4f" % t for t in times], best))
            ws, out = pairs[best]
            del pairs, spacers
            if ws.data_ptr() != self._ws.data_ptr():
                self._make_handle(ws)
            self._bind_outputs(out)
            if walked:  # (the walk's spacers and losing buffers: torch's go back to the driver, the pool's are unmapped — their
                vmm.trim()  # ranges retired, never re-used; without a walk the few losing buffers simply stay pooled)
            torch.cuda.empty_cache()
            self._out_flat.zero_()
            del own_ws, own_out, ws, out
        self.placement_trials = dict(ms_per_step=[round(t, 5) for t in times], observation_kernel_ms=[round(t, 5) for t in prc_times],
                                     chosen=best, pairs=labels, both_levels_seen=bool(both_levels(prc_times)), walk_steps=walked)
        self._loaded = None  # (the scratch episodes were the tuner's own: the handle is as new)
        return times

    @property
    def _obs_buf(self):
        """The tensor whose address the kernels get: `obs` itself (a caller may re-point it, like reward / done), or the
        padded buffer `obs` is a view of (obs_row_stride="line")."""
        return self.obs if self.obs_padded is None else self.obs_padded

    def _stream(self):
        if _RAW_STREAM is not None:  # same value as current_stream(device).cuda_stream, without the Stream object
            return C.c_void_p(_RAW_STREAM(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Context that makes self.device current for a launch (a no-op when it already is)."""
        return _NULL_CTX if torch.cuda.current_device() == self._dev_index else torch.cuda.device(self.device)

    def _dev(self, a, dtype, shape=None):
        if a is None:
            return None
        t = torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a)
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if shape is not None:
            assert tuple(t.shape) == tuple(shape), "expected shape %s, got %s" % (shape, tuple(t.shape))
        return t

    # ------------------------------------------------------------------ API
    def reset(self, init: Dict[str, object]) -> None:
        """Loads the initial state (EnvironmentGenerator.generate, environment_generator.py:52-106).
        init: ants_xyt f64[E,N,3], seed f64[E,N], walls u8[E,W,H], food f32[E,W,H],
        anthill_xyr i32[E,3], rocks f64[E,R,4] (if R>0), optional phero f32[E,C,W,H]."""
        c = self.cfg
        E, N, W, H, Cn, R = c.n_envs, c.n_ants, c.w, c.h, c.n_phero, c.n_rocks
        t = dict(
            ants_xyt=self._dev(init["ants_xyt"], torch.float64, (E, N, 3)),
            seed=self._dev(init["seed"], torch.float64, (E, N)),
            walls=self._dev(init["walls"], torch.uint8, (E, W, H)),
            food=self._dev(init["food"], torch.float32, (E, W, H)),
            anthill_xyr=self._dev(init["anthill_xyr"], torch.int32, (E, 3)),
            rocks=self._dev(init.get("rocks"), torch.float64, (E, R, 4)) if R > 0 else None,
            phero=self._dev(init.get("phero"), torch.float32, (E, Cn, W, H)),
        )
        ai = AntsInit(*[None if t[k] is None else t[k].data_ptr()
                        for k in ("ants_xyt", "seed", "walls", "food", "anthill_xyr", "rocks", "phero")])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.antsrl_reset(self._h, C.byref(ai), self._stream()), "reset")
        self._loaded = "reset"
        self._keep = t  # inputs must outlive the enqueued reset kernels

    def generate(self, gen=None, episode_seed: int = 0, walls=None) -> None:
        """Device-side episode generation (antsrl_generate): no host arrays, no upload.  With
        gen.auto_reset, step_update() regenerates all envs after the step that reported done.
        walls (uint8 / bool [E, W, H], for gen.wall_kind == WALLS_INPUT): the bitmaps a walls_generator returned."""
        g = gen if gen is not None else cfgmod.make_gen()
        if walls is not None:
            c = self.cfg
            self._gen_walls = self._dev(np.asarray(walls).astype(np.uint8) if not torch.is_tensor(walls) else walls,
                                        torch.uint8, (c.n_envs, c.w, c.h))
            g.walls_input = self._gen_walls.data_ptr()  # kept alive: auto-reset reads it again
        with torch.cuda.device(self.device):
            _lib.check(self.lib.antsrl_generate(self._h, C.byref(g), int(episode_seed), self._stream()), "generate")
        self._loaded = "given an episode (generate)"

    def _actions(self, rotation, phero):
        c = self.cfg
        if isinstance(rotation, np.ndarray) and isinstance(phero, np.ndarray):
            # host actions: both arrays through one pinned staging buffer, ONE upload
            n = c.n_envs * c.n_ants
            if self._host_act is None:
                self._host_act = torch.empty((2, c.n_envs, c.n_ants), dtype=torch.int8, pin_memory=True)
                self._dev_act = torch.empty((2, c.n_envs, c.n_ants), dtype=torch.int8, device=self.device)
                self._act_event = torch.cuda.Event()
            else:
                self._act_event.synchronize()  # the previous upload has left the staging buffer
            assert rotation.size == n and phero.size == n, "expected %d actions per array" % n
            h = self._host_act.numpy()
            h[0] = self._integral(rotation, "rotation").reshape(c.n_envs, c.n_ants)
            h[1] = self._integral(phero, "pheromone").reshape(c.n_envs, c.n_ants)
            with torch.cuda.device(self.device):
                self._dev_act.copy_(self._host_act, non_blocking=True)
                self._act_event.record()
            return self._dev_act[0], self._dev_act[1]
        rot = self._dev(self._integral(rotation, "rotation"), torch.int8, (c.n_envs, c.n_ants))
        ph = self._dev(self._integral(phero, "pheromone"), torch.int8, (c.n_envs, c.n_ants))
        return rot, ph

    #: check that action arrays hold whole numbers inside int8 before they are cast (one fused device reduction and ONE host
    #: read per non-int8 device tensor and step); set False in a loop whose actions are known good (e.g. int64 from
    #: torch.argmax - 1): no host synchronisation is left then.  int8 tensors are never checked (nothing to check).
    validate_actions = True

    def _integral(self, a, what):
        """Actions are small integers (rotation in {-1, 0, 1} times max_rot_speed, RL_api.py:191; pheromone index in
        {0, 1, 2}, ants.py:90-96) and travel as int8.  ONE rule for numpy arrays, lists and torch tensors (host or
        device): whole numbers of any dtype are accepted, a fractional value or a value outside [-128, 127] raises
        ValueError — nothing is truncated or wrapped."""
        if a is None:
            return None
        if torch.is_tensor(a):
            if a.dtype == torch.int8 or a.numel() == 0:
                return a
            if self.validate_actions:
                fl = a.is_floating_point()
                frac = (a != a.round()).any() if fl else torch.zeros((), dtype=torch.bool, device=a.device)
                lo, hi, bad = torch.stack([a.min().to(torch.float64), a.max().to(torch.float64), frac.to(torch.float64)]).tolist()
                if bad:
                    raise ValueError("%s actions must be whole numbers (the kernels take them as int8)" % what)
                if lo < -128 or hi > 127:
                    raise ValueError("%s actions out of the int8 range" % what)
            return a.to(torch.int8) if a.is_floating_point() else a
        arr = np.asarray(a)
        if arr.dtype.kind == "f":
            if not np.array_equal(arr, np.rint(arr)):
                raise ValueError("%s actions must be whole numbers (the kernels take them as int8)" % what)
            arr = arr.astype(np.int64)
        if arr.size and (arr.min() < -128 or arr.max() > 127):
            raise ValueError("%s actions out of the int8 range" % what)
        return arr

    def outputs_to_host(self, want_obs: bool = True):
        """(obs, agent_state, reward, done) of the last step as fresh numpy arrays: ONE device-to-host copy
        of the packed output buffer into pinned memory (obs left out when not wanted), then host copies."""
        if self.obs.dtype != torch.float32 and want_obs:
            raise _lib.AntsrlError("outputs_to_host: bfloat16 observations have no numpy dtype; use the device tensors")
        base, o = self._out_flat.data_ptr(), self._out_offs
        if any(t.data_ptr() != base + o[k] for k, t in (("obs", self._obs_buf), ("agent_state", self.agent_state),
                                                         ("reward", self.reward), ("done", self.done))):
            # an output was re-pointed (e.g. at a RewardGather slot): plain per-tensor copies
            return (self.obs.cpu().numpy() if want_obs else None, self.agent_state.cpu().numpy(),
                    self.reward.cpu().numpy(), self.done.cpu().numpy())
        if self._host_out is None:
            self._host_out = torch.empty((self._out_flat.numel(),), dtype=torch.uint8, pin_memory=True)
        n = self._out_flat.numel() if want_obs else self._small_bytes
        with torch.cuda.device(self.device):
            self._host_out[:n].copy_(self._out_flat[:n], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        hb, o = self._host_out.numpy(), self._out_offs

        def piece(name, t):
            nb = t.numel() * t.element_size()
            return hb[o[name]:o[name] + nb].view(np.float32 if t.dtype == torch.float32 else np.uint8).reshape(tuple(t.shape)).copy()
        obs_h = None
        if want_obs:
            obs_h = piece("obs", self._obs_buf)
            if self.obs_padded is not None:  # drop the padding on the host
                c = self.cfg
                obs_h = np.ascontiguousarray(obs_h[..., :c.pside * c.pside * c.n_channels]).reshape(tuple(self.obs.shape))
        return (obs_h, piece("agent_state", self.agent_state),
                piece("reward", self.reward), piece("done", self.done))

    def step(self, rotation, phero, want_obs: bool = True):
        """RLApi.step (RL_api.py:168-204) for all envs -> (obs, agent_state, reward, done)."""
        rot, ph = self._actions(rotation, phero)
        with self._on_device():
            _lib.check(self.lib.antsrl_step(self._h, _ptr(rot), _ptr(ph), _ptr(self._obs_buf if want_obs else None),
                                            _ptr(self.agent_state), _ptr(self.reward), _ptr(self.done),
                                            self._stream()), "step")
        return self.obs, self.agent_state, self.reward, self.done

    def observe(self, want_obs: bool = True):
        """RLApi.observation (RL_api.py:96-165) -> (obs, agent_state, reward)."""
        with self._on_device():
            _lib.check(self.lib.antsrl_observe(self._h, _ptr(self._obs_buf if want_obs else None),
                                               _ptr(self.agent_state), _ptr(self.reward), self._stream()),
                       "observe")
        return self.obs, self.agent_state, self.reward

    def update(self, wall_jitter=None) -> None:
        """Environment.update (environment.py:42-47)."""
        c = self.cfg
        j = self._dev(wall_jitter, torch.float64, (c.n_envs, c.n_ants))
        with self._on_device():
            _lib.check(self.lib.antsrl_update(self._h, _ptr(j), self._stream()), "update")

    def update_phase(self, phase: int, wall_jitter=None) -> None:
        """antsrl_update_phase: one of the four reference steps of Environment.update (config.PHASE_WALLS,
        PHASE_ROCKS_PHEROMONE, PHASE_ANTS, PHASE_ANTHILL, in this order) — for callers whose own EnvObjects run between
        the world's objects (rl_api.Environment.update does that)."""
        c = self.cfg
        j = self._dev(wall_jitter, torch.float64, (c.n_envs, c.n_ants)) if phase == cfgmod.PHASE_WALLS else None
        with self._on_device():
            _lib.check(self.lib.antsrl_update_phase(self._h, int(phase), _ptr(j), self._stream()), "update_phase")

    def flush(self) -> None:
        """antsrl_flush: enqueue a deferred Environment.update now (before copying / checkpointing the workspace, or
        to time the update on its own); a no-op when nothing is pending."""
        with self._on_device():
            _lib.check(self.lib.antsrl_flush(self._h, self._stream()), "flush")

    def step_update(self, rotation, phero, wall_jitter=None, want_obs: bool = True):
        """main.py:98 + main.py:131 in one call.  want_obs=False: no observation tensor is written (rewards,
        agent_state, done and — with LinearPolicy.attach — the next actions still are: the act-only rollout)."""
        c = self.cfg
        rot, ph = self._actions(rotation, phero)
        j = self._dev(wall_jitter, torch.float64, (c.n_envs, c.n_ants))
        with self._on_device():
            _lib.check(self.lib.antsrl_step_update(self._h, _ptr(rot), _ptr(ph), _ptr(j),
                                                   _ptr(self._obs_buf if want_obs else None), _ptr(self.agent_state),
                                                   _ptr(self.reward), _ptr(self.done), self._stream()),
                       "step_update")
        return self.obs, self.agent_state, self.reward, self.done

    def set_timing_events(self, events) -> None:
        """events: sequence of config.TIMING_EVENTS raw hipEvent_t handles (ints) or None; see
        antsrl_set_timing_events."""
        if events is None:
            _lib.check(self.lib.antsrl_set_timing_events(self._h, None), "set_timing_events")
        else:
            n = cfgmod.TIMING_EVENTS
            assert len(events) == n, "antsrl_set_timing_events takes %d events" % n
            arr = (C.c_void_p * n)(*[C.c_void_p(int(e)) for e in events])
            _lib.check(self.lib.antsrl_set_timing_events(self._h, arr), "set_timing_events")

    def query(self, what: int) -> int:
        """antsrl_query: what the handle resolved its configuration to (config.Q_*)."""
        v = C.c_longlong()
        _lib.check(self.lib.antsrl_query(self._h, int(what), C.byref(v)), "query")
        return int(v.value)

    def set_activation(self, act, new_deposit_strength: float = 0.0) -> None:
        """Ants.activate_all_pheromones (ants.py:86-87)."""
        c = self.cfg
        a = self._dev(act, torch.float32, (c.n_envs, c.n_ants, c.n_phero))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.antsrl_set_activation(self._h, _ptr(a), float(new_deposit_strength),
                                                      self._stream()), "set_activation")
        self._loaded = "given an activation matrix"

    def perceptive_field(self) -> torch.Tensor:
        """RLApi.perceptive_field (RL_api.py:144-153): bool [E, W, H], the cells some ant of the environment perceives
        (masked cells not counted) — from the positions as they stand: ask right behind step() / observe()."""
        c = self.cfg
        out = torch.empty((c.n_envs, c.w, c.h), dtype=torch.uint8, device=self.device)
        with self._on_device():
            _lib.check(self.lib.antsrl_perceptive_field(self._h, _ptr(out), self._stream()), "perceptive_field")
        return out.to(torch.bool)

    def read_state(self, which: int) -> torch.Tensor:
        c = self.cfg
        E, N, W, H, Cn, R = c.n_envs, c.n_ants, c.w, c.h, c.n_phero, c.n_rocks
        shapes = {
            cfgmod.S_ANTS_XYT: (E, N, 3), cfgmod.S_PREV_XY: (E, N, 2), cfgmod.S_HOLDING: (E, N),
            cfgmod.S_MANDIBLES: (E, N), cfgmod.S_ACTIVATION: (E, N, Cn), cfgmod.S_PHERO: (E, Cn, W, H),
            cfgmod.S_FOOD: (E, W, H), cfgmod.S_EXPLORED: (E, W, H), cfgmod.S_ANTHILL_FOOD: (E,),
            cfgmod.S_ROCK_CENTERS: (E, R, 2), cfgmod.S_TIMESTEP: (E,), cfgmod.S_REWARD_STATE: (E, N),
            cfgmod.S_WALLS: (E, W, H), cfgmod.S_ANTHILL_AREA: (E, W, H), cfgmod.S_SEED: (E, N),
            cfgmod.S_ANTHILL_XYR: (E, 3), cfgmod.S_ROCK_RW: (E, R, 2),
            cfgmod.S_PHERO_C0: (E, W, H), cfgmod.S_PHERO_C1: (E, W, H), cfgmod.S_PHERO_C2: (E, W, H),
            cfgmod.S_PHERO_C3: (E, W, H),
        }
        if cfgmod.S_PHERO_C0 <= which <= cfgmod.S_PHERO_C3 and which - cfgmod.S_PHERO_C0 >= Cn:
            raise _lib.AntsrlError("pheromone channel %d of %d" % (which - cfgmod.S_PHERO_C0, Cn))
        out = torch.empty(shapes[which], dtype=_STATE_DTYPES[which], device=self.device)
        if out.numel():
            with torch.cuda.device(self.device):
                _lib.check(self.lib.antsrl_read_state(self._h, which, _ptr(out), self._stream()), "read_state")
        return out
