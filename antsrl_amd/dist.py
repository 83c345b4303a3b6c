"""Multi-GPU sharding of the environment batch.

Environments are fully independent (every array is created per generate() call,
generator/environment_generator.py:57-99), so the batch is split into contiguous blocks, one
process per GPU, all state resident on its GPU for the whole episode: no halo, no migration, no
data-path collective.  The ONLY exchange is the all-gather of reward[E_local, N] (fp32) and
done[E_local] (u8) each step, so that every rank / the host sees the full batch.

Backend "nccl" is RCCL on ROCm (over xGMI inside a node); "gloo" is used by the CPU tests and by the one-GPU rehearsal of
world > 1 (tests/test_a_gpu_dist.py).  Set TORCH_NCCL_HIGH_PRIORITY=1 before init_process_group (bench.py does): on RCCL's
default-priority stream the per-step gather shares a hardware queue with the step's kernels.  What one collective per step
costs a rank, measured with ONE RCCL rank at c3 (0.198 ms/step without it; profiles/r05/dist_overhead_ab.txt): +24 us on the
high-priority stream, +36-50 us on the default one — not host time (67 us of enqueue per step, dist_host_time.txt), not the
SDMA engine, not k_perceive's occupancy (…_sdma_ab.txt, …_occupancy_ab.txt): the cross-stream event hand-overs around the
collective (an event record between two kernels costs the stream ~15 us of idle time on this runtime): a blocking
collective on the step's own stream (mode "inline" below) costs +8…+12 us with one rank, but puts the collective's latency
inside the step (dist_overhead_ab_inline.txt).  (Round 4 read +4 us off a 0.240 ms step.)

The gather is ONE collective per step: reward and done travel in a single fused fp32 buffer
[E_local, N+1] (done in the last column — xGMI all-gathers of this size are latency-bound, so one
2 MiB message beats two), and it can run asynchronously under the next step's kernels
(`start` / `finish`): the payload is copied out of the live reward tensor first, because the next
step overwrites it.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_envs_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global env ids owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_envs_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_cfg(n_envs_total: int, rank: int, world: int, n_ants: int, w: int, h: int, **make_cfg_kwargs):
    """AntsCfg of `rank`'s block of a batch of n_envs_total environments: n_envs = the block's size, env_id_base = its
    first GLOBAL environment id, n_envs_total = the whole batch — so every environment-keyed random stream of the
    library (wall jitter, device generators, auto-reset seeds) is the one the unsharded batch would use and sharding
    cannot change a result (include/antsrl.h, AntsCfg.env_id_base).  -> (cfg, lo, hi)."""
    from .config import make_cfg
    lo, hi = shard_range(n_envs_total, rank, world)
    return make_cfg(hi - lo, n_ants, w, h, env_id_base=lo, n_envs_total=n_envs_total, **make_cfg_kwargs), lo, hi


class _Works:
    """The works of one grouped exchange behind the one-collective interface (wait())."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        while self.works:  # (once each: a second wait() on a gloo send / receive blocks for ever)
            self.works.pop(0).wait()


class RewardGather:
    """Pre-allocated all-gather of (reward, done) across the ranks that share an env batch.

    algo "collective" (default): ONE `all_gather_into_tensor`; RCCL chooses the algorithm (its rings / trees over the links it
    found: `NCCL_ALGO`, `NCCL_PROTO` are the knobs).
    algo "direct": SURVEY.md §8(e)'s one-hop form, spelled out: every rank sends its block to each of the other world - 1 ranks
    and receives theirs, as ONE grouped batch of point-to-point operations (`batch_isend_irecv`: on RCCL a single
    ncclGroupStart / End launch), so that on a fully connected xGMI node every block rides the link between its two ranks and
    nothing is forwarded; the rank's own block is a device copy.  Same result, same buffers, same start / finish; opt-in
    (`bench.py --gather-algo direct`) until both have been timed on a node with more than one GPU — the gloo tests run it with
    two and three ranks."""

    def __init__(self, n_envs_total: int, n_ants: int, device, group=None, algo: str = "collective"):
        if algo not in ("collective", "direct"):
            raise ValueError("algo must be 'collective' or 'direct'")
        if algo == "direct" and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo":
            raise ValueError("RewardGather(algo='direct') on device tensors needs the RCCL ('nccl') backend: gloo's send / "
                             "receive take host memory")
        self.algo = algo
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.E, self.N = n_envs_total, n_ants
        self.ranges = [shard_range(n_envs_total, r, self.world) for r in range(self.world)]
        self.max_local = max(hi - lo for lo, hi in self.ranges)
        self.equal = all(hi - lo == self.max_local for lo, hi in self.ranges)
        # fused payload: [max_local, N+1] per rank, column N carries `done`
        # TWO staging slots used alternately (round 5): the gather of step t reads slot t % 2 while step t + 1 fills the other,
        # so the step's stream never waits for the collective it has just launched — only for the one before it, which has
        # had a whole step to finish.  (With one slot every step paid the collective's full latency: +31 us per 0.198 ms
        # step with one RCCL rank, profiles/r05/dist_overhead_ab.txt.)
        self._sends = [torch.zeros((self.max_local, n_ants + 1), dtype=torch.float32, device=device) for _ in range(2)]
        self._recvs = [torch.empty((self.world * self.max_local, n_ants + 1), dtype=torch.float32, device=device) for _ in range(2)]
        self._works = [None, None]
        self._slot = 0          # the slot of the most recent start()
        self._send, self._recv = self._sends[0], self._recvs[0]
        self._work = None
        if not self.equal:
            self._keep = torch.cat([torch.arange(r * self.max_local, r * self.max_local + (h - l))
                                    for r, (l, h) in enumerate(self.ranges)]).to(device)

    def _exchange(self, recv: torch.Tensor, send: torch.Tensor):
        """Launches the exchange of one slot (recv = world equal blocks along dim 0, block r from rank r) without blocking the
        current stream; -> something with wait()."""
        if self.algo == "collective" or self.world == 1:
            return dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        n = recv.shape[0] // self.world
        peer = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
        ops = []
        for d in range(1, self.world):  # rank + d and rank - d: at every distance the whole node forms disjoint pairs of links
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            ops.append(dist.P2POp(dist.isend, send, peer(to), self.group))
            ops.append(dist.P2POp(dist.irecv, recv[frm * n:(frm + 1) * n], peer(frm), self.group))
        works = dist.batch_isend_irecv(ops)
        recv[self.rank * n:(self.rank + 1) * n].copy_(send)  # (the rank's own block: on the caller's stream, like its readers)
        return _Works(works)

    def start(self, reward_local: torch.Tensor, done_local: torch.Tensor) -> None:
        """Snapshot this rank's (reward, done) and launch the all-gather without blocking the
        current stream; the caller may enqueue the next step right away."""
        lo, hi = self.ranges[self.rank]
        assert reward_local.shape == (hi - lo, self.N) and done_local.shape == (hi - lo,)
        k = self._slot ^ 1
        if self._works[k] is not None:  # the slot is reused: the gather that last read it (two steps ago) must have consumed it
            self._works[k].wait()
            self._works[k] = None
        self._slot = k
        self._send, self._recv = self._sends[k], self._recvs[k]
        self._send[: hi - lo, : self.N].copy_(reward_local)
        self._send[: hi - lo, self.N].copy_(done_local)
        self._work = self._works[k] = self._exchange(self._recv, self._send)

    def finish(self):
        """Wait for the gather launched by start(); -> (reward [E_total, N], done [E_total] uint8) on
        every rank, or None if nothing is pending."""
        if self._work is None:
            return None
        for k in (self._slot ^ 1, self._slot):  # (the older gather first: both are done when this returns)
            if self._works[k] is not None:
                self._works[k].wait()
                self._works[k] = None
        self._work = None
        return self.result()

    def result(self):
        full = self._recv if self.equal else self._recv[self._keep]
        return full[:, : self.N], full[:, self.N].to(torch.uint8)

    # ------------------------------------------------------------------ zero-copy form
    # The step kernels can write reward / done straight into the gather's send buffer: two byte buffers
    # [E_local*N*4 reward | E_local done] used alternately (the gather of step t reads slot t % 2 while
    # step t + 1 writes the other one), so no staging copy is enqueued at all.
    # EXPERIMENTAL until it has run on RCCL with more than one rank (the driver's 8-GPU node): its ordering rests
    # on ProcessGroupNCCL's stream semantics — the collective waits for everything enqueued on the CURRENT stream
    # when it is called, and Work.wait() makes the current stream wait for the collective.  The step kernels must
    # therefore be enqueued on the stream that is current at outputs() / start_slot() time: recorded and checked.
    @staticmethod
    def _current_stream(device):
        return torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else None

    def outputs(self, slot: int):
        """(reward float32 [E_local, N], done uint8 [E_local]) views of send slot `slot` (0 or 1): bind them
        as the env's output tensors for the step whose results start_slot(slot) will gather."""
        if not hasattr(self, "_zc"):
            lo, hi = self.ranges[self.rank]
            n_loc = hi - lo
            # one rank's payload, padded to 16 bytes: every rank's block of the gathered buffer then starts on an
            # aligned address (a float32 view of a block needs 4; found by tests/test_a_gpu_dist.py with 7 envs per rank)
            nb = (self.max_local * self.N * 4 + self.max_local + 15) // 16 * 16
            dev = self._send.device
            self._zc = dict(n_loc=n_loc, nb=nb,
                            send=[torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(2)],
                            recv=[torch.empty(self.world * nb, dtype=torch.uint8, device=dev) for _ in range(2)],
                            work=[None, None])
        z = self._zc
        if z["work"][slot] is not None:
            z["work"][slot].wait()  # the gather that last read this slot is ordered before the kernels that rewrite it
        z.setdefault("stream", [None, None])[slot] = self._current_stream(self._send.device)
        sb = z["send"][slot]
        rew = sb[: self.max_local * self.N * 4].view(torch.float32).view(self.max_local, self.N)[: z["n_loc"]]
        done = sb[self.max_local * self.N * 4: self.max_local * self.N * 4 + self.max_local][: z["n_loc"]]
        return rew, done

    def start_slot(self, slot: int) -> None:
        """All-gather send slot `slot` (written by the step just enqueued) without blocking the stream."""
        z = self._zc
        if z["work"][slot] is not None:
            z["work"][slot].wait()
        if z.get("stream") and z["stream"][slot] != self._current_stream(self._send.device):
            raise RuntimeError("RewardGather.start_slot: the current stream changed since outputs(%d): the step kernels "
                               "and the gather would not be ordered" % slot)
        z["work"][slot] = self._exchange(z["recv"][slot], z["send"][slot])

    def gather_slot_inline(self, slot: int) -> None:
        """All-gather send slot `slot` as a BLOCKING collective (async_op=False): ProcessGroupNCCL then enqueues it on the
        current stream, behind the step's kernels and ahead of the next step's — no side stream, no event hand-over, and the
        collective's whole latency inside the step.  finish_slot(slot) reads the result."""
        z = self._zc
        dist.all_gather_into_tensor(z["recv"][slot], z["send"][slot], group=self.group, async_op=False)
        z["inline"] = True

    def finish_slot(self, slot: int):
        """-> (reward [E_total, N], done [E_total] uint8) of the gather started on `slot`, or None."""
        z = getattr(self, "_zc", None)
        if z is None or (z["work"][slot] is None and not z.get("inline")):
            return None
        if z["work"][slot] is not None:
            z["work"][slot].wait()
            z["work"][slot] = None
        rb = z["recv"][slot].view(self.world, z["nb"])
        rew = rb[:, : self.max_local * self.N * 4].view(torch.float32).view(self.world, self.max_local, self.N)  # (a view: rows are 16-byte aligned)
        done = rb[:, self.max_local * self.N * 4: self.max_local * self.N * 4 + self.max_local]
        if self.equal:
            return rew.reshape(self.world * self.max_local, self.N), done.reshape(-1)
        keep = self._keep
        return rew.reshape(self.world * self.max_local, self.N)[keep], done.reshape(-1)[keep]

    def __call__(self, reward_local: torch.Tensor, done_local: torch.Tensor):
        """Blocking form: -> (reward [E_total, N], done [E_total]) on every rank."""
        self.start(reward_local, done_local)
        return self.finish()


class ShardedStepper:
    """The per-step sequence of a sharded run — what `bench.py --gpus N` executes on every rank, factored out so
    that the CPU (gloo, world size 2) test drives the very same code with a stand-in for the device step.

        stepper = ShardedStepper(env, gather, mode)          # env: anything with .reward / .done tensors
        for t in range(K): stepper.step(t, lambda: env.step_update(...))
        reward_all, done_all = stepper.drain()               # last step's gathered batch (None without a gather)

    mode "staged" (default): the step writes the env's own reward / done tensors, start() snapshots them into the
    send buffer (two small device copies) and launches the all-gather asynchronously under the next step.
    mode "zero_copy": the env's reward / done are re-pointed at the gather's alternating send slots, so the kernels
    write the collective's payload in place (experimental, see RewardGather.outputs).
    mode "inline": zero_copy's in-place payload on ONE slot, gathered by a blocking collective on the step's own stream
    (RewardGather.gather_slot_inline): nothing overlaps, nothing is handed between streams."""

    def __init__(self, env, gather: Optional[RewardGather], mode: str = "staged"):
        if mode not in ("staged", "zero_copy", "inline"):
            raise ValueError("mode must be 'staged', 'zero_copy' or 'inline'")
        self.env, self.gather, self.mode = env, gather, mode
        self._last_slot = None

    def step(self, t: int, do_step) -> None:
        g = self.gather
        if g is not None and self.mode == "zero_copy":
            self.env.reward, self.env.done = g.outputs(t % 2)
        elif g is not None and self.mode == "inline" and self._last_slot is None:
            self.env.reward, self.env.done = g.outputs(0)
        do_step()
        if g is None:
            return
        if self.mode == "inline":
            g.gather_slot_inline(0)
            self._last_slot = 0
            return
        # the path's only exchange: the reward / done all-gather, one fused collective per
        # step, left running under the next step's kernels
        if self.mode == "zero_copy":
            g.start_slot(t % 2)
            self._last_slot = t % 2
        else:
            g.start(self.env.reward, self.env.done)

    def drain(self):
        """Waits for every gather still in flight; -> the most recent (reward_all, done_all) or None."""
        g = self.gather
        if g is None:
            return None
        if self.mode == "inline":
            return g.finish_slot(0) if self._last_slot is not None else None
        if self.mode == "zero_copy":
            last = self._last_slot
            out = None
            for s in (0, 1):
                r = g.finish_slot(s)
                if s == last:
                    out = r
            return out
        return g.finish()
