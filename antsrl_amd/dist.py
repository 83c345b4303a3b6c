"""Multi-GPU sharding of the environment batch.

Environments are fully independent (every array is created per generate() call,
generator/environment_generator.py:57-99), so the batch is split into contiguous blocks, one
process per GPU, all state resident on its GPU for the whole episode: no halo, no migration, no
data-path collective.  The ONLY exchange is the all-gather of reward[E_local, N] (fp32) and
done[E_local] (u8) each step, so that every rank / the host sees the full batch.

Backend "nccl" is RCCL on ROCm (over xGMI inside a node); "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_envs_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global env ids owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_envs_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RewardGather:
    """Pre-allocated all-gather of (reward, done).  Equal shards use all_gather_into_tensor (one
    fused buffer per tensor, no per-rank list); ragged shards fall back to padded gathers."""

    def __init__(self, n_envs_total: int, n_ants: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.E, self.N = n_envs_total, n_ants
        self.ranges = [shard_range(n_envs_total, r, self.world) for r in range(self.world)]
        self.max_local = max(hi - lo for lo, hi in self.ranges)
        self.equal = all(hi - lo == self.max_local for lo, hi in self.ranges)
        self.reward_all = torch.empty((self.world * self.max_local, n_ants), dtype=torch.float32, device=device)
        self.done_all = torch.empty((self.world * self.max_local,), dtype=torch.uint8, device=device)
        if not self.equal:
            self._pad_r = torch.zeros((self.max_local, n_ants), dtype=torch.float32, device=device)
            self._pad_d = torch.zeros((self.max_local,), dtype=torch.uint8, device=device)

    def __call__(self, reward_local: torch.Tensor, done_local: torch.Tensor):
        """-> (reward [E_total, N], done [E_total]) on every rank (views of internal buffers)."""
        lo, hi = self.ranges[self.rank]
        assert reward_local.shape == (hi - lo, self.N) and done_local.shape == (hi - lo,)
        if self.equal:
            dist.all_gather_into_tensor(self.reward_all, reward_local.contiguous(), group=self.group)
            dist.all_gather_into_tensor(self.done_all, done_local.contiguous(), group=self.group)
            return self.reward_all, self.done_all
        self._pad_r[: hi - lo] = reward_local
        self._pad_d[: hi - lo] = done_local
        dist.all_gather_into_tensor(self.reward_all, self._pad_r, group=self.group)
        dist.all_gather_into_tensor(self.done_all, self._pad_d, group=self.group)
        keep = torch.cat([torch.arange(r * self.max_local, r * self.max_local + (h - l))
                          for r, (l, h) in enumerate(self.ranges)]).to(self.reward_all.device)
        return self.reward_all[keep], self.done_all[keep]
