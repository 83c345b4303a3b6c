"""Device-resident replay memory: agents/replay_memory.py:6-114 with the rolling
arrays kept as torch tensors ON THE GPU, so `agent.update_replay_memory(obs, agent_state, action, reward,
new_obs, new_agent_state, done)` (main.py:102) costs no PCIe round trip per step.

Same constructor, `extend`, `random_access`, `__len__`, `__getitem__` as the reference's
`ReplayMemory`.  Differences, both deliberate:
  * inputs may be torch tensors already on the device (numpy is accepted and uploaded);
  * a batch that crosses the end of the arrays wraps correctly.  The reference's recursive call
    (replay_memory.py:113-114) re-stacks an already stacked action array and mis-measures it, so it only
    works when batches never straddle `max_len`; this class implements the documented intent
    ("when head reaches the maximum length of arrays, it cycles back", replay_memory.py:7-9).
`random_access` draws its indices on the device (torch.randint, with replacement) instead of
`random.sample` on the host.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


class DeviceReplayMemory:
    def __init__(self, max_len: int, observation_space: Sequence[int], agent_space: Sequence[int],
                 action_space: Sequence[int], device="cuda"):
        self.max_len = max_len
        self.observation_space, self.agent_space, self.action_space = observation_space, agent_space, action_space
        self.device = torch.device(device)
        z = lambda shape, dt: torch.zeros([max_len] + list(shape), dtype=dt, device=self.device)  # noqa: E731
        self.states = z(observation_space, torch.float32)          # replay_memory.py:18-24
        self.agent_states = z(agent_space, torch.float32)
        self.actions = z(action_space, torch.int64)
        self.rewards = z([], torch.float32)
        self.new_states = z(observation_space, torch.float32)
        self.new_agent_states = z(agent_space, torch.float32)
        self.dones = z([], torch.bool)
        self.head = 0   # :36
        self.fill = 0   # :39

    def __len__(self):
        return self.fill

    def _t(self, a, dtype):
        if a is None:
            return None
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype)

    def __getitem__(self, idx):
        idx = idx if torch.is_tensor(idx) else torch.as_tensor(idx, device=self.device)
        idx = idx.to(self.device)
        return (self.states[idx], self.agent_states[idx], self.actions[idx], self.rewards[idx],
                self.new_states[idx], self.new_agent_states[idx], self.dones[idx])

    def random_access(self, n: int, generator: Optional[torch.Generator] = None):
        idx = torch.randint(0, self.fill, (n,), device=self.device, generator=generator)
        return self[idx]

    def extend(self, states, agent_states, actions, rewards, new_states, new_agent_states, done):
        """replay_memory.py:83-114.  `actions` = (rotation[n], pheromone[n] or None); `done`: bool, a per-entry
        bool array [n], or a per-environment one [E] (n = E * ants: repeated over each environment's ants)."""
        rot = self._t(actions[0], torch.int64).reshape(-1)
        n = rot.shape[0]
        ph = self._t(actions[1], torch.int64).reshape(-1) if actions[1] is not None else torch.ones_like(rot)  # :100-103
        act = torch.stack((rot, ph), dim=-1)
        st = self._t(states, torch.float32).reshape([n] + list(self.observation_space))
        ast = self._t(agent_states, torch.float32).reshape([n] + list(self.agent_space))
        rw = self._t(rewards, torch.float32).reshape(n)
        nst = self._t(new_states, torch.float32).reshape([n] + list(self.observation_space))
        nast = self._t(new_agent_states, torch.float32).reshape([n] + list(self.agent_space))
        if torch.is_tensor(done) or isinstance(done, np.ndarray):
            dn = self._t(done, torch.bool).reshape(-1)
            if dn.numel() == 1:
                dn = dn.expand(n)
            elif dn.numel() != n:  # one flag per environment: repeated over that environment's ants
                if n % dn.numel() != 0:
                    raise ValueError("done has %d entries for %d transitions" % (dn.numel(), n))
                dn = dn.repeat_interleave(n // dn.numel())
        else:
            dn = torch.full((n,), bool(done), dtype=torch.bool, device=self.device)
        if n > self.max_len:  # only the newest max_len entries can survive
            cut = n - self.max_len
            st, ast, act, rw, nst, nast, dn = (x[cut:] for x in (st, ast, act, rw, nst, nast, dn))
            self.head = (self.head + cut) % self.max_len
            n = self.max_len
        first = min(self.max_len - self.head, n)  # :97
        for dst, src in ((self.states, st), (self.agent_states, ast), (self.actions, act), (self.rewards, rw),
                         (self.new_states, nst), (self.new_agent_states, nast), (self.dones, dn)):
            dst[self.head:self.head + first] = src[:first]
            if first < n:  # wrap to the beginning (documented intent of :113-114)
                dst[: n - first] = src[first:]
        self.fill = min(self.max_len, max(self.fill, self.head + n))  # :109
        self.head = (self.head + n) % self.max_len                       # :112
