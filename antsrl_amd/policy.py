"""In-loop policy inference on the GPU (BASELINE config 5).

`LinearPolicy` holds the weights of the reference's DQN nets — `ExploreModel`
(agents/explore_agent_pytorch.py:24-45: layer1 Linear(F+2 -> 32), layer2 Linear(32 -> 3)) plus the
pheromone head `CollectModel.layer3` (agents/collect_agent.py:24-51) — and evaluates them on the
observation tensor with the bf16 MFMA kernel `antsrl_policy_mlp` (no torch matmul, no copy of the
observation): rotation = argmax(layer2(out)) - 1, pheromone = argmax(layer3(out)), as
agents/collect_agent_memory.py:196-199 does on the host.

The reference's checkpoints (agents/models/*.h5) are stale against its current code (150-wide
inputs), so weights are random-initialised like nn.Linear, or loaded from any
state_dict with layer1/layer2/(layer3) of the right shapes.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib


class LinearPolicy:
    def __init__(self, n_features: int, device, with_pheromone_head: bool = True, seed: int = 0):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)

        def linear(out_f, in_f):  # nn.Linear's default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)))
            b = 1.0 / math.sqrt(in_f)
            w = (torch.rand((out_f, in_f), generator=g) * 2 - 1) * b
            bias = (torch.rand((out_f,), generator=g) * 2 - 1) * b
            return w.to(device).contiguous(), bias.to(device).contiguous()

        self.n_features = n_features
        self.device = torch.device(device)
        self.w1, self.b1 = linear(32, n_features + 2)
        self.w2, self.b2 = linear(3, 32)
        self.w3, self.b3 = linear(3, 32) if with_pheromone_head else (None, None)
        self._lib = _lib.load()
        self._rot = self._ph = None

    def load_state_dict(self, sd) -> None:
        """Accepts the reference's parameter names: layer1 / layer2 / layer3 .weight / .bias as `ExploreModel.state_dict()`
        has them (explore_agent_pytorch.py:36-37) or behind the `explore_model.` prefix of `CollectModel.state_dict()`
        (collect_agent.py:28,44)."""
        sd = {(k[len("explore_model."):] if k.startswith("explore_model.") else k): v for k, v in sd.items()}
        for name, attr in (("layer1", "1"), ("layer2", "2"), ("layer3", "3")):
            if name + ".weight" in sd:
                w = torch.as_tensor(sd[name + ".weight"]).to(self.device, torch.float32).contiguous()
                b = torch.as_tensor(sd[name + ".bias"]).to(self.device, torch.float32).contiguous()
                assert w.shape == getattr(self, "w" + attr).shape, "%s: %s" % (name, tuple(w.shape))
                setattr(self, "w" + attr, w)
                setattr(self, "b" + attr, b)

    def act(self, obs: torch.Tensor, agent_state: torch.Tensor, logits: Optional[torch.Tensor] = None, env=None):
        """obs [..., P, P, K] (float32, or bfloat16 from a BatchedAntsEnv(obs_dtype=torch.bfloat16) passed as
        `env`) and agent_state float32 [..., 2] on the GPU -> (rotation int8 [...], pheromone int8 [...] or
        None), ready to pass to step()."""
        lead = obs.shape[:-3]
        m = 1
        for d in lead:
            m *= d
        assert obs.is_contiguous() and agent_state.is_contiguous()
        if obs.dtype == torch.bfloat16:
            assert env is not None and env.obs.dtype == torch.bfloat16, "bfloat16 observations: pass the env that produced them"
        else:
            assert obs.dtype == torch.float32 and (env is None or env.obs.dtype == torch.float32)
        handle = env._h if (env is not None and obs.dtype == torch.bfloat16) else None
        assert obs.numel() == m * self.n_features and agent_state.numel() == m * 2
        if self._rot is None or self._rot.numel() != m:
            self._rot = torch.empty((m,), dtype=torch.int8, device=self.device)
            self._ph = torch.empty((m,), dtype=torch.int8, device=self.device) if self.w3 is not None else None

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._lib.antsrl_policy_mlp(handle, p(obs), p(agent_state), m, self.n_features, p(self.w1),
                                                   p(self.b1), p(self.w2), p(self.b2), p(self.w3), p(self.b3),
                                                   p(self._rot), p(self._ph), p(logits), st), "policy_mlp")
        rot = self._rot.view(lead)
        return rot, (self._ph.view(lead) if self._ph is not None else None)

    def attach(self, env) -> None:
        """In-loop form (antsrl_set_inloop_policy): from now on every observation `env` writes also leaves the net's
        actions for the NEXT step in `env.next_rotation` / `env.next_pheromone` (int8 [E, N]) — the values act() returns
        for that observation, computed inside the observation kernel.  Needs bfloat16 observations on the cell-meta
        path; detach() switches it off."""
        assert env.obs.dtype == torch.bfloat16, "the in-loop policy reads bfloat16 observation rows"
        E, N = env.cfg.n_envs, env.cfg.n_ants
        env.next_rotation = torch.zeros((E, N), dtype=torch.int8, device=self.device)
        env.next_pheromone = torch.zeros((E, N), dtype=torch.int8, device=self.device) if self.w3 is not None else None

        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._lib.antsrl_set_inloop_policy(env._h, self.n_features, p(self.w1), p(self.b1), p(self.w2), p(self.b2),
                                                          p(self.w3), p(self.b3), p(env.next_rotation), p(env.next_pheromone), st),
                       "set_inloop_policy")
        env._loaded = "given an in-loop policy"

    def detach(self, env) -> None:
        _lib.check(self._lib.antsrl_set_inloop_policy(env._h, self.n_features, None, None, None, None, None, None, None, None, None),
                   "set_inloop_policy")
