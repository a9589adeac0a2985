"""Plugin surface of the reference (muax/nn.py) as torch modules.

`MZNetwork(representation_fn, prediction_fn, dynamic_fn)` keeps the reference's contract:
representation_fn(obs)->s, prediction_fn(s)->(v_logits, pi_logits), dynamic_fn(s, a)->(r_logits, ns)
(muax/model.py:52-54).  Parameters use haiku's Linear layout and default init (w[in][out] ~
TruncatedNormal(1/sqrt(fan_in)), b = 0) so reference checkpoints map one to one.  The default MLP trio
(muax/nn.py:59-115) is what the fused gfx950 kernel evaluates in place; any other torch module goes
through the step-wise search path.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple, Optional

import torch
from torch import nn

HIDDEN = 16  # hk.Linear(16) in every default head (muax/nn.py:77,81,97,101)


class MZNetworkParams(NamedTuple):
    representation: Optional[dict] = None
    prediction: Optional[dict] = None
    dynamic: Optional[dict] = None


class MZNetwork(NamedTuple):
    representation_fn: Callable
    prediction_fn: Callable
    dynamic_fn: Callable


def min_max_normalize(s: torch.Tensor) -> torch.Tensor:
    """muax/nn.py:37-44."""
    s_min = s.min(dim=1, keepdim=True).values
    s_max = s.max(dim=1, keepdim=True).values
    s_scale = s_max - s_min
    s_scale = torch.where(s_scale < 1e-5, s_scale + 1e-5, s_scale)
    return (s - s_min) / s_scale


class HkLinear(nn.Module):
    """hk.Linear: y = x @ w + b with w[in][out]; TruncatedNormal(1/sqrt(in)) / zeros init."""

    def __init__(self, in_features: int, out_features: int, generator: Optional[torch.Generator] = None):
        super().__init__()
        w = torch.empty(in_features, out_features)
        nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=generator)
        self.w = nn.Parameter(w / math.sqrt(in_features))
        self.b = nn.Parameter(torch.zeros(out_features))

    def forward(self, x):
        return x @ self.w + self.b


def _mlp2(i, o, gen):
    return nn.ModuleList([HkLinear(i, HIDDEN, gen), HkLinear(HIDDEN, o, gen)])


def _apply_mlp2(m, x):
    return m[1](torch.nn.functional.elu(m[0](x)))


class Representation(nn.Module):
    """muax/nn.py:59-70: Linear(embedding_dim) + min_max_normalize."""

    def __init__(self, embedding_dim: int, obs_dim: Optional[int] = None, generator=None, name="representation"):
        super().__init__()
        self.embedding_dim, self.obs_dim = embedding_dim, obs_dim
        self.repr_func = HkLinear(obs_dim, embedding_dim, generator) if obs_dim else None
        self._gen = generator

    def materialize(self, obs_dim: int):
        if self.repr_func is None:
            self.obs_dim = obs_dim
            self.repr_func = HkLinear(obs_dim, self.embedding_dim, self._gen)

    def forward(self, obs):
        self.materialize(obs.shape[-1])
        return min_max_normalize(self.repr_func(obs))


class Prediction(nn.Module):
    """muax/nn.py:73-90: value head and policy head, two Linear(16)-elu-Linear stacks."""

    def __init__(self, num_actions: int, full_support_size: int, embedding_dim: Optional[int] = None,
                 generator=None, name="prediction"):
        super().__init__()
        self.num_actions, self.full_support_size, self.embedding_dim = num_actions, full_support_size, embedding_dim
        self._gen = generator
        self.v_func = self.pi_func = None
        if embedding_dim:
            self.materialize(embedding_dim)

    def materialize(self, embedding_dim: int):
        if self.v_func is None:
            self.embedding_dim = embedding_dim
            self.v_func = _mlp2(embedding_dim, self.full_support_size, self._gen)
            self.pi_func = _mlp2(embedding_dim, self.num_actions, self._gen)

    def forward(self, s):
        self.materialize(s.shape[-1])
        return _apply_mlp2(self.v_func, s), _apply_mlp2(self.pi_func, s)


class Dynamic(nn.Module):
    """muax/nn.py:93-115: reward head and next-state head on [s, onehot(a)]."""

    def __init__(self, embedding_dim: int, num_actions: int, full_support_size: int, generator=None,
                 name="dynamic"):
        super().__init__()
        self.embedding_dim, self.num_actions, self.full_support_size = embedding_dim, num_actions, full_support_size
        # creation order follows the reference: ns_func first, then r_func
        self.ns_func = _mlp2(embedding_dim + num_actions, embedding_dim, generator)
        self.r_func = _mlp2(embedding_dim + num_actions, full_support_size, generator)

    def forward(self, s, a):
        sa = torch.cat([s, torch.nn.functional.one_hot(a.long(), self.num_actions).to(s.dtype)], dim=1)
        return _apply_mlp2(self.r_func, sa), min_max_normalize(_apply_mlp2(self.ns_func, sa))


def _init_representation_func(representation_module, embedding_dim):
    """muax/nn.py:417-421: returns the representation_fn callable."""
    return representation_module(embedding_dim)


def _init_prediction_func(prediction_module, num_actions, full_support_size):
    """muax/nn.py:423-427."""
    return prediction_module(num_actions, full_support_size)


def _init_dynamic_func(dynamic_module, embedding_dim, num_actions, full_support_size):
    """muax/nn.py:429-433."""
    return dynamic_module(embedding_dim, num_actions, full_support_size)


def create_muzero_network(representation_module, prediction_module, dynamic_module, embedding_dim: int,
                          num_actions: int, full_support_size: int) -> MZNetwork:
    """muax/nn.py:23-34."""
    return MZNetwork(_init_representation_func(representation_module, embedding_dim),
                     _init_prediction_func(prediction_module, num_actions, full_support_size),
                     _init_dynamic_func(dynamic_module, embedding_dim, num_actions, full_support_size))


def is_default_mlp_trio(network: MZNetwork) -> bool:
    r, p, d = network
    return type(r) is Representation and type(p) is Prediction and type(d) is Dynamic


def mlp_trio_weights(network: MZNetwork) -> dict:
    """The 18 arrays of the default trio in the C-ABI's order/layout (include/mzsearch.h)."""
    r, p, d = network
    return {"repr_w": r.repr_func.w, "repr_b": r.repr_func.b,
            "pv_w1": p.v_func[0].w, "pv_b1": p.v_func[0].b, "pv_w2": p.v_func[1].w, "pv_b2": p.v_func[1].b,
            "pp_w1": p.pi_func[0].w, "pp_b1": p.pi_func[0].b, "pp_w2": p.pi_func[1].w, "pp_b2": p.pi_func[1].b,
            "dr_w1": d.r_func[0].w, "dr_b1": d.r_func[0].b, "dr_w2": d.r_func[1].w, "dr_b2": d.r_func[1].b,
            "dn_w1": d.ns_func[0].w, "dn_b1": d.ns_func[0].b, "dn_w2": d.ns_func[1].w, "dn_b2": d.ns_func[1].b}
