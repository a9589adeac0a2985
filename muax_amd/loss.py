"""k-step unrolled MuZero loss of the reference (muax/loss.py:10-88) on torch autograd.

INTERIM implementation of SURVEY.md 8(f) n1: the arithmetic is plain PyTorch ops on the GPU (autograd
for the backward pass), not yet a hand-written fused fwd/bwd HIP kernel -- that kernel is the next-round
item.  What is pinned here is the reference's formula:

    loss = sum_{i<L} [ mean_B CE(r_logits_i, support(r_i)) + mean_B CE(v_logits_i, support(Rn_i))
                       + mean_B CE(pi_logits_i, pi_i) ]  +  1e-4 * 0.5 * sum ||param||^2

with the hidden state's gradient halved at the start of every dynamics step (Appendix G,
muax/loss.py:60-61).  Reference quirks handled explicitly:
  * muax/loss.py:83 reads `loss` before assignment; the only consistent reading is an initial 0;
  * the coax variant divides by L (muax/frameworks/coax/loss.py:70-71): `divide_by_length=True`;
  * muax stores pi per step as [1, A] (muax/model.py:176), so batch.pi[:, i] is [B, 1, A] against logits
    [B, A] and optax broadcasts to an all-pairs [B, B] cross entropy.  Here pi is squeezed to [B, L, A]
    (the intended per-sample target); `pi_all_pairs=True` reproduces the broadcast.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch

from . import utils as mx_utils


@dataclass
class Transition:
    """muax/episode_tracer.py:41-56, batched: every field is [B, L, ...]."""
    obs: Any = 0.
    a: Any = 0
    r: Any = 0.
    done: Any = False
    Rn: Any = 0.
    v: Any = 0.
    pi: Any = 0.
    w: Any = 1.


def softmax_cross_entropy(logits, labels):
    """optax.softmax_cross_entropy: -sum(labels * log_softmax(logits), -1)."""
    return -(labels * torch.log_softmax(logits, dim=-1)).sum(dim=-1)


def default_loss_fn(muzero_instance, batch: Transition, divide_by_length: bool = False,
                    pi_all_pairs: bool = False):
    """muax/loss.py:10-88.  `batch` fields are tensors/arrays of shape [B, L, ...]."""
    dev = muzero_instance.device
    t = lambda x, dt=torch.float32: torch.as_tensor(x, dtype=dt, device=dev)  # noqa: E731
    a = t(batch.a, torch.long)
    B, L = a.shape[:2]
    a = a.reshape(B, L)
    S = muzero_instance._support_size
    r_t = mx_utils.scalar_to_support(t(batch.r).reshape(B, L), S).detach()
    Rn_t = mx_utils.scalar_to_support(t(batch.Rn).reshape(B, L), S).detach()
    pi = t(batch.pi)
    pi = pi.reshape(B, L, 1, -1) if pi_all_pairs else pi.reshape(B, L, -1)
    obs = t(batch.obs)
    s = muzero_instance.repr_func(obs[:, 0])
    loss = torch.zeros((), device=dev)
    for i in range(L):
        v, logits = muzero_instance.pred_func(s)
        s = mx_utils.scale_gradient(s, 0.5)  # Appendix G
        r, ns = muzero_instance.dy_func(s, a[:, i])
        loss = loss + softmax_cross_entropy(r, r_t[:, i]).mean() \
            + softmax_cross_entropy(v, Rn_t[:, i]).mean() \
            + softmax_cross_entropy(logits, pi[:, i].detach()).mean()
        s = ns
    if divide_by_length:
        loss = loss / L
    l2 = 0.5 * sum((p ** 2).sum() for m in muzero_instance.network if isinstance(m, torch.nn.Module)
                   for p in m.parameters())
    return loss + 1e-4 * l2
