"""Categorical value codec of the reference (muax/utils.py:65-102) on torch tensors."""
from __future__ import annotations

import torch


def _scaling(x, eps: float = 1e-3):
    """muax/utils.py:65-67 (https://arxiv.org/abs/1805.11593)."""
    return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x


def _inv_scaling(x, eps: float = 1e-3):
    """muax/utils.py:70-76."""
    return torch.sign(x) * (((torch.sqrt(1 + 4 * eps * (torch.abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1)


def scalar_to_support(x, support_size: int):
    """muax/utils.py:79-91: two-hot encoding of the scaled scalar on 2*support_size+1 bins."""
    x = torch.clamp(_scaling(x), -support_size, support_size)
    low = torch.floor(x).long()
    high = torch.ceil(x).long()
    prob_high = x - low
    prob_low = 1.0 - prob_high
    n = 2 * support_size + 1
    oh = torch.nn.functional.one_hot
    return oh(low + support_size, n) * prob_low[..., None] + oh(high + support_size, n) * prob_high[..., None]


def support_to_scalar(probs, support_size: int):
    """muax/utils.py:94-102."""
    bins = torch.arange(-support_size, support_size + 1, dtype=probs.dtype, device=probs.device)
    return _inv_scaling((bins * probs).sum(dim=-1))


def scale_gradient(g, scale: float = 1.0):
    """muax/utils.py:55-57: same forward value, gradient scaled."""
    return g * scale + g.detach() * (1.0 - scale)
