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


# ---------------------------------------------------------------------------------------------------
# The small helpers of muax/utils.py that sit beside the codec (muax/utils.py:37-68,104-223), on NumPy / torch
# ---------------------------------------------------------------------------------------------------
class sliceable_deque(__import__("collections").deque):
    """collections.deque that answers slices with another sliceable_deque (muax/utils.py:37-46)."""

    def __getitem__(self, index):
        if isinstance(index, slice):
            from itertools import islice
            return type(self)(islice(self, index.start, index.stop, index.step))
        return super().__getitem__(index)


def min_max(state, _min: float, _max: float):
    """muax/utils.py:61-63: affine map of [_min, _max] onto [0, 1]."""
    return (state - _min) / (_max - _min)


def diff_transform_matrix(num_frames: int, dtype="float32"):
    """muax/utils.py:104-146: M such that X @ M turns `num_frames` stacked frames (oldest first on the last axis)
    into discrete derivatives -- the last column is the newest frame, the one before it the first difference, ...
    Column c (from the right, c = 0 .. n-1) holds the c-th finite-difference stencil (-1)^k C(c, k) on the c + 1
    newest frames."""
    import numpy as np
    from math import comb
    assert isinstance(num_frames, int) and num_frames >= 1
    n = num_frames
    m = np.zeros((n, n), dtype=np.float64)
    for c in range(n):           # order of the derivative
        for k in range(c + 1):   # k frames back from the newest
            m[n - 1 - k, n - 1 - c] = (-1) ** k * comb(c, k)
    return m.astype(dtype)


def diff_transform(X, dtype="float32"):
    """muax/utils.py:149-170: X @ diff_transform_matrix(X.shape[-1])."""
    import numpy as np
    M = diff_transform_matrix(int(X.shape[-1]), dtype=dtype)
    if isinstance(X, torch.Tensor):
        return X @ torch.as_tensor(M, dtype=X.dtype, device=X.device)
    return np.dot(X, M)


def n_step_bootstrapped_returns(r_t, discount_t, v_t, n: int, lambda_t=1.0, stop_target_gradients: bool = False):
    """muax/utils.py:173-219 (rlax's strided n-step return): G_t = r_{t+1} + g_{t+1} [(1 - l_{t+1}) v_{t+1} + l_{t+1} G_{t+1}]
    iterated n times over a length-T sequence, bootstrapping from v shifted by n (the tail repeats the last value).
    NumPy arrays in, NumPy array [T] out."""
    import numpy as np
    r_t, discount_t, v_t = (np.asarray(x, dtype=np.float64) for x in (r_t, discount_t, v_t))
    T = r_t.shape[0]
    lam = np.ones_like(discount_t) * lambda_t
    pad = min(n - 1, T)
    targets = np.concatenate([v_t[n - 1:], np.full(pad, v_t[-1])])
    r_p = np.concatenate([r_t, np.zeros(n - 1)])
    d_p = np.concatenate([discount_t, np.ones(n - 1)])
    l_p = np.concatenate([lam, np.ones(n - 1)])
    v_p = np.concatenate([v_t, np.full(n - 1, v_t[-1])])
    for i in reversed(range(n)):
        targets = r_p[i:i + T] + d_p[i:i + T] * ((1.0 - l_p[i:i + T]) * v_p[i:i + T] + l_p[i:i + T] * targets)
    return targets.astype(np.float32)


def action2plane(action, shape):
    """muax/utils.py:222-223: the action broadcast to a plane of `shape` (the ResNet dynamics' action channel)."""
    import numpy as np
    if isinstance(action, torch.Tensor):
        return action.expand(shape)
    return np.broadcast_to(action, shape)


_SIGNAL_POOL_KEEPALIVE = []


def warm_runtime(n_events: int = 2048, device=None):
    """Grow the HIP runtime's completion-signal pool once, ahead of time (no reference counterpart: a property of the
    ROCm runtime this path runs on).  Every synchronised launch takes a signal from a pool that grows on demand, and a
    growth step stalls the host: tools/diag_stall.py shows ONE act of 50-70 ms among thousands of 0.118 ms around the
    1100th synchronised launch of a process (and a 1-3 ms one around the 420th).  Recording `n_events` throwaway events
    (kept alive for the process) makes the pool grow here instead: no step above 1 ms in 4000 afterwards.  bench.py and
    the timing tools call it before anything is timed; a latency-sensitive caller of act() may do the same at start-up."""
    if _SIGNAL_POOL_KEEPALIVE or not torch.cuda.is_available():
        return
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        for _ in range(n_events):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _SIGNAL_POOL_KEEPALIVE.append(e)
        torch.cuda.synchronize()
