"""Host-side JAX PRNG key arithmetic (threefry2x32, non-partitionable stream) in NumPy.

The reference drives everything from `jax.random.PRNGKey` / `jax.random.split`
(muax/train.py:139-154: one split per environment step).  The search kernels draw mctx's own
streams from the key on the device; the host only needs key bookkeeping, which is this module.
Restated from jax/_src/prng.py (threefry_2x32, threefry_split, threefry_random_bits).
"""
from __future__ import annotations

import numpy as np

_R = ((13, 15, 26, 6), (17, 29, 16, 24))
_M = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M


def threefry2x32(key, x0, x1):
    """Hash counter words (x0, x1) under `key`; arrays broadcast. Returns (y0, y1) uint32."""
    k0, k1 = np.uint64(int(key[0])), np.uint64(int(key[1]))
    ks = (k0, k1, k0 ^ k1 ^ np.uint64(0x1BD11BDA))
    x0 = (np.asarray(x0, np.uint64) + ks[0]) & _M
    x1 = (np.asarray(x1, np.uint64) + ks[1]) & _M
    for g in range(5):
        for r in _R[g & 1]:
            x0 = (x0 + x1) & _M
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M
        x1 = (x1 + ks[(g + 2) % 3] + np.uint64(g + 1)) & _M
    return x0.astype(np.uint32), x1.astype(np.uint32)


def _threefry_int(k0: int, k1: int, x0: int, x1: int):
    """One block on Python ints: an order of magnitude faster than NumPy for the two or three blocks of a
    key split (the per-environment-step cost of muax.fit's loop)."""
    M = 0xFFFFFFFF
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0 = (x0 + k0) & M
    x1 = (x1 + k1) & M
    for g in range(5):
        for r in _R[g & 1]:
            x0 = (x0 + x1) & M
            x1 = (((x1 << r) | (x1 >> (32 - r))) & M) ^ x0
        x0 = (x0 + ks[(g + 1) % 3]) & M
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & M
    return x0, x1


def random_bits(key, size: int) -> np.ndarray:
    """threefry_2x32(key, iota(size)): odd sizes are zero padded, the pad's output dropped."""
    half = (size + 1) // 2
    if half <= 8:
        k0, k1 = int(key[0]), int(key[1])
        out = [_threefry_int(k0, k1, i, half + i if half + i < size else 0) for i in range(half)]
        return np.array([o[0] for o in out] + [o[1] for o in out], np.uint32)[:size]
    x0 = np.arange(half, dtype=np.uint64)
    x1 = x0 + np.uint64(half)
    x1 = np.where(x1 < size, x1, 0)
    y0, y1 = threefry2x32(key, x0, x1)
    return np.concatenate([y0, y1])[:size]


def PRNGKey(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) for the default threefry implementation."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split(key, num) -> uint32[num, 2]."""
    return random_bits(np.asarray(key, np.uint32), 2 * num).reshape(num, 2)


def uniform(key, size: int) -> np.ndarray:
    """jax.random.uniform(key, (size,), float32) in [0, 1)."""
    bits = random_bits(np.asarray(key, np.uint32), size)
    return ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def as_key(key) -> np.ndarray:
    """Accept an int seed or two uint32 words."""
    if isinstance(key, (int, np.integer)):
        return PRNGKey(int(key))
    try:
        import torch
        if isinstance(key, torch.Tensor):
            key = key.detach().cpu().numpy()
    except ImportError:
        pass
    k = np.asarray(key).astype(np.uint64).ravel()
    if k.size != 2:
        raise ValueError("rng_key must be an int seed or two uint32 words")
    return (k & 0xFFFFFFFF).astype(np.uint32)
