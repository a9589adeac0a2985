"""Sharding of a batch of independent roots over the GPUs of a node.

Roots never interact, so the act path needs no collective: rank g owns a contiguous slice of the
global batch and tells the kernels its (global_batch, root_offset) so that every per-root PRNG stream
is the one the un-sharded batch would have used -- results do not depend on the number of GPUs.
torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only to gather outputs.
"""
from __future__ import annotations

import torch


def shard_roots(global_batch: int, world_size: int, rank: int):
    """Contiguous split; the first (global_batch % world_size) ranks take one extra root."""
    base, extra = divmod(global_batch, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def gather_roots(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """all_gather per-root outputs ([B_local, ...]) back into global root order on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = [shard_roots(global_batch, world, r)[1] for r in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def allreduce_mean_flat(tensors, group=None, even_if_alone: bool = False):
    """Data-parallel gradient mean (the only collective of the whole project; reference precedent:
    jax.lax.pmean in muax/frameworks/acme/jax/muzero/learning.py:151).  All tensors are packed into ONE
    flat fp32 buffer -> one all-reduce (RCCL over xGMI on GPUs; a few KB to MB, latency-bound, so one
    message instead of a ring of small ones) -> scaled by 1/world -> unpacked in place.  A world of one returns
    at once unless `even_if_alone` (tests: the RCCL call itself on a 1-GPU box)."""
    import torch.distributed as dist
    tensors = [t for t in tensors if t is not None]
    if not tensors or not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(group) == 1 and not even_if_alone:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].reshape(t.shape))
        off += n
