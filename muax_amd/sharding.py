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
