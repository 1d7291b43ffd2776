"""Multi-GPU plumbing of the query path: one process per GPU, index replicated, query windows sharded by
contiguous, order-preserving ranges, no collective on the data path, one gather of the per-rank result
buffers to rank 0 (rank order = the reference's output order).  SURVEY.md 8(e).

torch.distributed is plumbing here (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests); the compute of a shard is whatever callable the caller passes (the C-ABI query on a GPU rank).
"""
from __future__ import annotations

from typing import Callable

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range of rank `rank`: sizes differ by at most one, earlier ranks get the extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> list[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_to_root(local, n_items: int, dst: int = 0, group=None):
    """Gather per-rank result tensors (1-D, same dtype, rank r holds shard_sizes(n_items)[r] elements) to `dst`
    and return their concatenation there (None elsewhere).  Shards are padded to the largest size because
    the collective wants equal shapes; sizes differ by at most one element."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_items, world)
    assert local.numel() == sizes[rank], (local.numel(), sizes[rank])
    width = max(sizes) if sizes else 0
    padded = local
    if local.numel() != width:
        padded = torch.zeros(width, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def sharded_query(kmers: np.ndarray, compute: Callable[[np.ndarray], np.ndarray], device=None, dst: int = 0, group=None):
    """Every rank holds (or can produce) the same global k-mer array; rank r answers its contiguous range with
    `compute` and the byte results are gathered to `dst` in rank order.  Returns the full result array on
    `dst`, None elsewhere.  With world size 1 (or no process group) it is just `compute(kmers)`."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(compute(kmers))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a, b = shard_range(len(kmers), rank, world)
    local = np.ascontiguousarray(compute(kmers[a:b]))
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    out = gather_to_root(t, len(kmers), dst=dst, group=group)
    return None if out is None else out.cpu().numpy()
