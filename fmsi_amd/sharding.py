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


def comm_from_process_group(device_index: int, group=None):
    """An fmsi_hip_comm over the ranks of a torch.distributed group: rank 0 draws the RCCL unique id and the group's
    own channel carries its 128 bytes to the others (any launcher-side channel would do; include/fmsi_hip.h)."""
    import torch.distributed as dist

    import fmsi_amd as fm

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [fm.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return fm.Comm(box[0], world, rank, device_index)


def gather_to_root_abi(comm, local, n_items: int, dst: int = 0, stream: int = 0):
    """gather_to_root() through the C-ABI (fmsi_hip_gather): `local` is this rank's 1-D device tensor of
    shard_sizes(n_items)[rank] elements; returns the concatenation on `dst` (rank order = output order), None elsewhere.
    No padding: every peer's bytes go straight to their place."""
    import torch

    sizes = shard_sizes(n_items, comm.world)
    assert local.is_cuda and local.is_contiguous() and local.numel() == sizes[comm.rank], (local.numel(), sizes[comm.rank])
    item = local.element_size()
    out = torch.empty(n_items if comm.rank == dst else 0, dtype=local.dtype, device=local.device)
    comm.gather(local.data_ptr() if local.numel() else 0, out.data_ptr() if out.numel() else 0, [s * item for s in sizes], root=dst,
                stream=stream)
    return out if comm.rank == dst else None


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
