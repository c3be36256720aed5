"""Multi-GPU sharding of the forward path: one process per GPU, batch sharded by rank, ONE collective.

Samples are independent in the eval-mode forward, so the data path has no exchange at
all: every rank runs its contiguous shard of the batch through the same replicated weights.  The only
collective is the eval collation — an all-gather of the per-rank predictions (or, cheaper, of the
per-sample IoU scalars) — which is RCCL over xGMI with backend "nccl" on the GPU box and gloo on CPU
(tests/test_collate_cpu.py runs it with world_size 2).

The reference itself evidences no multi-GPU path (its run command is a single process,
/root/reference/README.md:85,91); BASELINE.json configs[4] asks for this one.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` samples owned by `rank`: the first total % world ranks get one
    extra sample (ragged batches never drop or duplicate a sample)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} world={world} rank={rank}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


def all_gather_ragged(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Concatenate every rank's `local` (its shard_bounds slice, sample axis first) in rank order.

    Equal shards use one all_gather_into_tensor (a single RCCL all-gather; 33.5 MB per rank for 256
    voxel grids).  Ragged shards pad to the largest shard so the collective stays a single fixed-size
    all-gather, and the padding is cut after it."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise RuntimeError(f"rank holds {local.shape[0]} samples, its shard is {sizes[dist.get_rank(group)]}")
    big = max(sizes)
    tail = tuple(local.shape[1:])
    if big == 0:
        return local.new_empty((0,) + tail)
    send = local
    if local.shape[0] != big:
        send = local.new_zeros((big,) + tail)
        send[: local.shape[0]] = local
    out = local.new_empty((world * big,) + tail)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    if all(s == big for s in sizes):
        return out
    return torch.cat([out[r * big: r * big + sizes[r]] for r in range(world)], 0)


def sharded_forward(forward: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], left: torch.Tensor,
                    right: torch.Tensor, group=None, gather: bool = True) -> Optional[torch.Tensor]:
    """Run `forward` on this rank's shard of a globally known batch and (optionally) collate.

    `left` / `right` hold the WHOLE batch on every rank (e.g. a replicated eval list); each rank slices
    its shard, so the result is identical to forward(left, right) on one device, in the same order."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = left.shape[0]
    b, e = shard_bounds(total, world, rank)
    local = forward(left[b:e], right[b:e])
    return all_gather_ragged(local, total, group) if gather else local
