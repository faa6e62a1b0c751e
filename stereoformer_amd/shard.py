"""Multi-GPU sharding of stereo pairs: one process per GPU, pairs split over ranks, no
collective on the data path (every hot-path op is independent per pair -- SURVEY 8(e)).
RCCL/gloo (`torch.distributed`) carries only result gathering and timing reductions.

Reference counterpart: ``nn.DataParallel`` scatter/gather in ``train_file/trainer_LC.py:108-112``.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def pair_slice(total_pairs: int, rank: int, world: int) -> slice:
    """Contiguous, balanced shard of ``total_pairs`` for ``rank`` (the first ``total % world``
    ranks take one extra pair; empty shards are legal when total < world)."""
    if world <= 0 or not (0 <= rank < world) or total_pairs < 0:
        raise ValueError(f"bad shard request: total={total_pairs} rank={rank} world={world}")
    base, extra = divmod(total_pairs, world)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def shard_sizes(total_pairs: int, world: int) -> list[int]:
    return [pair_slice(total_pairs, r, world).stop - pair_slice(total_pairs, r, world).start for r in range(world)]


def gather_pairs(local: torch.Tensor, total_pairs: int, group=None) -> torch.Tensor:
    """all_gather ragged per-rank result shards ``[n_r, ...]`` back into ``[total_pairs, ...]`` on
    every rank (direct all-gather; shards are padded to the largest shard for the collective)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(total_pairs, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"rank {dist.get_rank(group)} holds {local.shape[0]} pairs, expected {sizes[dist.get_rank(group)]}")
    mx = max(sizes)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":  # gloo has no device all_gather: stage through the host
        local = local.cpu()
    padded = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded.contiguous(), group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], 0).to(dev)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """Whole-job time = the slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=None if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value: float, device=None, group=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=None if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


def all_ranks(value: float, device=None, group=None) -> list[float]:
    """Every rank's scalar, in rank order, on every rank (per-rank step times beside the max)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    world = dist.get_world_size(group)
    t = torch.zeros(world, dtype=torch.float64, device=None if dist.get_backend(group) == "gloo" else device)
    t[dist.get_rank(group)] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in t.tolist()]
