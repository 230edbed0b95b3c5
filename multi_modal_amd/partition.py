"""Batch partitioner: shards independent clips over the GPUs of one node, one process per GPU.

The reference is single-GPU (`torch.cuda.set_device(args.cuda)`, classification/vggsound/vanilla_multi.py:113)
and builds its spectrogram batches with DataLoader workers (vanilla_multi.py:76-77).  Clips are
independent, so the front-end shards by contiguous batch ranges with no halo and no reduction; the only
collective is an all-gather (RCCL over xGMI when the tensors are on GPUs, gloo on CPU in tests) for callers
that need the whole feature tensor on every rank.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of rank `rank`: ceil(n/world) items per rank, the tail ranks may get fewer
    (possibly none)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    per = -(-n_items // world_size)
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi


def shard_size(n_items: int, world_size: int) -> int:
    """Common (padded) shard size used by the collective."""
    return -(-n_items // world_size)


def all_gather_features(local: torch.Tensor, n_items: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Reassemble the (n_items, ...) feature tensor on every rank from the per-rank shards produced over
    shard_range().  Shards are padded to the common shard size for one all_gather_into_tensor call and the
    padding is sliced off afterwards."""
    if not dist.is_initialized():
        if local.shape[0] != n_items:
            raise ValueError("single process: the local shard must be the whole batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = shard_size(n_items, world)
    lo, hi = shard_range(n_items, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} items, expected {hi - lo}")
    tail = tuple(local.shape[1:])
    if local.shape[0] != per:
        padded = local.new_zeros((per,) + tail)
        padded[: local.shape[0]] = local
        local = padded
    local = local.contiguous()
    full_shape = (per * world,) + tail
    if out is None or tuple(out.shape) != full_shape:
        out = local.new_empty(full_shape)
    dist.all_gather_into_tensor(out, local, group=group)
    return out[:n_items]


def sharded_apply(fn: Callable[[torch.Tensor], torch.Tensor], batch: torch.Tensor, gather: bool = True, group=None) -> torch.Tensor:
    """Run `fn` (a batched front-end such as preprocessing_audio_batch) on this rank's contiguous shard of
    `batch` and, if `gather`, return the whole feature tensor on every rank."""
    n = batch.shape[0]
    if not dist.is_initialized():
        return fn(batch)
    lo, hi = shard_range(n, dist.get_rank(group), dist.get_world_size(group))
    local = fn(batch[lo:hi])
    return all_gather_features(local, n, group) if gather else local
