"""Batch partitioner: shards independent clips over the GPUs of one node, one process per GPU.

The reference is single-GPU (`torch.cuda.set_device(args.cuda)`, classification/vggsound/vanilla_multi.py:113)
and builds its spectrogram batches with DataLoader workers (vanilla_multi.py:76-77).  Clips are
independent, so the front-end shards by contiguous batch ranges with no halo and no reduction; the only
exchange is an all-gather (RCCL over xGMI when the tensors are on GPUs, gloo on CPU in tests) for callers
that need the whole feature tensor on every rank, in two forms: the library's all-gather collective, or every
shard sent point-to-point to every peer (one xGMI link per peer, all links busy at once).
"""
from __future__ import annotations

import os
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


GATHER_MODES = ("auto", "collective", "direct")


def pick_gather_mode(mode: str, world: int, even: bool, on_gpu: bool) -> str:
    """What "auto" resolves to.  Each MI355X has one xGMI link to each of its 7 peers (about 153 GB/s each), so the bytes a
    rank must receive, (world - 1) shards, can arrive over world - 1 links at once if every rank sends its shard straight to
    every peer; a ring moves the same bytes through ONE incoming link per rank.  Which of the two an all-gather collective
    runs is the library's choice at run time, so:
      direct      GPU tensors and 3 or more ranks (the links are the bound, use all of them), or shards of unequal size
                  (every shard lands in its final slot, no padded copy of a shard is made)
      collective  2 ranks (one peer, one link either way), or CPU tensors with equal shards (tests)
    MMFE_ALLGATHER_MODE=collective|direct overrides "auto" (measure both with `bench.py --config vgg200k`)."""
    if mode not in GATHER_MODES:
        raise ValueError(f"mode must be one of {GATHER_MODES}, got {mode!r}")
    if mode == "auto":
        env = os.environ.get("MMFE_ALLGATHER_MODE", "")
        if env in ("collective", "direct"):
            return env
        return "direct" if (not even or (on_gpu and world >= 3)) else "collective"
    return mode


def _gather_collective(local: torch.Tensor, n_items: int, per: int, world: int, group, out: Optional[torch.Tensor]) -> torch.Tensor:
    """One all_gather_into_tensor.  The collective needs equal contributions: a rank whose shard is short (the tail ranks
    of an uneven split) contributes a zero-padded copy of it, and the padding at the end of the gathered tensor is sliced off."""
    tail = tuple(local.shape[1:])
    if local.shape[0] != per:
        padded = local.new_zeros((per,) + tail)
        padded[: local.shape[0]] = local
        local = padded
    full_shape = (per * world,) + tail
    if out is None or tuple(out.shape) != full_shape or not out.is_contiguous():
        out = local.new_empty(full_shape)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_items]


def _gather_direct(local: torch.Tensor, n_items: int, rank: int, world: int, group, out: Optional[torch.Tensor]) -> torch.Tensor:
    """Every rank sends its shard to every peer and receives every peer's shard straight into its final slot of the output
    (one batch of point-to-point operations: grouped send / receive pairs under RCCL, one xGMI link per peer).  No padding,
    no staging copy; empty shards are skipped on both sides (both sides know every shard's range)."""
    tail = tuple(local.shape[1:])
    if out is None or tuple(out.shape[1:]) != tail or out.shape[0] < n_items or not out.is_contiguous():
        out = local.new_empty((n_items,) + tail)
    local = local.contiguous()
    lo, hi = shard_range(n_items, rank, world)
    ops = []
    for step in range(1, world):                 # peers in rotating order: at step s every rank sends to rank + s and receives from rank - s
        dst = (rank + step) % world
        src = (rank - step) % world
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, local, dst if group is None else dist.get_global_rank(group, dst), group))
        slo, shi = shard_range(n_items, src, world)
        if shi > slo:
            ops.append(dist.P2POp(dist.irecv, out[slo:shi], src if group is None else dist.get_global_rank(group, src), group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if hi > lo:
        out[lo:hi].copy_(local)
    for r in reqs:
        r.wait()
    return out[:n_items]


def all_gather_features(local: torch.Tensor, n_items: int, group=None, out: Optional[torch.Tensor] = None, mode: str = "auto") -> torch.Tensor:
    """Reassemble the (n_items, ...) feature tensor on every rank from the per-rank shards produced over shard_range().
    mode "collective": one all_gather_into_tensor (short tail shards are zero-padded to the common size first);
    mode "direct": every shard sent point-to-point into its final slot on every peer; "auto": see pick_gather_mode.
    Both give the same tensor.  `out` is used when it fits (collective: (shard_size * world, ...); direct: (n_items, ...))."""
    if mode not in GATHER_MODES:
        raise ValueError(f"mode must be one of {GATHER_MODES}, got {mode!r}")
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
    chosen = pick_gather_mode(mode, world, per * world == n_items, local.is_cuda)
    if chosen == "direct":
        return _gather_direct(local, n_items, rank, world, group, out)
    return _gather_collective(local, n_items, per, world, group, out)


def _gather_chunked(fn, local_in: torch.Tensor, n_items: int, rank: int, world: int, chunks: int, group) -> torch.Tensor:
    """Compute and exchange in `chunks` pieces: while piece c of every shard travels (asynchronous point-to-point operations into the
    final slots of the output, on the communication library's own stream), the transform of piece c + 1 runs.  Every rank cuts every
    shard at the same places (piece c of a shard of s items is [c * ceil(s / chunks), ...)), so both sides of every transfer know its
    size without a handshake.  The exchange hides at most the compute, which is the smaller part."""
    lo, hi = shard_range(n_items, rank, world)

    def piece(r: int, c: int) -> Tuple[int, int]:
        rlo, rhi = shard_range(n_items, r, world)
        step = -(-(rhi - rlo) // chunks) if rhi > rlo else 0
        a = min(rlo + c * step, rhi)
        return a, min(a + step, rhi)

    out = None
    pending = []
    for c in range(chunks):
        a, b = piece(rank, c)
        y = fn(local_in[a - lo:b - lo]) if b > a else None
        if out is None:
            probe = y if y is not None else fn(local_in[:0])
            out = probe.new_empty((n_items,) + tuple(probe.shape[1:]))
        if y is not None:
            out[a:b].copy_(y)
        ops = []
        for step in range(1, world):
            dst, src = (rank + step) % world, (rank - step) % world
            if b > a:
                ops.append(dist.P2POp(dist.isend, out[a:b], dst if group is None else dist.get_global_rank(group, dst), group))
            sa, sb = piece(src, c)
            if sb > sa:
                ops.append(dist.P2POp(dist.irecv, out[sa:sb], src if group is None else dist.get_global_rank(group, src), group))
        if ops:
            pending += dist.batch_isend_irecv(ops)
    for r in pending:
        r.wait()
    return out


def sharded_apply(fn: Callable[[torch.Tensor], torch.Tensor], batch: torch.Tensor, gather: bool = True, group=None,
                  mode: str = "auto", chunks: int = 1) -> torch.Tensor:
    """Run `fn` (a batched front-end such as preprocessing_audio_batch) on this rank's contiguous shard of
    `batch` and, if `gather`, return the whole feature tensor on every rank (`mode`: see all_gather_features).
    chunks > 1: the shard is transformed and exchanged in that many pieces, the exchange of one piece under the transform of the
    next (point-to-point form; `mode` is ignored)."""
    n = batch.shape[0]
    if not dist.is_initialized():
        return fn(batch)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n, rank, world)
    if gather and chunks > 1 and world > 1:
        return _gather_chunked(fn, batch[lo:hi], n, rank, world, int(chunks), group)
    local = fn(batch[lo:hi])
    return all_gather_features(local, n, group, mode=mode) if gather else local
