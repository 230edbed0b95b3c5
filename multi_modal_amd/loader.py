"""Host-to-device staging in front of the batched transforms.

The reference computes its features inside `Dataset.__getitem__` on DataLoader workers and moves finished spectrograms to
the GPU in the training loop (`batch[k].to(device)`, classification/vggsound/vanilla_multi.py:76-77, 89-92).  With the
transform on the GPU the workers return raw waveforms (float32, or the int16 PCM the files hold) and the collated batch
has to cross PCIe before the kernel can start.  `DevicePrefetcher` wraps any iterable of collated batches and keeps that
copy off the critical path: batch i + 1 is copied from a pinned staging buffer on a copy stream while the caller's stream
runs the transform and the model on batch i.

    for feats, labels in DevicePrefetcher(loader, transform=lambda b: (mm.preprocessing_audio_batch(b[0]), b[1])):
        ...

PyTorch is plumbing here (pinned memory, streams, events); the arithmetic stays in libmmfe.so.
"""
from __future__ import annotations

from typing import Any, Callable, Iterable, Iterator, List, Optional

import torch

from .plans import require_gpu


def _map_tensors(obj: Any, fn: Callable[[torch.Tensor], torch.Tensor]) -> Any:
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    return obj


class _Slot:
    """One staging set: pinned host buffers + device buffers, reused for every batch of the same shapes, and the event that
    marks the end of its host-to-device copies."""

    def __init__(self) -> None:
        self.pinned: List[torch.Tensor] = []
        self.device: List[torch.Tensor] = []
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.used = False


class DevicePrefetcher:
    """Iterates over `batches` (collated CPU batches: tensors, or lists / tuples / dicts of tensors), yielding each batch with
    its tensors on `device` - after `transform(batch)` when one is given - while the next batch's copies are in flight.

    depth staging sets rotate.  A staging set is reused only after the work that consumed it on the caller's stream has finished
    (an event recorded when the following batch is requested, waited for on the host before the set's pinned buffers are
    refilled), so the device buffers a `transform` reads are never overwritten under it.  With the default of 3 that wait is for
    work two batches back, so the GPU always has the previous batch's work queued while the host stages; depth 2 saves a
    staging set but leaves the GPU idle while the host refills the pinned buffers.
    Tensors a batch would still share with a staging set when it is handed out (labels a transform passes through; every
    tensor when there is no transform) are copied out first, so yielded batches never change afterwards.  Non-tensor leaves pass through untouched.  Tensors that are already on the device are passed through without a copy."""

    def __init__(self, batches: Iterable[Any], device: Optional[torch.device] = None, transform: Optional[Callable[[Any], Any]] = None,
                 depth: int = 3) -> None:
        require_gpu()
        if depth < 2:
            raise ValueError("depth must be at least 2 (one batch in use, one in flight)")
        self.batches = batches
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.transform = transform
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = [_Slot() for _ in range(depth)]

    def _issue(self, slot: _Slot, batch: Any) -> Any:
        """Stage `batch` into the slot's pinned buffers and start the copies on the copy stream."""
        if slot.used:
            slot.consumed.synchronize()          # host-side wait: the pinned buffers are about to be overwritten by the CPU
        idx = 0

        def stage(t: torch.Tensor) -> torch.Tensor:
            nonlocal idx
            if t.is_cuda:
                return t
            i = idx
            idx += 1
            if i >= len(slot.pinned) or slot.pinned[i].shape != t.shape or slot.pinned[i].dtype != t.dtype:
                pinned = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                # Allocated ON the copy stream: the caching allocator hands out blocks per stream, so a block the caller's
                # stream freed a moment ago (a temporary that kernels still queued there read) cannot come back here and be
                # overwritten by the copy.  The caller's stream reads the buffer later; record_stream tells the allocator, so
                # the block is not recycled for the copy stream while those reads are pending once the set is replaced.
                with torch.cuda.stream(self.copy_stream):
                    dev = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                dev.record_stream(torch.cuda.current_stream(self.device))
                if i >= len(slot.pinned):
                    slot.pinned.append(pinned)
                    slot.device.append(dev)
                else:
                    slot.pinned[i], slot.device[i] = pinned, dev
            slot.pinned[i].copy_(t)
            with torch.cuda.stream(self.copy_stream):
                slot.device[i].copy_(slot.pinned[i], non_blocking=True)
            return slot.device[i]

        staged = _map_tensors(batch, stage)
        slot.ready.record(self.copy_stream)
        slot.used = True
        return staged

    def __iter__(self) -> Iterator[Any]:
        it = iter(self.batches)
        pending = []                              # (slot, staged batch), oldest first
        n = 0

        def issue_next() -> bool:
            nonlocal n
            try:
                batch = next(it)
            except StopIteration:
                return False
            slot = self.slots[n % self.depth]
            pending.append((slot, self._issue(slot, batch)))
            n += 1
            return True

        # depth - 2 batches are staged ahead (at least one): when batch k is handed out, the sets in use are those of batch k - 1
        # (its work is running), of batches k .. k + depth - 3 (staged) and the one being refilled, which last held batch k - 2
        more = True
        while more and len(pending) < max(1, self.depth - 2):
            more = issue_next()
        previous: Optional[_Slot] = None
        while pending:
            slot, staged = pending.pop(0)
            stream = torch.cuda.current_stream(self.device)
            if previous is not None:
                previous.consumed.record(stream)   # everything the caller queued on the previous batch is ahead of this point
            if more:
                more = issue_next()                # refills the set that held batch k - 2 (depth >= 3): its work is a whole batch
                                                   # behind what the GPU is running now, so the host wait inside is short
            stream.wait_event(slot.ready)          # device-side wait: the caller's stream starts after this batch's copies
            out = self.transform(staged) if self.transform is not None else staged
            # whatever still aliases a staging buffer (labels a transform passes through, or everything without a transform) is
            # copied out on the caller's stream, so a yielded batch stays valid however long the caller keeps it
            owned = {d.untyped_storage().data_ptr() for d in slot.device}
            out = _map_tensors(out, lambda t: t.clone() if t.is_cuda and t.untyped_storage().data_ptr() in owned else t)
            previous = slot
            yield out
        if previous is not None:
            previous.consumed.record(torch.cuda.current_stream(self.device))
