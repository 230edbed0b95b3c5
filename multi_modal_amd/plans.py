"""Plan cache and the one place where torch tensors meet the C ABI.

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic happens in
libmmfe.so.  Inputs may live on the CPU (they are copied to the current GPU and the result is copied
back, like the reference's per-sample CPU call) or on a GPU (result stays there).
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from typing import Callable, Dict, Hashable, Tuple

import torch

from . import _native as N

OUT_DTYPES = {N.OUT_F32: torch.float32, N.OUT_F16: torch.float16, N.OUT_BF16: torch.bfloat16}
OUT_CODES = {v: k for k, v in OUT_DTYPES.items()}
IN_DTYPES = {N.IN_F32: torch.float32, N.IN_S16: torch.int16, N.IN_F64: torch.float64, N.IN_F16: torch.float16, N.IN_BF16: torch.bfloat16}
IN_CODES = {v: k for k, v in IN_DTYPES.items()}

_plans: Dict[Hashable, N.Plan] = {}
_plans_lock = threading.Lock()
_last_kernel = "none"


def last_kernel() -> str:
    """Kernel family ("fbank512", "rfft400", "dftgen", ...) that served the most recent run_rows call (diagnostic)."""
    return _last_kernel


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise N.MmfeError("multi_modal_amd needs a ROCm GPU (gfx950): torch.cuda.is_available() is False and "
                          "there is no CPU fallback")


# Plans whose key holds caller data (a frame_signal window, a stream handle) form bounded families: the least recently
# used plan of a full family is dropped from the cache; its device tables are freed when the last reference to it goes (a
# thread that is launching it at that moment still holds one, and hipFree waits for kernels already queued).
_family_limits: Dict[str, int] = {}
_family_order: Dict[str, "OrderedDict[Hashable, None]"] = {}


def cached_plan(key: Hashable, device: int, build: Callable[[int], N.Plan], family: str = None, limit: int = 0) -> N.Plan:
    k = (key, device)
    evicted = []      # released after the lock is dropped: freeing device tables synchronises with the device
    with _plans_lock:
        plan = _plans.get(k)
        if plan is None:
            plan = build(device)
            _plans[k] = plan
        if family is not None and limit > 0:
            order = _family_order.setdefault(family, OrderedDict())
            order[k] = None
            order.move_to_end(k)
            while len(order) > limit:
                old, _ = order.popitem(last=False)
                evicted.append(_plans.pop(old, None))
    del evicted
    return plan


def family_size(family: str) -> int:
    with _plans_lock:
        return len(_family_order.get(family, ()))


def clear_plans() -> None:
    with _plans_lock:
        for p in _plans.values():
            p.close()
        _plans.clear()
        _family_order.clear()


def to_device_rows(x: torch.Tensor, pcm_ok: bool = False, keep_float_dtype: bool = False) -> Tuple[torch.Tensor, torch.device]:
    """(rows, n) float32 tensor with unit inner stride on a GPU, plus the device the caller's data was on.
    With pcm_ok an int16 tensor stays int16 (16-bit PCM, decoded inside the kernel as s / 32768); with keep_float_dtype
    float64 / float16 / bfloat16 rows keep their type (frame_signal)."""
    require_gpu()
    origin = x.device
    if x.dim() != 2:
        raise ValueError(f"expected a 2-D (rows, samples) tensor, got {tuple(x.shape)}")
    if x.dtype == torch.int16 and not pcm_ok:
        raise TypeError("this transform takes float samples; int16 PCM is read by the log-mel and log-spectrogram transforms only")
    if x.dtype != torch.float32 and x.dtype != torch.int16 and not (keep_float_dtype and x.dtype in (torch.float64, torch.float16, torch.bfloat16)):
        x = x.to(torch.float32)
    if not x.is_cuda:
        x = x.cuda(non_blocking=False)
    if x.shape[0] > 1 and (x.stride(1) != 1 or x.stride(0) < x.shape[1]):
        x = x.contiguous()
    elif x.stride(1) != 1:
        x = x.contiguous()
    return x, origin


def row_lengths(lengths, rows: int, n: int, device: torch.device) -> torch.Tensor:
    """Per-row sample counts of a ragged batch as the int32 device tensor mmfe_run_ragged reads."""
    t = torch.as_tensor(lengths)
    if t.dim() != 1 or t.shape[0] != rows:
        raise ValueError(f"lengths must hold one entry per row ({rows}), got {tuple(t.shape)}")
    if t.is_floating_point():
        raise TypeError("lengths must be integers")
    if not t.is_cuda and rows > 0 and (int(t.min()) < 0 or int(t.max()) > n):
        raise ValueError(f"lengths must lie in [0, {n}] (the width of the batch tensor)")
    return t.to(device=device, dtype=torch.int32).contiguous()


def run_rows(plan: N.Plan, x: torch.Tensor, out: torch.Tensor = None, lengths=None) -> torch.Tensor:
    """Run `plan` over the rows of a CUDA tensor (rows, n).  Returns (rows, frames, feat) for time-major
    plans and (rows, feat, frames) for feature-major ones.  Asynchronous on the current stream.
    `lengths` (one integer per row, <= n) makes the batch ragged: row r is transformed as if it were x[r, :lengths[r]]
    passed alone, in the same single launch; frames it does not have are filled with the plan's pad value."""
    assert x.is_cuda and x.dim() == 2 and x.dtype == IN_DTYPES[plan.spec.in_dtype]
    rows, n = x.shape
    frames, feat = plan.out_shape(n)
    shape = (rows, frames, feat) if plan.spec.layout == N.LAYOUT_TIME_MAJOR or plan.spec.kind == N.KIND_FRAME \
        else (rows, feat, frames)
    dtype = x.dtype if plan.spec.kind == N.KIND_FRAME else OUT_DTYPES[plan.spec.out_dtype]     # frames have their rows' type
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=x.device)
    else:
        if tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"out must be a contiguous {dtype} tensor of shape {shape} on {x.device}")
    if rows == 0 or frames == 0 or feat == 0:
        return out
    if x.device.index != plan.device:
        raise ValueError(f"tensor on {x.device} but plan on cuda:{plan.device}")
    global _last_kernel
    _last_kernel = plan.kernel
    stream = torch.cuda.current_stream(x.device).cuda_stream
    stride = x.stride(0) if rows > 1 else max(x.stride(0), n)
    if lengths is None:
        plan.run_ptr(x.data_ptr(), stride, rows, n, out.data_ptr(), stream)
    else:
        d_len = row_lengths(lengths, rows, n, x.device)
        plan.run_ragged_ptr(x.data_ptr(), stride, d_len.data_ptr(), rows, n, out.data_ptr(), stream)
        d_len.record_stream(torch.cuda.current_stream(x.device))      # must outlive the asynchronous launch
    return out
