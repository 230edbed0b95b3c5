"""frame_signal on MI355X: drop-in for classification/vggsound/utils/transforms.py:19-61
(identical copies under classification/recognition/utils/ and classification/retrieval/utils/)."""
from __future__ import annotations

import hashlib
import threading
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import _native as N
from .plans import IN_CODES, cached_plan, run_rows, to_device_rows

_NATIVE = (torch.float32, torch.float64, torch.float16, torch.bfloat16)
_MAX_WINDOWS = 16       # distinct (window, frame_length, hop_length, dtype) plans kept per process

# Windows already seen, by tensor identity: id -> (weak reference, data_ptr, version counter, work dtype, digest).  A repeated call
# with the same unmodified window tensor then needs neither a device-to-host copy of the window nor a hash of its bytes.  The weak
# reference guards against a new tensor that reuses a dead one's id and address.
_seen = OrderedDict()
_seen_lock = threading.Lock()


def _host_window(window: torch.Tensor, work: torch.dtype) -> np.ndarray:
    """The window as the multiplication sees it: cast to the product's dtype, then widened (exactly) to what the C ABI takes."""
    w = window.detach().to("cpu").to(work)
    return np.ascontiguousarray(w.to(torch.float64 if work == torch.float64 else torch.float32).numpy())


def _window_key(window: torch.Tensor, work: torch.dtype):
    with _seen_lock:
        hit = _seen.get(id(window))
        if hit is not None and hit[0]() is window and hit[1:4] == (window.data_ptr(), window._version, work):
            _seen.move_to_end(id(window))
            return hit[4], None
    win = _host_window(window, work)
    digest = hashlib.sha1(win.tobytes()).hexdigest()
    with _seen_lock:
        _seen[id(window)] = (weakref.ref(window), window.data_ptr(), window._version, work, digest)
        while len(_seen) > 4 * _MAX_WINDOWS:
            _seen.popitem(last=False)
    return digest, win


def frame_signal(signal: torch.Tensor, frame_length: int, hop_length: int, window: torch.Tensor = None) -> torch.Tensor:
    """(..., L) -> (..., num_frames, frame_length): centre zero padding (left gets floor(pad/2)),
    frames every hop_length samples, each multiplied by `window` (ones of the signal's dtype by default).
    The product is taken in the dtype torch's promotion gives signal * window, as the reference's
    `pad_signal[..., indices] * window` does (transforms.py:24-25, 58-59): float32, float64, float16 and bfloat16 run
    in the kernel in that type; integer results (an integer signal with the default window) are framed exactly in float64."""
    if window is None:
        window = torch.ones(frame_length, dtype=signal.dtype, device=signal.device)
    if window.shape[0] != frame_length:
        raise ValueError('Wrong `window` length: expected {}, got {}'.format(window.shape[0], frame_length))
    result = torch.result_type(signal, window)
    work = result if result in _NATIVE else torch.float64
    lead = tuple(signal.shape[:-1])
    rows = signal.reshape(-1, signal.shape[-1]).to(work)
    x, origin = to_device_rows(rows, keep_float_dtype=True)
    digest, win = _window_key(window, work)
    key = ("frame", int(frame_length), int(hop_length), str(work), digest)

    def build(dev: int) -> N.Plan:
        w = win if win is not None else _host_window(window, work)     # (None: the window was known but its plan has been evicted)
        spec = N.make_spec(kind=N.KIND_FRAME, win_len=int(frame_length), hop=int(hop_length), frames_mode=N.FRAMES_CEIL,
                           in_dtype=IN_CODES[work], window_f64=1 if work == torch.float64 else 0)
        return N.Plan(spec, w, device=dev)

    y = run_rows(cached_plan(key, x.device.index, build, family="frame", limit=_MAX_WINDOWS), x)
    y = y.reshape(*lead, y.shape[-2], y.shape[-1]).to(result)
    return y if origin.type == "cuda" else y.to(origin)
