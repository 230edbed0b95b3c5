"""frame_signal on MI355X: drop-in for classification/vggsound/utils/transforms.py:19-61
(identical copies under classification/recognition/utils/ and classification/retrieval/utils/)."""
from __future__ import annotations

import numpy as np
import torch

import hashlib

from . import _native as N
from .plans import cached_plan, run_rows, to_device_rows


def frame_signal(signal: torch.Tensor, frame_length: int, hop_length: int, window: torch.Tensor = None) -> torch.Tensor:
    """(..., L) -> (..., num_frames, frame_length): centre zero padding (left gets floor(pad/2)),
    frames every hop_length samples, each multiplied by `window` (ones by default)."""
    if window is None:
        window = torch.ones(frame_length, dtype=signal.dtype, device=signal.device)
    if window.shape[0] != frame_length:
        raise ValueError('Wrong `window` length: expected {}, got {}'.format(window.shape[0], frame_length))
    lead = tuple(signal.shape[:-1])
    rows = signal.reshape(-1, signal.shape[-1])
    if rows.dtype == torch.int16:
        rows = rows.to(torch.float32)      # framing moves values, it does not decode them: exact in float32
    x, origin = to_device_rows(rows)
    # the window is data: plans are cached by its bytes, so repeated calls with the same window neither rebuild the device
    # table nor synchronise the stream
    win = np.ascontiguousarray(window.detach().to("cpu", torch.float32).numpy())
    key = ("frame", int(frame_length), int(hop_length), hashlib.sha1(win.tobytes()).hexdigest())

    def build(dev: int) -> N.Plan:
        spec = N.make_spec(kind=N.KIND_FRAME, win_len=int(frame_length), hop=int(hop_length), frames_mode=N.FRAMES_CEIL)
        return N.Plan(spec, win, device=dev)

    y = run_rows(cached_plan(key, x.device.index, build), x)
    y = y.reshape(*lead, y.shape[-2], y.shape[-1]).to(signal.dtype)
    return y if origin.type == "cuda" else y.to(origin)
