"""frame_signal on MI355X: drop-in for classification/vggsound/utils/transforms.py:19-61
(identical copies under classification/recognition/utils/ and classification/retrieval/utils/)."""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N
from .plans import run_rows, to_device_rows


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
    spec = N.make_spec(kind=N.KIND_FRAME, win_len=int(frame_length), hop=int(hop_length), frames_mode=N.FRAMES_CEIL)
    plan = N.Plan(spec, window.detach().to("cpu", torch.float32).numpy(), device=x.device.index)   # window is data: no cache
    try:
        y = run_rows(plan, x)
        torch.cuda.current_stream(x.device).synchronize()   # the plan (and its window table) dies with this call
    finally:
        plan.close()
    y = y.reshape(*lead, y.shape[-2], y.shape[-1]).to(signal.dtype)
    return y if origin.type == "cuda" else y.to(origin)
