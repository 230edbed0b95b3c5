"""HAR IMU front-end on MI355X: drop-in for the STFT line of ADBox.__getitem__
(classification/har/adbox.py:20):  np.abs(scipy.signal.stft(imu.T, fs=100, nperseg=32, noverlap=28)[-1])."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as N
from . import melbank
from .plans import cached_plan, run_rows, to_device_rows


def _stft_plan(device: int, nperseg: int, noverlap: int) -> N.Plan:
    key = ("har", nperseg, noverlap)

    def build(dev: int) -> N.Plan:
        hop = nperseg - noverlap
        if noverlap >= nperseg:
            raise ValueError("noverlap must be less than nperseg.")
        win64 = melbank.periodic_hann(nperseg)
        spec = N.make_spec(kind=N.KIND_STFT_MAG, n_fft=nperseg, win_len=nperseg, hop=hop, pad_mode=N.PAD_ZERO,
                           frames_mode=N.FRAMES_SCIPY, n_mel=0, power=1, spec_scale=float(1.0 / win64.sum()),
                           log_kind=N.LOG_NONE, layout=N.LAYOUT_FEAT_MAJOR)
        return N.Plan(spec, win64.astype(np.float32), device=dev)

    return cached_plan(key, device, build)


def stft_mag(x, fs: float = 100, nperseg: int = 32, noverlap: int = 28, out: Optional[torch.Tensor] = None):
    """(..., T) -> (..., nperseg//2 + 1, 1 + ceil(T / (nperseg - noverlap))) STFT magnitudes with scipy's
    defaults (periodic Hann, zero boundary extension, zero padding to whole hops, 'spectrum' scaling).
    `fs` only labels scipy's frequency axis and does not change the values.  numpy in -> numpy out,
    tensor in -> tensor out on the tensor's device."""
    is_np = isinstance(x, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if is_np else x
    if t.shape[-1] < nperseg:
        # scipy shrinks nperseg to the input length in this case (a different transform); the reference's IMU
        # windows are always longer, so it is rejected rather than silently computed with nperseg taps
        raise ValueError(f"input length {t.shape[-1]} is shorter than nperseg {nperseg}")
    lead = tuple(t.shape[:-1])
    rows = t.reshape(-1, t.shape[-1])
    d, origin = to_device_rows(rows)
    plan = _stft_plan(d.device.index, int(nperseg), int(noverlap))
    y = run_rows(plan, d, out.reshape(d.shape[0], *out.shape[-2:]) if out is not None else None)
    y = y.reshape(*lead, y.shape[-2], y.shape[-1])
    if origin.type != "cuda":
        y = y.to(origin)
    return y.numpy() if is_np else y
