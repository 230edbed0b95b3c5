"""Sample-rate conversion in front of the spectrogram path, on the GPU.

The reference converts its audio once, offline, on the CPU: `librosa.load(file, sr=16000, duration=10)` for VGGSound
(classification/vggsound/dataset/vggsound.py:114) and `ffmpeg -ar 24000` for EPIC
(classification/recognition/dataset/extract_audio.py:6-13).  This module is the same step as one batched launch of a
rational-ratio polyphase FIR resampler, so a pipeline can hand the decoded 44.1 / 48 kHz PCM of the files to the GPU and
go on to the log-mel kernels without a CPU pass.  The filter, the alignment and the output length are those of
scipy.signal.resample_poly (Kaiser-windowed sinc, beta 5, 20 * max(up, down) + 1 taps, zeros beyond the clip ends);
librosa's default resampler (soxr) is a different low-pass design and is not reproduced bit for bit.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import _native as N
from .plans import IN_CODES, cached_plan, to_device_rows


def lowpass_taps(up: int, down: int, beta: float = 5.0) -> np.ndarray:
    """The FIR low-pass scipy.signal.resample_poly designs for float32 input: firwin(20 * max(up, down) + 1, 1 / max(up, down),
    window=('kaiser', beta)) rounded to float32, then multiplied by `up`."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    cutoff = 1.0 / max_rate
    m = np.arange(n, dtype=np.float64) - half_len
    h = cutoff * np.sinc(cutoff * m)
    h *= np.kaiser(n, beta)
    h /= h.sum()
    h32 = h.astype(np.float32)
    h32 *= np.float32(up)
    return h32


def _ratio(orig_sr: int, target_sr: int):
    if orig_sr <= 0 or target_sr <= 0 or int(orig_sr) != orig_sr or int(target_sr) != target_sr:
        raise ValueError("sample rates must be positive integers")
    g = math.gcd(int(orig_sr), int(target_sr))
    return int(target_sr) // g, int(orig_sr) // g


def _plan(device: int, up: int, down: int, out_len: int, in_dtype: int) -> N.Plan:
    key = ("resample", up, down, out_len, in_dtype)

    def build(dev: int) -> N.Plan:
        taps = lowpass_taps(up, down)
        spec = N.make_spec(kind=N.KIND_RESAMPLE, n_fft=up, win_len=int(taps.shape[0]), hop=down, out_frames=out_len, in_dtype=in_dtype)
        return N.Plan(spec, taps, device=dev)

    return cached_plan(key, device, build)


def resample_batch(waveforms: torch.Tensor, orig_sr: int, target_sr: int, out_samples: int = 0,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, n) clips at orig_sr -> (B, ceil(n * target_sr / orig_sr)) float32 clips at target_sr, one launch for the batch.
    out_samples > 0 cuts / zero-pads every clip to that many output samples.  An int16 tensor is taken as 16-bit PCM
    (value s / 32768, decoded inside the kernel).  CPU tensors are copied to the current GPU and the result copied back."""
    x, origin = to_device_rows(waveforms, pcm_ok=True)
    up, down = _ratio(orig_sr, target_sr)
    rows, n = x.shape
    if up == 1 and down == 1:
        y = x.to(torch.float32) / 32768.0 if x.dtype == torch.int16 else x.clone()
        if out_samples > 0:
            y = torch.nn.functional.pad(y, (0, max(0, out_samples - n)))[:, :out_samples].contiguous()
    else:
        plan = _plan(x.device.index, up, down, int(out_samples), IN_CODES[x.dtype])
        n_out, _ = plan.out_shape(n)
        if out is None:
            out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (rows, n_out) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(rows, n_out)} on {x.device}")
        if rows > 0 and n_out > 0:
            stride = x.stride(0) if rows > 1 else max(x.stride(0), n)
            plan.run_ptr(x.data_ptr(), stride, rows, n, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        y = out
    return y if origin.type == "cuda" else y.to(origin)


def load_resampled(waveforms: torch.Tensor, orig_sr: int, sr: int = 16000, duration: Optional[float] = None) -> torch.Tensor:
    """The resampling half of `librosa.load(file, sr=sr, duration=duration)` (vggsound.py:114) for decoded clips: the first
    `duration` seconds at the original rate are kept, then converted; (B, n) -> (B, ceil(min(n, duration * orig_sr) * sr / orig_sr)).
    The low-pass is scipy.signal.resample_poly's Kaiser design, not librosa's default soxr_hq (absent here, parity with it
    unpinned): features computed from these clips differ from the reference's at the level of the resampling filter."""
    if waveforms.dim() == 1:
        return load_resampled(waveforms[None], orig_sr, sr, duration)[0]
    if duration is not None:
        waveforms = waveforms[:, : int(round(duration * orig_sr))]
    return resample_batch(waveforms, orig_sr, sr)
