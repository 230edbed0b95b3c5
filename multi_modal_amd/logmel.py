"""Long-utterance log-mel front-end (n_fft 400, hop 160, 80 mel bands, 30 s utterances).

These parameters do not occur in the reference (its EPIC audio path uses n_fft 511 / hop 120, see
epic.py); the configuration is the Whisper-style feature extractor and is offered as a throughput
configuration of the same kernel set: periodic Hann, centre reflect padding, |X|^2, Slaney mel bank,
log10(max(., 1e-10)), last frame dropped, per-utterance clamp to (max - 8), (x + 4) / 4, output
(B, n_mels, frames).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as N
from . import melbank
from .plans import IN_CODES, OUT_CODES, cached_plan, run_rows, to_device_rows


def _plan(device: int, n_fft: int, hop: int, n_mels: int, sr: int, f64: bool, in_dtype: int = N.IN_F32, stream: int = 0,
          out_dtype: int = N.OUT_F32) -> N.Plan:
    # A plan with clamp_below_max owns the per-row extremes its two passes exchange, so two calls on one plan must not
    # overlap on the device: the cache keeps one plan per HIP stream (calls on one stream are ordered), at most _MAX_STREAM_PLANS
    # of them (each holds an 8 MiB row-extreme scratch; code that keeps creating streams reuses the oldest entries' slots).
    key = ("logmel", n_fft, hop, n_mels, sr, f64, int(in_dtype), int(stream), int(out_dtype))

    def build(dev: int) -> N.Plan:
        fb = melbank.slaney_mel_filters(1 + n_fft // 2, n_mels, sr)
        bs, bl, bw = melbank.dense_to_bands(fb.astype(np.float32))
        spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=n_fft, win_len=n_fft, win_off=0, hop=hop, pad_mode=N.PAD_REFLECT,
                           frames_mode=N.FRAMES_CENTER, n_mel=n_mels, power=2, log_kind=N.LOG_LOG10_MAX,
                           log_floor=1e-10, out_shift=4.0, out_scale=0.25, drop_last=1, clamp_below_max=8.0,
                           layout=N.LAYOUT_FEAT_MAJOR, window_f64=1 if f64 else 0, in_dtype=in_dtype, out_dtype=out_dtype)
        return N.Plan(spec, melbank.periodic_hann(n_fft), bs, bl, bw, device=dev)

    return cached_plan(key, device, build, family="logmel", limit=_MAX_STREAM_PLANS)


_MAX_STREAM_PLANS = 8


def log_mel_spectrogram(waveforms: torch.Tensor, n_fft: int = 400, hop_length: int = 160, n_mels: int = 80,
                        sampling_rate: int = 16000, out: Optional[torch.Tensor] = None,
                        precision: str = "float32", dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(B, n) -> (B, n_mels, n // hop_length).

    precision="float32" computes the spectrum in float32 (the radix kernel when n_fft is 400, as torch.stft
    would); "float64" accumulates the DFT in float64 like the numpy feature extractor (any-length kernel).
    An int16 tensor is taken as 16-bit PCM (value s / 32768) and decoded inside the kernel.
    dtype=torch.float16 / torch.bfloat16 (float32 precision only) stores the value `.to(dtype)` of the float32 features would give.
    """
    if precision not in ("float32", "float64"):
        raise ValueError("precision must be 'float32' or 'float64'")
    if dtype not in OUT_CODES:
        raise TypeError(f"dtype must be float32, float16 or bfloat16, got {dtype}")
    x, origin = to_device_rows(waveforms, pcm_ok=True)
    plan = _plan(x.device.index, n_fft, hop_length, n_mels, sampling_rate, precision == "float64", IN_CODES[x.dtype],
                 torch.cuda.current_stream(x.device).cuda_stream, OUT_CODES[dtype])
    y = run_rows(plan, x, out)
    return y if origin.type == "cuda" else y.to(origin)
