"""EPIC-Kitchens audio front-end on MI355X.

Drop-ins for EPICKitchen._log_specgram and the slice arithmetic of EPICKitchen._extract_sound_feature
(classification/recognition/dataset/epic_kitchen.py:114-154; identical code at
classification/retrieval/dataset/epic_kitchen.py:141-181) and for the Spec transform chain
Stack -> ToTorchFormatTensor(div=True) -> GroupNormalize([0],[1])
(classification/recognition/dataset/transform.py:253-285, wired at epic_kitchen.py:62-66).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from . import melbank
from .plans import IN_CODES, OUT_CODES, cached_plan, require_gpu, run_rows, to_device_rows

N_FFT = 511               # epic_kitchen.py:119
RESAMPLING_RATE = 24000   # epic_kitchen.py:77
HALF_SPAN_SEC = 0.639     # epic_kitchen.py:131-132
EDGE_SEC = 1.279          # epic_kitchen.py:147,150


def _logspec_plan(device: int, nperseg: int, hop: int, eps: float, out_scale: float, in_dtype: int = N.IN_F32,
                  out_dtype: int = N.OUT_F32) -> N.Plan:
    key = ("epic", nperseg, hop, float(eps), float(out_scale), int(in_dtype), int(out_dtype))

    def build(dev: int) -> N.Plan:
        if nperseg > N_FFT:
            raise ValueError(f"win_length {nperseg} larger than n_fft {N_FFT}")
        spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=N_FFT, win_len=nperseg, win_off=(N_FFT - nperseg) // 2, hop=hop,
                           pad_mode=N.PAD_ZERO, frames_mode=N.FRAMES_CENTER, n_mel=0, power=2,
                           log_kind=N.LOG_LN_ADD, log_floor=float(eps), out_shift=0.0, out_scale=out_scale,
                           layout=N.LAYOUT_FEAT_MAJOR, window_f64=1, in_dtype=in_dtype, out_dtype=out_dtype)
        win = melbank.periodic_hann(nperseg)      # float64, multiplied in float64 like librosa does
        return N.Plan(spec, win, device=dev)

    return cached_plan(key, device, build)


def _params(window_size, step_size, resampling_rate):
    nperseg = int(round(window_size * resampling_rate / 1e3))
    hop = int(round(step_size * resampling_rate / 1e3))
    return nperseg, hop


def log_specgram_batch(audio: torch.Tensor, window_size: int = 10, step_size: int = 5, eps: float = 1e-6,
                       resampling_rate: int = RESAMPLING_RATE, div255: bool = False,
                       out: Optional[torch.Tensor] = None, lengths=None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(B, n) slices -> (B, 256, 1 + (n-1)//hop) log power spectrograms (frequency-major), one launch.
    lengths: slices of different lengths in one batch (30 672-sample interior and 30 696-sample edge slices both give
    256 frames, epic_kitchen.py:143-154); row r is audio[r, :lengths[r]], zero-padded at its own end like librosa does.
    div255=True folds the ToTorchFormatTensor division in, giving what the encoder consumes.
    int16 input is the pcm_s16le audio the reference extracts (recognition/dataset/extract_audio.py:9), decoded in
    the kernel as s / 32768 like librosa.load does (epic_kitchen.py:135).
    dtype=torch.float16 / torch.bfloat16 stores the value `.to(dtype)` of the float32 result would give (what an encoder under
    autocast casts its input to), rounded in the kernel."""
    if dtype not in OUT_CODES:
        raise TypeError(f"dtype must be float32, float16 or bfloat16, got {dtype}")
    x, origin = to_device_rows(audio, pcm_ok=True)
    nperseg, hop = _params(window_size, step_size, resampling_rate)
    plan = _logspec_plan(x.device.index, nperseg, hop, eps, 1.0 / 255.0 if div255 else 1.0, IN_CODES[x.dtype], OUT_CODES[dtype])
    y = run_rows(plan, x, out, lengths)
    return y if origin.type == "cuda" else y.to(origin)


def log_specgram(audio, window_size: int = 10, step_size: int = 5, eps: float = 1e-6,
                 resampling_rate: int = RESAMPLING_RATE):
    """_log_specgram(self, audio, window_size=10, step_size=5, eps=1e-6): (n,) -> (256, frames).
    numpy in -> numpy out, tensor in -> tensor out (on the tensor's device)."""
    is_np = isinstance(audio, np.ndarray)
    if is_np:
        t = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.int16 if audio.dtype == np.int16 else np.float32))
    else:
        t = audio
    if t.dim() != 1:
        raise ValueError(f"expected a 1-D audio slice, got {tuple(t.shape)}")
    y = log_specgram_batch(t[None, :], window_size, step_size, eps, resampling_rate)[0]
    return y.numpy() if is_np else y


def spec_to_tensor(spec) -> torch.Tensor:
    """The Spec transform chain applied to one (H, W) float image: (1, H, W) / 255, mean 0 / std 1."""
    t = torch.from_numpy(np.ascontiguousarray(spec)) if isinstance(spec, np.ndarray) else spec
    t = t.to(torch.float32).unsqueeze(0)
    # a true division, like ToTorchFormatTensor's `.div(255)` on the CPU (transform.py:285): on a GPU torch turns division by a
    # Python scalar into a multiplication by its reciprocal, one rounding away; a tensor divisor keeps the division
    return (torch.div(t, t.new_full((), 255.0)) - 0.0) / 1.0


def sound_slice_bounds_batch(start_frame, idx, fps_spec, n_total, resampling_rate: int = RESAMPLING_RATE) -> np.ndarray:
    """The [left, right) sample ranges _extract_sound_feature cuts (epic_kitchen.py:130-152) for a whole batch at once:
    array arguments of equal length (or scalars) -> (B, 2) int64.  +-0.639 s around (start_frame + idx) / fps, rounded
    half-to-even like Python's round(); before the first / beyond the last sample the first / last round(1.279 * sr)
    samples are taken instead (all of a shorter recording).  float64 throughout, like the Python scalars of the reference."""
    frame = np.asarray(start_frame, dtype=np.int64) + np.asarray(idx, dtype=np.int64)
    centre = frame.astype(np.float64) / np.asarray(fps_spec, dtype=np.float64)
    n_total = np.broadcast_to(np.asarray(n_total, dtype=np.int64), centre.shape)
    sr = float(resampling_rate)
    left_sec, right_sec = centre - HALF_SPAN_SEC, centre + HALF_SPAN_SEC
    edge = int(round(resampling_rate * EDGE_SEC))
    at_start = left_sec < 0
    at_end = ~at_start & (right_sec > n_total / sr)
    left = np.rint(left_sec * resampling_rate).astype(np.int64)
    right = np.rint(right_sec * resampling_rate).astype(np.int64)
    left = np.where(at_start, 0, np.where(at_end, np.maximum(n_total - edge, 0), left))
    right = np.where(at_start, np.minimum(edge, n_total), np.where(at_end, n_total, right))
    return np.stack([left, right], axis=-1)


def sound_slice_bounds(start_frame: int, idx: int, fps_spec: float, n_total: int,
                       resampling_rate: int = RESAMPLING_RATE) -> Tuple[int, int]:
    """One sample's [left, right) range: see sound_slice_bounds_batch."""
    l, r = sound_slice_bounds_batch([start_frame], [idx], [fps_spec], [n_total], resampling_rate)[0]
    return int(l), int(r)


def extract_sound_features(recordings: Sequence[torch.Tensor], rec_index: Sequence[int], start_frame: Sequence[int],
                           idx: Sequence[int], fps_spec: Sequence[float], resampling_rate: int = RESAMPLING_RATE,
                           div255: bool = True, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Batched _extract_sound_feature + _log_specgram (+ /255): for sample i the slice of
    recordings[rec_index[i]] around (start_frame[i] + idx[i]) / fps_spec[i] is transformed straight from
    the packed device-resident recordings (no per-slice copy).  Interior slices (30 672 samples) and edge slices
    (30 696) may be mixed, as a DataLoader batch of the reference mixes them: one launch serves all of them.
    Recordings that are all int16 stay 16-bit PCM on the device (half the memory) and are decoded in the kernel.
    Returns (B, 1, 256, frames), float32 or (dtype) the float16 / bfloat16 rounding of it."""
    if dtype not in OUT_CODES:
        raise TypeError(f"dtype must be float32, float16 or bfloat16, got {dtype}")
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    pcm = len(recordings) > 0 and all(r.dtype == torch.int16 for r in recordings)
    el = torch.int16 if pcm else torch.float32
    recs = [r.to(device=dev, dtype=el).reshape(-1) for r in recordings]
    lengths = [int(r.numel()) for r in recs]
    starts = np.zeros(len(recs), dtype=np.int64)
    if recs:
        starts[1:] = np.cumsum(lengths[:-1])
    packed = torch.cat(recs) if recs else torch.empty(0, dtype=el, device=dev)
    bounds = sound_slice_bounds_batch(start_frame, idx, fps_spec, [lengths[i] for i in rec_index], resampling_rate).reshape(-1, 2).tolist()
    sizes = [r - l for l, r in bounds]
    n = max(sizes) if sizes else 0
    ragged = len(set(sizes)) > 1
    nperseg, hop = _params(10, 5, resampling_rate)
    plan = _logspec_plan(dev.index, nperseg, hop, 1e-6, 1.0 / 255.0 if div255 else 1.0, IN_CODES[el], OUT_CODES[dtype])
    frames, feat = plan.out_shape(n)
    out = torch.empty((len(bounds), 1, feat, frames), dtype=dtype, device=dev)
    if len(bounds) == 0 or frames == 0:
        return out
    d_start = torch.from_numpy(starts).to(dev)
    d_index = torch.tensor(list(rec_index), dtype=torch.int32, device=dev)
    d_off = torch.tensor([l for l, _ in bounds], dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    d_len = torch.tensor(sizes, dtype=torch.int32, device=dev)
    if ragged:
        plan.run_slices_ragged_ptr(packed.data_ptr(), d_start.data_ptr(), d_index.data_ptr(), d_off.data_ptr(), d_len.data_ptr(),
                                   len(bounds), n, out.data_ptr(), stream)
    else:
        plan.run_slices_ptr(packed.data_ptr(), d_start.data_ptr(), d_index.data_ptr(), d_off.data_ptr(), len(bounds), n,
                            out.data_ptr(), stream)
    # the index tensors must outlive the asynchronous launch on this stream
    for t in (packed, d_start, d_index, d_off, d_len):
        t.record_stream(torch.cuda.current_stream(dev))
    return out
