"""Host-side construction of windows and mel filter banks, and their banded (CSR-by-filter) form.

The tables are built once per plan on the host, in the dtype and operation order of the library the
reference calls, and uploaded; the device never recomputes them (fp64-built kaldi weights alone would
move log-mel outputs by ~4e-4).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch


def kaldi_window(window_type: str, size: int, blackman_coeff: float = 0.42) -> np.ndarray:
    """Feature window of kaldi.fbank (symmetric, fp32), as used through
    classification/vggsound/dataset/vggsound.py:45-47 (window_type='hanning')."""
    if window_type == "hanning":
        w = torch.hann_window(size, periodic=False, dtype=torch.float32)
    elif window_type == "hamming":
        w = torch.hamming_window(size, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    elif window_type == "povey":
        w = torch.hann_window(size, periodic=False, dtype=torch.float32).pow(0.85)
    elif window_type == "rectangular":
        w = torch.ones(size, dtype=torch.float32)
    elif window_type == "blackman":
        a = 2 * math.pi / (size - 1)
        n = torch.arange(size, dtype=torch.float32)
        w = blackman_coeff - 0.5 * torch.cos(a * n) + (0.5 - blackman_coeff) * torch.cos(2 * a * n)
    else:
        raise ValueError("Invalid window type " + window_type)
    return w.numpy()


def periodic_hann(size: int) -> np.ndarray:
    """scipy.signal.get_window('hann', size, fftbins=True), bit for bit: scipy evaluates the symmetric
    size+1 point generalised cosine 0.5*cos(0*t) + 0.5*cos(1*t) on t = linspace(-pi, pi, size+1) and drops
    the last point (librosa.stft and scipy.signal.stft both take their window from there)."""
    t = np.linspace(-np.pi, np.pi, size + 1)
    w = np.zeros(size + 1)
    for k, a in enumerate((0.5, 0.5)):
        w += a * np.cos(k * t)
    return w[:-1]


def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float, low_freq: float = 20.0,
                    high_freq: float = 0.0) -> torch.Tensor:
    """kaldi triangular filters, (num_bins, padded_window // 2) fp32.  Band edges are Python
    float64 scalars, the per-bin arithmetic runs on fp32 tensors in this exact order (vtln_warp = 1):
    up = (mel - left) / (center - left), down = (right - mel) / (right - center), max(0, min(up, down))."""
    num_fft_bins = padded_window / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    if not (0.0 <= low_freq < nyquist and 0.0 < high_freq <= nyquist and low_freq < high_freq):
        raise ValueError(f"Bad values in options: low-freq {low_freq} and high-freq {high_freq} vs. nyquist {nyquist}")
    fft_bin_width = sample_freq / padded_window
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    idx = torch.arange(num_bins).unsqueeze(1)
    left = mel_low + idx * delta
    center = mel_low + (idx + 1.0) * delta
    right = mel_low + (idx + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def slaney_mel_filters(n_freqs: int, n_mels: int, sr: float, fmin: float = 0.0, fmax: float = None) -> np.ndarray:
    """(n_mels, n_freqs) Slaney-scale, Slaney-normalised triangular filters (long-utterance log-mel)."""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp = 200.0 / 3.0
    logstep = math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), f_sp * m)

    fft_freqs = np.linspace(0.0, sr / 2.0, n_freqs)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    diff = np.diff(edges)
    slopes = edges[:, None] - fft_freqs[None, :]
    down = -slopes[:-2] / diff[:-1, None]
    up = slopes[2:] / diff[1:, None]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb


def dense_to_bands(weights) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(n_mel, n_bins) dense filter bank -> (band_start, band_len, band_weights): for every filter the
    span from its first to its last non-zero bin (empty filters get length 0)."""
    w = np.asarray(weights, dtype=np.float32)
    n_mel = w.shape[0]
    start = np.zeros(n_mel, dtype=np.int32)
    length = np.zeros(n_mel, dtype=np.int32)
    chunks = []
    for m in range(n_mel):
        nz = np.flatnonzero(w[m])
        if nz.size:
            start[m] = nz[0]
            length[m] = nz[-1] - nz[0] + 1
            chunks.append(w[m, nz[0]:nz[-1] + 1])
    flat = np.concatenate(chunks).astype(np.float32) if chunks else np.zeros(0, dtype=np.float32)
    return start, length, flat
