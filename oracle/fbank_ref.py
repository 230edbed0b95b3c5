"""ctypes binding of oracle/fbank_ref.c (the C / OpenMP restatement of the VGGSound front-end): TEST INFRASTRUCTURE.

Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this module.  The window and the mel bank it hands
to the C code are the float32 tables of oracle/frontend_oracle.py (torchaudio's formulas); everything per sample and per
frame happens in C.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import frontend_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libfbank_ref.so")
_lib = None


class Params(ctypes.Structure):
    _fields_ = [("n_fft", ctypes.c_int), ("win_len", ctypes.c_int), ("hop", ctypes.c_int), ("n_mel", ctypes.c_int),
                ("out_frames", ctypes.c_int), ("preemph", ctypes.c_float), ("remove_dc", ctypes.c_int),
                ("log_floor", ctypes.c_float), ("pad_value", ctypes.c_float), ("out_shift", ctypes.c_float),
                ("out_scale", ctypes.c_float)]


def build() -> str:
    """Compile the library with gcc (make -C oracle); returns its path."""
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        lib.fbank_ref_batch.restype = ctypes.c_int
        lib.fbank_ref_batch.argtypes = [ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.fbank_ref_num_frames.restype = ctypes.c_int64
        lib.fbank_ref_num_frames.argtypes = [ctypes.POINTER(Params), ctypes.c_int64]
        lib.fbank_ref_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _tables(sample_frequency: float, frame_length: float, num_mel_bins: int, window_type: str, low_freq: float, high_freq: float):
    size = int(sample_frequency * frame_length * 0.001)
    padded = O._next_pow2(size)
    window = np.ascontiguousarray(O.kaldi_window(window_type, size).numpy(), dtype=np.float32)
    bank = O.kaldi_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq).numpy()      # (n_mel, padded / 2)
    start = np.zeros(num_mel_bins, np.int32)
    length = np.zeros(num_mel_bins, np.int32)
    weights = []
    for m in range(num_mel_bins):
        nz = np.nonzero(bank[m])[0]
        if nz.size:
            start[m], length[m] = nz[0], nz[-1] - nz[0] + 1
            weights.append(bank[m, nz[0]:nz[-1] + 1])
    w = np.ascontiguousarray(np.concatenate(weights) if weights else np.zeros(1), dtype=np.float32)
    return size, padded, window, start, length, w


def fbank_batch(waves: np.ndarray, *, sample_frequency: float = 16000.0, frame_length: float = 25.0, frame_shift: float = 10.0,
                num_mel_bins: int = 23, window_type: str = "povey", preemphasis_coefficient: float = 0.97,
                remove_dc_offset: bool = True, low_freq: float = 20.0, high_freq: float = 0.0, out_frames: int = 0,
                pad_value: float = 0.0, out_shift: float = 0.0, out_scale: float = 1.0, threads: int = 0) -> np.ndarray:
    """(B, n) float32 clips -> (B, frames, num_mel_bins) kaldi log-mel features (snip_edges, dither 0, no energy, power spectrum),
    optionally cut / padded to out_frames and mapped through (x + out_shift) * out_scale.  threads <= 0: OpenMP's default."""
    lib = load()
    x = np.ascontiguousarray(waves, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("expected (clips, samples)")
    size, padded, window, start, length, w = _tables(sample_frequency, frame_length, num_mel_bins, window_type, low_freq, high_freq)
    p = Params(n_fft=padded, win_len=size, hop=int(sample_frequency * frame_shift * 0.001), n_mel=num_mel_bins, out_frames=out_frames,
               preemph=preemphasis_coefficient, remove_dc=int(remove_dc_offset), log_floor=O.KALDI_EPS, pad_value=pad_value,
               out_shift=out_shift, out_scale=out_scale)
    rows = out_frames if out_frames > 0 else int(lib.fbank_ref_num_frames(ctypes.byref(p), x.shape[1]))
    out = np.empty((x.shape[0], rows, num_mel_bins), np.float32)
    rc = lib.fbank_ref_batch(ctypes.byref(p), x.ctypes.data, x.shape[0], x.shape[1], x.shape[1], window.ctypes.data, start.ctypes.data,
                             length.ctypes.data, w.ctypes.data, out.ctypes.data, int(threads))
    if rc != 0:
        raise RuntimeError("fbank_ref_batch failed")
    return out


def preprocessing_audio_batch(waves: np.ndarray, threads: int = 0) -> np.ndarray:
    """VGGSound.preprocessing_audio (vggsound.py:44-61) for a batch: (B, n) -> (B, 384, 128)."""
    return fbank_batch(waves, sample_frequency=16000.0, frame_shift=25.0, num_mel_bins=128, window_type="hanning", out_frames=384,
                       pad_value=0.0, out_shift=5.1991, out_scale=1.0 / 4.4049, threads=threads)


def max_threads() -> int:
    return int(load().fbank_ref_max_threads())
