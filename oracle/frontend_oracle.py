"""CPU oracle for the signal front-end: TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  Nothing under multi_modal_amd/ imports it.

It restates, on the CPU, the arithmetic that the reference's three dataset
classes delegate to third-party libraries, in the same dtype and operation
order those libraries use:

* kaldi_fbank / preprocessing_audio  -> torchaudio.compliance.kaldi.fbank as
  called at classification/vggsound/dataset/vggsound.py:45-47, post-processing
  of vggsound.py:48-59.  torchaudio is not installed in this image, it is not
  vendored in the reference and its version is not pinned there: PARITY
  UNPINNED.  The restatement follows torchaudio's published algorithm (fp32
  torch ops in torchaudio's order) and is cross-checked against an independent
  fp64 implementation (transformers.audio_utils.spectrogram, kaldi mel scale)
  and against known-answer cells.
* log_specgram / spec_to_tensor -> librosa.stft as called at
  classification/recognition/dataset/epic_kitchen.py:119-125 (same code at
  classification/retrieval/dataset/epic_kitchen.py:146-152) followed by the
  /255 of classification/recognition/dataset/transform.py:268-285.  librosa is
  not installed / vendored / pinned: PARITY UNPINNED.  Cross-checked against
  numpy.fft.rfft on explicitly framed data.
* stft_mag -> numpy.abs(scipy.signal.stft(...)[-1]) as called at
  classification/har/adbox.py:20.  scipy IS installed, so this restatement is
  pinned bit-for-bit by the third-party call itself (tests/golden/har_stft.npz).
* frame_signal -> classification/vggsound/utils/transforms.py:19-61 (identical
  copies under recognition/ and retrieval/).  Pinned by outputs of the
  reference function itself (tests/golden/frame_signal.npz).
* patch_embed -> PatchEmbed.forward (classification/vggsound/models/vit_model.py:193-199) with the projection
  AudioTransformer installs (vit_model.py:435), as @autocast() (vit_model.py:388) evaluates it: operands rounded
  to the 16-bit type, float32 sums.  Pinned by outputs of the reference's own class (tests/golden/patch_embed.npz).
* whisper_logmel -> not in the reference at all; the "n_fft=400 hop=160 80-mel
  30 s" configuration is defined by transformers' WhisperFeatureExtractor and
  is restated here only so the throughput configuration has a checker.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.signal
import torch

# --------------------------------------------------------------------------------------
# VGGSound: kaldi fbank (torchaudio.compliance.kaldi.fbank semantics), fp32 torch ops
# --------------------------------------------------------------------------------------

KALDI_EPS = float(torch.finfo(torch.float32).eps)  # 1.1920929e-07, log floor of fbank


def _next_pow2(x: int) -> int:
    return 1 if x == 0 else 2 ** (x - 1).bit_length()


def kaldi_window(window_type: str, size: int, blackman_coeff: float = 0.42) -> torch.Tensor:
    """Window table, fp32, symmetric (periodic=False) like kaldi's feature-window."""
    if window_type == "hanning":
        return torch.hann_window(size, periodic=False, dtype=torch.float32)
    if window_type == "hamming":
        return torch.hamming_window(size, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    if window_type == "povey":
        return torch.hann_window(size, periodic=False, dtype=torch.float32).pow(0.85)
    if window_type == "rectangular":
        return torch.ones(size, dtype=torch.float32)
    if window_type == "blackman":
        a = 2 * math.pi / (size - 1)
        n = torch.arange(size, dtype=torch.float32)
        return blackman_coeff - 0.5 * torch.cos(a * n) + (0.5 - blackman_coeff) * torch.cos(2 * a * n)
    raise ValueError("Invalid window type " + window_type)


def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float,
                    low_freq: float = 20.0, high_freq: float = 0.0) -> torch.Tensor:
    """(num_bins, padded_window/2) triangular filters, equally spaced on the kaldi mel
    scale 1127*ln(1+f/700); scalar edges in Python float64, everything per-bin in fp32
    tensors, in the order torchaudio evaluates them (vtln_warp == 1)."""
    num_fft_bins = padded_window / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded_window
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def kaldi_fbank(waveform: torch.Tensor, *, sample_frequency: float = 16000.0, frame_length: float = 25.0,
                frame_shift: float = 10.0, num_mel_bins: int = 23, window_type: str = "povey",
                preemphasis_coefficient: float = 0.97, remove_dc_offset: bool = True,
                round_to_power_of_two: bool = True, low_freq: float = 20.0, high_freq: float = 0.0,
                channel: int = -1, use_power: bool = True, use_log_fbank: bool = True,
                blackman_coeff: float = 0.42, compute_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """fp32 restatement of kaldi fbank with snip_edges=True, dither=0, use_energy=False,
    vtln_warp=1, subtract_mean=False (the only combination the reference uses).
    compute_dtype=torch.float64 evaluates the SAME formula (same fp32-built window and mel weights)
    in float64: the yardstick that tells how far any fp32 evaluation, the reference's included, is
    from the exact value of a cell."""
    x = waveform[max(channel, 0), :].to(torch.float32).to(compute_dtype)
    shift = int(sample_frequency * frame_shift * 0.001)
    size = int(sample_frequency * frame_length * 0.001)
    padded = _next_pow2(size) if round_to_power_of_two else size
    n = x.shape[0]
    if n < size:
        return torch.empty((0, 0), dtype=compute_dtype)
    m = 1 + (n - size) // shift
    frames = x.as_strided((m, size), (shift, 1))
    if remove_dc_offset:
        frames = frames - frames.mean(dim=1, keepdim=True)
    if preemphasis_coefficient != 0.0:
        prev = torch.cat((frames[:, :1], frames[:, :-1]), dim=1)  # replicate-pad on the left
        frames = frames - preemphasis_coefficient * prev
    frames = frames * kaldi_window(window_type, size, blackman_coeff).to(compute_dtype).unsqueeze(0)
    if padded != size:
        frames = torch.nn.functional.pad(frames, (0, padded - size))
    spec = torch.fft.rfft(frames).abs()
    if use_power:
        spec = spec.pow(2.0)
    bank = kaldi_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq)
    bank = torch.nn.functional.pad(bank, (0, 1)).to(compute_dtype)
    mel = torch.mm(spec, bank.T)
    if use_log_fbank:
        mel = torch.max(mel, torch.tensor(KALDI_EPS, dtype=compute_dtype)).log()
    return mel


VGG_TARGET_FRAMES = 384      # vggsound.py:48
VGG_NORM_MEAN = -5.1991      # vggsound.py:59
VGG_NORM_STD = 4.4049        # vggsound.py:59


def preprocessing_audio(audio: torch.Tensor) -> torch.Tensor:
    """VGGSound.preprocessing_audio (vggsound.py:44-61): (1, n) -> (384, 128)."""
    fb = kaldi_fbank(audio, sample_frequency=16000, window_type="hanning", num_mel_bins=128, frame_shift=25)
    p = VGG_TARGET_FRAMES - fb.shape[0]
    if p > 0:
        fb = torch.nn.functional.pad(fb, (0, 0, 0, p))
    elif p < 0:
        fb = fb[0:VGG_TARGET_FRAMES, :]
    return (fb - VGG_NORM_MEAN) / VGG_NORM_STD


# --------------------------------------------------------------------------------------
# EPIC-Kitchens: librosa.stft(n_fft=511, hop=120, win=240, hann, centre, zero pad) -> ln(|X|^2+eps)
# --------------------------------------------------------------------------------------

def centred_stft(y: np.ndarray, n_fft: int, hop_length: int, win_length: int, pad_mode: str = "constant") -> np.ndarray:
    """librosa-style centred STFT: periodic Hann of win_length (fp64) zero-padded to n_fft
    around its centre, signal padded by n_fft//2 both sides, frames multiplied in fp64,
    rfft in fp64, result stored complex64, layout (1 + n_fft//2, n_frames)."""
    y = np.asarray(y, dtype=np.float32)
    w = scipy.signal.get_window("hann", win_length, fftbins=True)      # what librosa.filters.get_window returns (fp64)
    lpad = (n_fft - win_length) // 2
    w = np.pad(w, (lpad, n_fft - win_length - lpad))
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (yp.shape[0] - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx]                                    # (n_fft, n_frames) fp32
    spec = np.fft.rfft(w[:, None] * frames, axis=0)     # fp64
    return spec.astype(np.complex64)


def log_specgram(audio: np.ndarray, window_size: int = 10, step_size: int = 5, eps: float = 1e-6,
                 resampling_rate: int = 24000) -> np.ndarray:
    """EPICKitchen._log_specgram (recognition/dataset/epic_kitchen.py:114-126)."""
    nperseg = int(round(window_size * resampling_rate / 1e3))
    hop = int(round(step_size * resampling_rate / 1e3))
    spec = centred_stft(audio, n_fft=511, hop_length=hop, win_length=nperseg, pad_mode="constant")
    return np.log(np.real(spec * np.conj(spec)) + eps)


def spec_to_tensor(spec: np.ndarray) -> torch.Tensor:
    """Stack -> ToTorchFormatTensor(div=True) -> GroupNormalize([0],[1]) applied to one mode-'F'
    image (recognition/dataset/transform.py:253-285, 63-76): (H, W) -> (1, H, W) / 255."""
    t = torch.from_numpy(np.ascontiguousarray(spec[:, :, None])).permute(2, 0, 1).contiguous()
    t = t.float().div(255)
    return (t - 0.0) / 1.0


def sound_slice_bounds(start_frame: int, idx: int, fps_spec: float, n_total: int, resampling_rate: int = 24000):
    """Slice chosen by EPICKitchen._extract_sound_feature (epic_kitchen.py:128-154): returns
    (left, right) sample indices into the untrimmed recording of n_total samples."""
    centre = (start_frame + idx) / fps_spec
    left_sec = centre - 0.639
    right_sec = centre + 0.639
    duration = n_total / float(resampling_rate)
    edge = int(round(resampling_rate * 1.279))
    if left_sec < 0:
        return 0, min(edge, n_total)
    if right_sec > duration:
        return max(n_total - edge, 0), n_total
    return int(round(left_sec * resampling_rate)), int(round(right_sec * resampling_rate))


# --------------------------------------------------------------------------------------
# HAR: numpy.abs(scipy.signal.stft(x, fs, nperseg, noverlap)[-1])  (adbox.py:20)
# --------------------------------------------------------------------------------------

def stft_mag(x: np.ndarray, fs: float = 100, nperseg: int = 32, noverlap: int = 28) -> np.ndarray:
    """scipy.signal.stft defaults restated: periodic Hann, boundary='zeros', padded=True,
    scaling='spectrum', one-sided; fp64 arithmetic on fp32 input, window rounded to fp32,
    result rounded to complex64, magnitude taken in fp32.  (..., T) -> (..., nperseg/2+1, S)."""
    x = np.asarray(x)
    if x.shape[-1] < nperseg:
        # scipy silently shrinks nperseg to the input length here (and then fails when noverlap >= nperseg);
        # the IMU windows of the reference are always longer, so this case is rejected, not imitated
        raise ValueError(f"input length {x.shape[-1]} is shorter than nperseg {nperseg}")
    hop = nperseg - noverlap
    win = scipy.signal.get_window("hann", nperseg)                      # scipy.signal.stft's default window (periodic, fp64)
    scale = 1.0 / win.sum()                       # sqrt(1/sum^2), 'spectrum' scaling
    win32 = win.astype(np.float32).astype(np.float64)
    half = nperseg // 2
    lead = x.shape[:-1]
    xe = np.concatenate((np.zeros(lead + (half,)), x, np.zeros(lead + (half,))), axis=-1)
    nadd = (-(xe.shape[-1] - nperseg) % hop) % nperseg
    xe = np.concatenate((xe, np.zeros(lead + (nadd,))), axis=-1)
    n_seg = 1 + (xe.shape[-1] - nperseg) // hop
    idx = hop * np.arange(n_seg)[:, None] + np.arange(nperseg)[None, :]
    seg = xe[..., idx] * win32                    # (..., S, nperseg) fp64
    z = np.fft.rfft(seg, axis=-1) * scale
    z = z.astype(np.complex64)
    z = np.moveaxis(z, -1, -2)                    # (..., bins, S)
    return np.abs(z)


# --------------------------------------------------------------------------------------
# frame_signal  (classification/vggsound/utils/transforms.py:19-61)
# --------------------------------------------------------------------------------------

def frame_signal(signal: torch.Tensor, frame_length: int, hop_length: int, window: torch.Tensor = None) -> torch.Tensor:
    if window is None:
        window = torch.ones(frame_length, dtype=signal.dtype)
    if window.shape[0] != frame_length:
        raise ValueError("Wrong `window` length: expected {}, got {}".format(window.shape[0], frame_length))
    n = signal.shape[-1]
    num_frames = 1 if n <= frame_length else 1 + int(math.ceil((1.0 * n - frame_length) / hop_length))
    pad_len = (num_frames - 1) * hop_length + frame_length
    if pad_len > n:
        z = pad_len - n
        signal = torch.nn.functional.pad(signal, (z // 2, z - z // 2))
    out = signal.unfold(-1, frame_length, hop_length)[..., :num_frames, :]
    return out * window


# --------------------------------------------------------------------------------------
# Long-utterance log-mel (Whisper-style; not in the reference)
# --------------------------------------------------------------------------------------

def slaney_mel_filters(n_freqs: int, n_mels: int, sr: float, fmin: float = 0.0, fmax: float = None) -> np.ndarray:
    """(n_freqs, n_mels) triangular filters on the Slaney mel scale with Slaney area
    normalisation, fp64 (the filter bank WhisperFeatureExtractor builds)."""
    fmax = sr / 2.0 if fmax is None else fmax

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        lin = 3.0 * f / 200.0
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4)), lin)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), 200.0 * m / 3.0)

    fft_freqs = np.linspace(0.0, sr / 2.0, n_freqs)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    diff = np.diff(edges)
    slopes = edges[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[None, :]
    return fb


def whisper_logmel(wave: np.ndarray, n_fft: int = 400, hop: int = 160, n_mels: int = 80, sr: int = 16000,
                   compute_dtype=np.float64) -> np.ndarray:
    """(n,) -> (n_mels, n // hop): periodic Hann, centre reflect padding, |X|^2, Slaney mel,
    log10(max(., 1e-10)), last frame dropped, clamp to (max - 8), (x + 4) / 4.  Maths in
    `compute_dtype` (float64 like the numpy feature extractor, or float32 like a torch.stft one),
    fp32 result."""
    y = np.asarray(wave, dtype=compute_dtype)
    w = scipy.signal.get_window("hann", n_fft, fftbins=True).astype(compute_dtype)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (yp.shape[0] - n_fft) // hop
    idx = hop * np.arange(n_frames)[:, None] + np.arange(n_fft)[None, :]
    spec = np.abs(np.fft.rfft(yp[idx] * w, axis=-1)) ** 2           # (frames, bins); numpy keeps float32 input in float32
    fb = slaney_mel_filters(1 + n_fft // 2, n_mels, sr).astype(compute_dtype)
    mel = np.maximum(compute_dtype(1e-10), spec @ fb)
    lg = np.log10(mel).T[:, :-1]
    lg = np.maximum(lg, lg.max() - 8.0)
    return ((lg + 4.0) / 4.0).astype(np.float32)


def resample_poly(x, orig_sr: int, target_sr: int):
    """Sample-rate conversion of the rows of `x` (float32): scipy.signal.resample_poly itself (Kaiser beta 5 low-pass,
    zeros beyond the ends).  PINNED by construction for that function; the reference's own call is
    librosa.load(file, sr=16000, duration=10) (classification/vggsound/dataset/vggsound.py:114), whose default resampler
    (soxr) is a different filter design and is absent here: parity with librosa is UNPINNED."""
    import math
    import scipy.signal
    g = math.gcd(int(orig_sr), int(target_sr))
    return scipy.signal.resample_poly(np.asarray(x, dtype=np.float32), int(target_sr) // g, int(orig_sr) // g, axis=-1)


# ---------------------------------------------------------------------------------------------
# patch embedding (classification/vggsound/models/vit_model.py:179-199, 435)
def patch_embed(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, compute_dtype=torch.float16,
                out_dtype=torch.float32, patch: int = 16) -> torch.Tensor:
    """`self.proj(x).flatten(2).transpose(1, 2)` (vit_model.py:198) for proj = Conv2d(1, E, 16, stride 16) (vit_model.py:435),
    x (B, 1, H, W) or (B, H, W) -> (B, P, E).  compute_dtype None: the plain float32 convolution, as the module computes outside
    autocast.  Otherwise the autocast evaluation (vit_model.py:388): image, weight and bias rounded to compute_dtype, products
    of those values summed in higher precision (float64 here, so that the checker's own summation error stays out of the
    comparison), result rounded to out_dtype."""
    if x.dim() == 3:
        x = x[:, None]
    if weight.dim() == 2:
        weight = weight.reshape(weight.shape[0], 1, patch, patch)
    if compute_dtype is None:
        y = torch.nn.functional.conv2d(x.float(), weight.float(), None if bias is None else bias.float(), stride=patch)
    else:
        r = lambda t: t.to(compute_dtype).to(torch.float64)
        y = torch.nn.functional.conv2d(r(x.float()), r(weight.float()), None if bias is None else r(bias.float()), stride=patch)
    return y.flatten(2).transpose(1, 2).to(out_dtype)
