"""Drop-in for torchaudio.compliance.kaldi.fbank on MI355X.

Same call signature and defaults as the function the reference calls at
classification/vggsound/dataset/vggsound.py:45-47.  Argument combinations the reference never uses and
the kernels do not implement raise NotImplementedError instead of silently computing something else.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as N
from . import melbank
from .plans import IN_CODES, OUT_CODES, cached_plan, run_rows, to_device_rows

_EPS = float(torch.finfo(torch.float32).eps)


def _next_pow2(x: int) -> int:
    return 1 if x == 0 else 2 ** (x - 1).bit_length()


def _fbank_plan(device: int, *, sample_frequency: float, frame_length: float, frame_shift: float, num_mel_bins: int,
                window_type: str, blackman_coeff: float, preemphasis_coefficient: float, remove_dc_offset: bool,
                round_to_power_of_two: bool, low_freq: float, high_freq: float, use_log_fbank: bool,
                out_frames: int = 0, pad_value: float = 0.0, out_shift: float = 0.0, out_scale: float = 1.0,
                out_dtype: int = N.OUT_F32, in_dtype: int = N.IN_F32, feature_major: bool = False) -> N.Plan:
    key = ("fbank", float(sample_frequency), float(frame_length), float(frame_shift), int(num_mel_bins), window_type,
           float(blackman_coeff), float(preemphasis_coefficient), bool(remove_dc_offset), bool(round_to_power_of_two),
           float(low_freq), float(high_freq), bool(use_log_fbank), int(out_frames), float(pad_value), float(out_shift),
           float(out_scale), int(out_dtype), int(in_dtype), bool(feature_major))

    def build(dev: int) -> N.Plan:
        shift = int(sample_frequency * frame_shift * 0.001)
        size = int(sample_frequency * frame_length * 0.001)
        if size < 2:
            raise ValueError("choose a window size {} that is [2, {}]".format(size, "len(waveform)"))
        if shift <= 0:
            raise ValueError("`window_shift` must be greater than 0")
        padded = _next_pow2(size) if round_to_power_of_two else size
        if padded % 2 != 0:
            raise ValueError("the padded `window_size` must be divisible by two. use `round_to_power_of_two` or change `frame_length`")
        bank = melbank.kaldi_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq)
        bank = torch.nn.functional.pad(bank, (0, 1)).numpy()      # the Nyquist bin gets zero weight
        bs, bl, bw = melbank.dense_to_bands(bank)
        spec = N.make_spec(kind=N.KIND_FBANK, n_fft=padded, win_len=size, hop=shift, pad_mode=N.PAD_NONE,
                           frames_mode=N.FRAMES_SNIP, n_mel=num_mel_bins, remove_dc=int(remove_dc_offset),
                           preemph=float(preemphasis_coefficient), power=2,
                           log_kind=N.LOG_LN_MAX if use_log_fbank else N.LOG_NONE, log_floor=_EPS,
                           out_shift=out_shift, out_scale=out_scale, out_frames=out_frames, pad_value=pad_value,
                           layout=N.LAYOUT_FEAT_MAJOR if feature_major else N.LAYOUT_TIME_MAJOR, out_dtype=out_dtype, in_dtype=in_dtype)
        win = melbank.kaldi_window(window_type, size, blackman_coeff)
        return N.Plan(spec, win, bs, bl, bw, device=dev)

    return cached_plan(key, device, build)


def _check_supported(dither, use_energy, vtln_warp, snip_edges, subtract_mean, use_power):
    if dither != 0.0:
        raise NotImplementedError("fbank: dither != 0 is not implemented (the reference passes dither=0.0)")
    if use_energy:
        raise NotImplementedError("fbank: use_energy=True is not implemented (the reference passes use_energy=False)")
    if vtln_warp != 1.0:
        raise NotImplementedError("fbank: vtln_warp != 1.0 is not implemented")
    if not snip_edges:
        raise NotImplementedError("fbank: snip_edges=False is not implemented")
    if subtract_mean:
        raise NotImplementedError("fbank: subtract_mean=True is not implemented")
    if not use_power:
        raise NotImplementedError("fbank: use_power=False is not implemented")


def fbank_batch(waveforms: torch.Tensor, *, out_frames: int = 0, pad_value: float = 0.0, out_shift: float = 0.0,
                out_scale: float = 1.0, out: Optional[torch.Tensor] = None, blackman_coeff: float = 0.42,
                frame_length: float = 25.0, frame_shift: float = 10.0, high_freq: float = 0.0, low_freq: float = 20.0,
                num_mel_bins: int = 23, preemphasis_coefficient: float = 0.97, remove_dc_offset: bool = True,
                round_to_power_of_two: bool = True, sample_frequency: float = 16000.0, use_log_fbank: bool = True,
                window_type: str = "povey", dtype: torch.dtype = torch.float32, lengths=None,
                feature_major: bool = False) -> torch.Tensor:
    """Batched form: (B, n) -> (B, m, num_mel_bins), one launch for the whole batch; feature_major=True stores (B, num_mel_bins, m)
    instead, the (frequency, time) image a patch convolution takes.  With out_frames > 0
    each clip is cut / padded (pad_value) to out_frames frames and (x + out_shift) * out_scale is applied.
    dtype=torch.float16 / torch.bfloat16 stores the float32 result rounded to nearest even (what a consumer
    under autocast would cast it to); only the power-of-two 512-point path implements it.
    lengths (one integer per clip, <= n): a ragged batch, clip r is waveforms[r, :lengths[r]] (the tensor is padded to the longest
    clip); frames a shorter clip does not have carry pad_value, as if out_frames had padded that clip alone.
    An int16 tensor is taken as 16-bit PCM (value s / 32768, what torchaudio.load makes of the reference's FLAC
    files, vggsound.py:66) and decoded inside the kernel: same bits out as for the decoded float32 waveform."""
    if dtype not in OUT_CODES:
        raise ValueError("dtype must be torch.float32, torch.float16 or torch.bfloat16")
    x, origin = to_device_rows(waveforms, pcm_ok=True)
    plan = _fbank_plan(x.device.index, sample_frequency=sample_frequency, frame_length=frame_length,
                       frame_shift=frame_shift, num_mel_bins=num_mel_bins, window_type=window_type,
                       blackman_coeff=blackman_coeff, preemphasis_coefficient=preemphasis_coefficient,
                       remove_dc_offset=remove_dc_offset, round_to_power_of_two=round_to_power_of_two,
                       low_freq=low_freq, high_freq=high_freq, use_log_fbank=use_log_fbank, out_frames=out_frames,
                       pad_value=pad_value, out_shift=out_shift, out_scale=out_scale, out_dtype=OUT_CODES[dtype],
                       in_dtype=IN_CODES[x.dtype], feature_major=feature_major)
    y = run_rows(plan, x, out, lengths)
    return y if origin.type == "cuda" else y.to(origin)


def fbank(waveform: torch.Tensor, blackman_coeff: float = 0.42, channel: int = -1, dither: float = 0.0,
          energy_floor: float = 1.0, frame_length: float = 25.0, frame_shift: float = 10.0, high_freq: float = 0.0,
          htk_compat: bool = False, low_freq: float = 20.0, min_duration: float = 0.0, num_mel_bins: int = 23,
          preemphasis_coefficient: float = 0.97, raw_energy: bool = True, remove_dc_offset: bool = True,
          round_to_power_of_two: bool = True, sample_frequency: float = 16000.0, snip_edges: bool = True,
          subtract_mean: bool = False, use_energy: bool = False, use_log_fbank: bool = True, use_power: bool = True,
          vtln_high: float = -500.0, vtln_low: float = 100.0, vtln_warp: float = 1.0,
          window_type: str = "povey") -> torch.Tensor:
    """(c, n) waveform -> (m, num_mel_bins) log-mel filterbank of channel max(channel, 0)."""
    _check_supported(dither, use_energy, vtln_warp, snip_edges, subtract_mean, use_power)
    if waveform.dim() != 2:
        raise ValueError(f"fbank expects a (channels, samples) tensor, got {tuple(waveform.shape)}")
    channel = max(channel, 0)
    if channel >= waveform.shape[0]:
        raise ValueError("Invalid channel {} for size {}".format(channel, waveform.shape[0]))
    row = waveform[channel:channel + 1, :]
    size = int(sample_frequency * frame_length * 0.001)
    n = row.shape[1]
    if n < min_duration * sample_frequency or n < size:
        return torch.empty((0, 0) if n >= min_duration * sample_frequency else (0,), dtype=torch.float32, device=waveform.device)
    y = fbank_batch(row, blackman_coeff=blackman_coeff, frame_length=frame_length, frame_shift=frame_shift,
                    high_freq=high_freq, low_freq=low_freq, num_mel_bins=num_mel_bins,
                    preemphasis_coefficient=preemphasis_coefficient, remove_dc_offset=remove_dc_offset,
                    round_to_power_of_two=round_to_power_of_two, sample_frequency=sample_frequency,
                    use_log_fbank=use_log_fbank, window_type=window_type)
    return y[0]
