"""VGGSound audio front-end: drop-in for VGGSound.preprocessing_audio
(classification/vggsound/dataset/vggsound.py:44-61) with pad/cut and normalisation fused into the kernel."""
from __future__ import annotations

from typing import Optional

import torch

from .kaldi import fbank_batch

TARGET_LENGTH = 384        # vggsound.py:48
NORM_MEAN = -5.1991        # vggsound.py:59
NORM_STD = 4.4049          # vggsound.py:59
SAMPLE_RATE = 16000
NUM_MEL_BINS = 128
FRAME_SHIFT_MS = 25        # vggsound.py:47 (frame_shift=25 -> hop 400 samples)

_KW = dict(sample_frequency=SAMPLE_RATE, window_type="hanning", num_mel_bins=NUM_MEL_BINS, frame_shift=FRAME_SHIFT_MS,
           out_frames=TARGET_LENGTH, pad_value=0.0, out_shift=-NORM_MEAN, out_scale=1.0 / NORM_STD)


def preprocessing_audio_batch(audio: torch.Tensor, out: Optional[torch.Tensor] = None,
                              dtype: torch.dtype = torch.float32, lengths=None, nchw: bool = False) -> torch.Tensor:
    """(B, n) or (B, 1, n) waveforms -> (B, 384, 128) normalised log-mel, one launch per batch.
    nchw=True stores (B, 1, 128, 384) = (batch, channel, frequency, time) instead: the image layout of spectrogram encoders that
    put frequency on the height axis, written directly in that order (`out`, when given, is (B, 128, 384)).  The reference's
    own encoders take the time-major tensor as it is: `audio.unsqueeze(1)` (classification/vggsound/models/dynamicvit_runtime.py:110)
    is (B, 1, 384, 128), a view of the default output.
    lengths: clips of different lengths in one batch (each row of `audio` padded to the longest); every clip is padded /
    cut to 384 frames on its own, as vggsound.py:48-56 does per sample.
    dtype=torch.float16 hands the encoder the tensor its @autocast() would cast to
    (classification/vggsound/models/vit_model.py:388), without a float32 round trip through HBM.
    This is the call that replaces `batch[k].to(device)` of a collated spectrogram batch
    (classification/vggsound/vanilla_multi.py:90): workers hand over raw waveforms instead."""
    if audio.dim() == 3:
        if audio.shape[1] != 1:
            raise ValueError("expected mono clips (B, 1, n)")
        audio = audio[:, 0, :]
    if audio.dim() != 2:
        raise ValueError(f"expected (B, n) or (B, 1, n), got {tuple(audio.shape)}")
    y = fbank_batch(audio, out=out, dtype=dtype, lengths=lengths, feature_major=nchw, **_KW)
    return y.unsqueeze(1) if nchw else y


def preprocessing_audio(audio: torch.Tensor) -> torch.Tensor:
    """(1, n) waveform -> (384, 128); a (B, 1, n) batch -> (B, 384, 128)."""
    if audio.dim() == 3:
        return preprocessing_audio_batch(audio)
    if audio.dim() != 2:
        raise ValueError(f"expected a (channels, samples) waveform, got {tuple(audio.shape)}")
    return preprocessing_audio_batch(audio[0:1, :])[0]
