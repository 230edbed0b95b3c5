"""multi_modal_amd: MI355X-native signal front-end (framing, window, real FFT, |X|^2, mel, log) behind
the Python call signatures of lixinghe1999/multi_modal's dataset classes.

    kaldi.fbank                      torchaudio.compliance.kaldi.fbank            (vggsound.py:45-47)
    vggsound.preprocessing_audio     VGGSound.preprocessing_audio                 (vggsound.py:44-61)
    epic.log_specgram                EPICKitchen._log_specgram                    (epic_kitchen.py:114-126)
    epic.extract_sound_features      EPICKitchen._extract_sound_feature, batched  (epic_kitchen.py:128-154)
    har.stft_mag                     np.abs(scipy.signal.stft(...)[-1])           (adbox.py:20)
    transforms.frame_signal          frame_signal                                 (utils/transforms.py:19-61)
    stats.cal_norm                   cal_norm                                     (vggsound.py:85-107)
    resample.load_resampled          the resampling of librosa.load(sr=16000)     (vggsound.py:114; extract_audio.py:6-13)
    patch_embed.PatchEmbed           PatchEmbed.forward with the audio projection (models/vit_model.py:193-199, 435)
    loader.DevicePrefetcher          `batch[k].to(device)` of the training loop, overlapped    (vanilla_multi.py:76-77, 89-92)
    partition.*                      batch sharding over the GPUs of a node + all-gather of features

All arithmetic runs in libmmfe.so (hand-written HIP for gfx950, C ABI in include/mmfe.h).
"""
from . import _native, epic, har, kaldi, loader, logmel, melbank, partition, patch_embed, resample, stats, transforms, vggsound  # noqa: F401
from ._native import MmfeError, library_path, load_library  # noqa: F401
from .epic import log_specgram, log_specgram_batch, spec_to_tensor  # noqa: F401
from .har import stft_mag  # noqa: F401
from .resample import load_resampled, resample_batch  # noqa: F401
from .kaldi import fbank, fbank_batch  # noqa: F401
from .transforms import frame_signal  # noqa: F401
from .vggsound import preprocessing_audio, preprocessing_audio_batch  # noqa: F401

__all__ = [
    "MmfeError", "library_path", "load_library", "fbank", "fbank_batch", "preprocessing_audio",
    "preprocessing_audio_batch", "log_specgram", "log_specgram_batch", "spec_to_tensor", "stft_mag", "frame_signal", "resample_batch", "load_resampled",
]
