"""Launch times of the less-travelled variants of every transform (ragged batches, slices, other layouts and dtypes, the
any-length kernel), to catch a path that is far off its family's regular rate.  One line per variant: ms and algorithmic GB/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402


def t(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def line(name, ms, nbytes):
    print(f"{name:70s} {ms:8.3f} ms {nbytes / ms / 1e6:9.1f} GB/s   [{mm.plans.last_kernel()}]", flush=True)


g = torch.Generator(device="cuda").manual_seed(1)
B = 2048
x = (0.1 * torch.randn(B, 160000, device="cuda", generator=g)).clamp_(-1, 1)
by = B * 811008
line("vggsound regular", t(lambda: mm.preprocessing_audio_batch(x)), by)
lens = torch.randint(100000, 160001, (B,), device="cuda", generator=g).to(torch.int32)
line("vggsound ragged lengths", t(lambda: mm.preprocessing_audio_batch(x, lengths=lens)), by)
line("vggsound ragged, all full length", t(lambda: mm.preprocessing_audio_batch(x, lengths=torch.full((B,), 160000, dtype=torch.int32, device="cuda"))), by)
line("vggsound unaligned rows (odd offset)", t(lambda: mm.preprocessing_audio_batch(x.flatten()[1:1 + (B - 1) * 160000].view(B - 1, 160000))), by)
line("vggsound feature-major", t(lambda: mm.preprocessing_audio_batch(x, nchw=True)), by)
line("vggsound 96000-sample clips (240 frames + padding)", t(lambda: mm.preprocessing_audio_batch(x[:, :96000])), B * (96000 * 4 + 384 * 128 * 4))
line("fbank povey 10 ms shift 80 bins (AST style)", t(lambda: mm.fbank_batch(x[:512], sample_frequency=16000, num_mel_bins=80, frame_shift=10.0)), 512 * (160000 * 4 + 998 * 80 * 4))
line("fbank n_fft 400 (round_to_power_of_two=False): any-length kernel", t(lambda: mm.fbank_batch(x[:128], sample_frequency=16000, num_mel_bins=80, frame_shift=10.0, round_to_power_of_two=False), 5), 128 * (160000 * 4 + 998 * 80 * 4))
e = (0.1 * torch.randn(2048, 30696, device="cuda", generator=g)).clamp_(-1, 1)
line("epic batch", t(lambda: mm.log_specgram_batch(e, div255=True)), 2048 * (30696 * 4 + 65536 * 4))
line("epic batch, mixed 30672 / 30696", t(lambda: mm.log_specgram_batch(e, div255=True, lengths=[30672 + 24 * (i & 1) for i in range(2048)])), 2048 * (30696 * 4 + 65536 * 4))
recs = [(0.1 * torch.randn(24000 * 60, device="cuda", generator=g)).clamp_(-1, 1) for _ in range(8)]
n = 2048
args = (recs, [i % 8 for i in range(n)], [int(v) for v in np.random.default_rng(0).integers(0, 3000, n)], [int(v) for v in np.random.default_rng(1).integers(0, 300, n)], [60.0] * n)
line("epic extract_sound_features (slices of packed recordings, incl. host index work)", t(lambda: mm.epic.extract_sound_features(*args), 5), n * (30696 * 4 + 65536 * 4))
u = (0.1 * torch.randn(256, 480000, device="cuda", generator=g)).clamp_(-1, 1)
line("logmel regular", t(lambda: mm.logmel.log_mel_spectrogram(u)), 256 * 2880000)
line("logmel bfloat16 out", t(lambda: mm.logmel.log_mel_spectrogram(u, dtype=torch.bfloat16)), 256 * (1920000 + 480000))
line("logmel 10 s utterances", t(lambda: mm.logmel.log_mel_spectrogram(u[:, :160000])), 256 * 960000)
line("logmel unaligned rows", t(lambda: mm.logmel.log_mel_spectrogram(u.flatten()[1:1 + 255 * 480000].view(255, 480000))), 255 * 2880000)
line("logmel float64 accumulation: any-length kernel", t(lambda: mm.logmel.log_mel_spectrogram(u[:16], precision="float64"), 3), 16 * 2880000)
h = torch.randn(65536, 6, 192, device="cuda", generator=g)
line("har regular", t(lambda: mm.stft_mag(h)), 65536 * 24600)
line("har T = 190 (unaligned rows: phase-serial kernel)", t(lambda: mm.stft_mag(h[:, :, :190].contiguous())), 65536 * 24600)
line("har nperseg 64 noverlap 48: any-length kernel", t(lambda: mm.stft_mag(h[:8192], nperseg=64, noverlap=48), 5), 8192 * 6 * (192 * 4 + 33 * 13 * 4))
f = torch.randn(64, 160000, device="cuda", generator=g)
line("frame_signal 400 / 160", t(lambda: mm.frame_signal(f, 400, 160)), 64 * (160000 * 4 + 999 * 400 * 4))
feats = mm.preprocessing_audio_batch(x)
line("cal_norm over the features", t(lambda: mm.stats.cal_norm(feats)), feats.numel() * 4)
