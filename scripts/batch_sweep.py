"""Time per call against batch size, for the three fused transforms: GPU time per call from events around a run of calls, and
the host time one call takes to return (queueing only).  Small batches show the fixed cost per launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402


def t(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    s.record()
    for _ in range(it):
        fn()
    e.record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it, (h1 - h0) / it * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
which = sys.argv[1:] or ["vggsound", "har", "logmel", "epic"]
if "vggsound" in which:
    x = (0.1 * torch.randn(8192, 160000, device="cuda", generator=g)).clamp_(-1, 1)
    for b in (1, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
        ms, host = t(lambda: mm.preprocessing_audio_batch(x[:b]))
        print(f"vggsound B={b:6d}  {ms:8.4f} ms/call  host {host:7.4f} ms  {b * 811008 / ms / 1e6:8.1f} GB/s  {ms / b * 1e3:8.3f} us/clip", flush=True)
    del x
if "har" in which:
    h = torch.randn(262144, 6, 192, device="cuda", generator=g)
    for b in (64, 512, 4096, 16384, 65536, 131072, 262144):
        ms, host = t(lambda: mm.stft_mag(h[:b]))
        print(f"har      B={b:6d}  {ms:8.4f} ms/call  host {host:7.4f} ms  {b * 24600 / ms / 1e6:8.1f} GB/s", flush=True)
    del h
if "logmel" in which:
    u = (0.1 * torch.randn(1024, 480000, device="cuda", generator=g)).clamp_(-1, 1)
    for b in (1, 4, 16, 64, 256, 512, 1024):
        ms, host = t(lambda: mm.logmel.log_mel_spectrogram(u[:b]))
        print(f"logmel   B={b:6d}  {ms:8.4f} ms/call  host {host:7.4f} ms  {b * 2880000 / ms / 1e6:8.1f} GB/s", flush=True)
    del u
if "epic" in which:
    e = (0.1 * torch.randn(4096, 30696, device="cuda", generator=g)).clamp_(-1, 1)
    for b in (1, 16, 64, 256, 1024, 4096):
        ms, host = t(lambda: mm.log_specgram_batch(e[:b], div255=True))
        print(f"epic     B={b:6d}  {ms:8.4f} ms/call  host {host:7.4f} ms  {b * (30696 + 65536) * 4 / ms / 1e6:8.1f} GB/s", flush=True)
