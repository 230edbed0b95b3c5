"""Rate of the fused log-mel path when the caller hands over pinned HOST buffers (copy + kernel, one stream),
for float32 clips and for the 16-bit PCM the audio files hold."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402

B, N_ITER = 1024, 10
wave = (0.1 * torch.randn(B, 160000)).clamp_(-1, 1)
out = torch.empty(B, 384, 128, device="cuda")
for name, host in (("float32", wave), ("int16 PCM", (wave * 32767.0).round().to(torch.int16))):
    host = host.pin_memory()
    dev = torch.empty_like(host, device="cuda")
    for _ in range(3):
        dev.copy_(host, non_blocking=True)
        mm.preprocessing_audio_batch(dev, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N_ITER):
        dev.copy_(host, non_blocking=True)
        mm.preprocessing_audio_batch(dev, out=out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nbytes = host.element_size() * host.numel()
    print("%-9s host->device inclusive: %.1f clips/s, %.2f GB/s over PCIe" % (name, B * N_ITER / el, N_ITER * nbytes / el / 1e9))
