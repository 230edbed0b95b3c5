"""Rate of the fused log-mel path when the caller hands over HOST batches, for float32 clips and for the 16-bit PCM the
audio files hold: (a) copy + kernel on one stream from a pinned buffer, (b) through loader.DevicePrefetcher (pageable
DataLoader-style batches -> pinned staging -> copy stream, overlapped with the kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402
from multi_modal_amd.loader import DevicePrefetcher  # noqa: E402

B, N_ITER = 1024, 10
wave = (0.1 * torch.randn(B, 160000)).clamp_(-1, 1)
out = torch.empty(B, 384, 128, device="cuda")
for name, host in (("float32", wave), ("int16 PCM", (wave * 32767.0).round().to(torch.int16))):
    pinned = host.pin_memory()
    dev = torch.empty_like(pinned, device="cuda")
    for _ in range(3):
        dev.copy_(pinned, non_blocking=True)
        mm.preprocessing_audio_batch(dev, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N_ITER):
        dev.copy_(pinned, non_blocking=True)
        mm.preprocessing_audio_batch(dev, out=out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nbytes = host.element_size() * host.numel()
    print("%-9s pinned buffer, one stream:   %.1f clips/s, %.2f GB/s over PCIe" % (name, B * N_ITER / el, N_ITER * nbytes / el / 1e9))
    for depth in (2, 3):
        batches = [host] * (N_ITER + 3)
        it = iter(DevicePrefetcher(batches, depth=depth, transform=lambda b: mm.preprocessing_audio_batch(b, out=out)))
        for _ in range(3):
            next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in it:
            pass
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("%-9s DevicePrefetcher depth %d:     %.1f clips/s, %.2f GB/s over PCIe (pageable batches, host staging copy included)"
              % (name, depth, B * N_ITER / el, N_ITER * nbytes / el / 1e9))
