#!/usr/bin/env python3
"""Per-call time of the fused VGGSound log-mel path at several batch sizes (device-resident input)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402

for B in (1, 8, 32, 128, 512, 4096):
    x = (0.1 * torch.randn(B, 160000, device="cuda")).clamp_(-1, 1)
    out = torch.empty(B, 384, 128, device="cuda")
    for _ in range(20):
        mm.preprocessing_audio_batch(x, out=out)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        mm.preprocessing_audio_batch(x, out=out)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    print(f"B={B:5d}: {el * 1e6:9.1f} us per call, {B / el:12.0f} clips/s")
