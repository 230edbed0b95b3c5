"""Launch time of the fused VGGSound log-mel kernel for both output layouts and float32 / bfloat16 output (batch 4096).
MMFE_FBANK_CHUNK (read at plan creation) sets how many consecutive frame quads a wave takes per visit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402

x = (0.1 * torch.randn(4096, 160000, device='cuda')).clamp_(-1, 1)


def t(fn, it=100):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for nchw in (False, True):
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty((4096, 128, 384) if nchw else (4096, 384, 128), device='cuda', dtype=dt)
        print("chunk", os.environ.get("MMFE_FBANK_CHUNK", "default"), "(B, 1, 128, 384)" if nchw else "(B, 384, 128)   ", dt,
              round(t(lambda: mm.preprocessing_audio_batch(x, out=out, dtype=dt, nchw=nchw)), 4), "ms", flush=True)
