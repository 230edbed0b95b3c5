"""Launch time of the fused log-mel kernel with float32, float16 and bfloat16 output (batch 4096)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm
x = (0.1 * torch.randn(4096, 160000, device='cuda')).clamp_(-1, 1)
for dt in (torch.float32, torch.float16, torch.bfloat16):
    out = torch.empty(4096, 384, 128, device='cuda', dtype=dt)
    for _ in range(200): mm.preprocessing_audio_batch(x, out=out, dtype=dt)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): mm.preprocessing_audio_batch(x, out=out, dtype=dt)
    e.record(); torch.cuda.synchronize()
    print(dt, round(s.elapsed_time(e) / 200, 4), "ms")
