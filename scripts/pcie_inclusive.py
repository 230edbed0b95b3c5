"""Rate of the fused log-mel path when the caller hands over pinned HOST buffers (copy + kernel, one stream)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm
B = 1024
x = (0.1 * torch.randn(B, 160000)).clamp_(-1, 1).pin_memory()
xd = torch.empty(B, 160000, device='cuda'); out = torch.empty(B, 384, 128, device='cuda')
for _ in range(3):
    xd.copy_(x, non_blocking=True); mm.preprocessing_audio_batch(xd, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    xd.copy_(x, non_blocking=True); mm.preprocessing_audio_batch(xd, out=out)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("host->device inclusive: %.1f clips/s, %.2f GB/s over PCIe" % (B * n / el, B * n * 640000 / el / 1e9))
