import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm
x = torch.randn(4096, 384, 128, device='cuda')
for _ in range(3): mm.stats.cal_norm(x)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): mm.stats.cal_norm(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("cal_norm over %.0f MB: %.3f ms per call, %.0f GB/s" % (x.numel() * 4 / 1e6, ms, x.numel() * 4 / ms / 1e6))
