import sys, torch
sys.path.insert(0, '/root/repo')
import multi_modal_amd as mm
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for orig, target, secs in [(44100, 16000, 10), (48000, 16000, 10), (48000, 24000, 10), (22050, 16000, 10), (44100, 24000, 10), (16000, 44100, 4), (8000, 16000, 10)]:
    for dt in (torch.int16, torch.float32):
        n = orig * secs
        x = (torch.randn(256, n, device='cuda') * 3000).to(dt) if dt == torch.int16 else torch.randn(256, n, device='cuda') * 0.1
        n_out = -(-n * target // orig)
        out = torch.empty(256, n_out, device='cuda')
        ms = t(lambda: mm.resample_batch(x, orig, target, out=out))
        by = 256 * (n * x.element_size() + n_out * 4)
        print(f"{orig}->{target} {str(dt):14s} {ms:.3f} ms  {by / ms / 1e9:.2f} TB/s  {256 / ms:.1f} k clips/s", flush=True)
