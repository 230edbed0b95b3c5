#!/usr/bin/env python3
"""Why does the same kernel time differently in bench.py and in scripts/ab_fbank.py?  One process, one library:
the bench-style loop (wrapper call, 100 launches) and the A/B-style loop (run_ptr, 20 launches) over the same
buffers, then the output buffer at several offsets from a large arena and the input at several offsets.

    python scripts/bench_vs_ab.py
"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multi_modal_amd as mm  # noqa: E402
from multi_modal_amd import kaldi, vggsound  # noqa: E402


def timed(fn, n):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return statistics.median(ts), ts[0], sum(ts) / len(ts)


def main():
    mm.load_library()
    dev = torch.device("cuda", 0)
    B, n = 4096, 160000
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.empty(B, n, device=dev)
    slab = max(1, (1 << 30) // n)
    for lo in range(0, B, slab):
        hi = min(B, lo + slab)
        x[lo:hi] = (0.1 * torch.randn(hi - lo, n, device=dev, generator=g)).clamp_(-1, 1)
    out = torch.empty(B, 384, 128, device=dev)
    print(f"x at {x.data_ptr():#x}  out at {out.data_ptr():#x}", flush=True)
    kw = dict(vggsound._KW)
    plan = kaldi._fbank_plan(0, sample_frequency=16000, frame_length=25.0, frame_shift=kw["frame_shift"], num_mel_bins=128,
                             window_type="hanning", blackman_coeff=0.42, preemphasis_coefficient=0.97, remove_dc_offset=True,
                             round_to_power_of_two=True, low_freq=20.0, high_freq=0.0, use_log_fbank=True, out_frames=384,
                             pad_value=0.0, out_shift=kw["out_shift"], out_scale=kw["out_scale"])
    stream = torch.cuda.current_stream().cuda_stream

    def wrapper():
        mm.preprocessing_audio_batch(x, out=out)

    def direct():
        plan.run_ptr(x.data_ptr(), x.stride(0), B, n, out.data_ptr(), stream)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(20):
            direct()
        torch.cuda.synchronize()
    for rnd in range(3):
        for name, fn, k in (("wrapper x100", wrapper, 100), ("direct x100", direct, 100), ("direct x20", direct, 20),
                            ("wrapper x20", wrapper, 20)):
            med, mn, avg = timed(fn, k)
            print(f"round {rnd} {name:<14} median {med:.4f}  min {mn:.4f}  avg {avg:.4f} ms", flush=True)
    # sustained: 1000 launches, time in blocks of 100
    for blk in range(8):
        med, mn, avg = timed(direct, 100)
        print(f"sustained block {blk}: median {med:.4f} min {mn:.4f} avg {avg:.4f}", flush=True)
    # output placement
    arena = torch.empty((B * 384 * 128 * 4 + (64 << 20)) // 4, device=dev)
    for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, 33 << 20):
        o = arena[off // 4: off // 4 + B * 384 * 128]

        def f():
            plan.run_ptr(x.data_ptr(), x.stride(0), B, n, o.data_ptr(), stream)
        for _ in range(5):
            f()
        med, mn, avg = timed(f, 40)
        print(f"out offset {off:>9}: at {o.data_ptr():#x} median {med:.4f} min {mn:.4f} avg {avg:.4f}", flush=True)
    del arena
    xa = torch.empty(B * n + (64 << 20) // 4, device=dev)
    for off in (0, 256, 4096, 1 << 20, (2 << 20) + 4096, 33 << 20):
        xv = xa[off // 4: off // 4 + B * n].view(B, n)
        xv.copy_(x)

        def f():
            plan.run_ptr(xv.data_ptr(), n, B, n, out.data_ptr(), stream)
        for _ in range(5):
            f()
        med, mn, avg = timed(f, 40)
        print(f"in offset {off:>9}: at {xv.data_ptr():#x} median {med:.4f} min {mn:.4f} avg {avg:.4f}", flush=True)


if __name__ == "__main__":
    main()
