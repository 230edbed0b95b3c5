#!/usr/bin/env python3
"""A/B timing of several builds of libmmfe.so on the patch embedding (4096 spectrograms of 384 x 128 -> 192 x 768 tokens), in ONE
process with interleaved rounds, as scripts/ab_fbank.py does for the transforms.

    python scripts/ab_patch.py [--out float32|float16|bfloat16] [--compute float16|bfloat16] [--batch 4096] [--rounds 6] [--launches 20] name=path.so ...
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multi_modal_amd import _native as N   # noqa: E402

DT = {"float32": (torch.float32, N.OUT_F32), "float16": (torch.float16, N.OUT_F16), "bfloat16": (torch.bfloat16, N.OUT_BF16)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="float32")
    ap.add_argument("--compute", default="float16")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--embed", type=int, default=768)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B, H, W, E = a.batch, a.height, a.width, a.embed
    x = torch.randn(B, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    w = (torch.randn(E, 256, generator=torch.Generator().manual_seed(7)) * 0.02).numpy()
    bias = (torch.randn(E, generator=torch.Generator().manual_seed(8)) * 0.1).numpy()
    P = (H // 16) * (W // 16)
    names, hs, outs = [], [], []
    for b in a.builds:
        name, path = b.split("=", 1)
        N._lib = N.bind_library(os.path.abspath(path))
        names.append(name)
        hs.append(N.PatchEmbedHandle(H, W, w, bias, DT[a.compute][1], DT[a.out][1], device=0))
        outs.append(torch.zeros(B, P, E, device=dev, dtype=DT[a.out][0]))
    stream = torch.cuda.current_stream().cuda_stream

    def launch(i):
        hs[i].run_ptr(x.data_ptr(), H * W, B, outs[i].data_ptr(), P * E, stream)

    for i in range(len(hs)):
        launch(i)
    torch.cuda.synchronize()
    diffs = [float((outs[i].float() - outs[0].float()).abs().max()) for i in range(len(hs))]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for i in range(len(hs)):
            for _ in range(10):
                launch(i)
        torch.cuda.synchronize()
    per = [[] for _ in hs]
    rmed = [[] for _ in hs]
    for r in range(a.rounds):
        order = list(range(len(hs)))
        if r % 2:
            order.reverse()
        for i in order:
            for _ in range(3):
                launch(i)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
            for s, e in evs:
                s.record()
                launch(i)
                e.record()
            torch.cuda.synchronize()
            ts = [s.elapsed_time(e) for s, e in evs]
            per[i] += ts
            rmed[i].append(statistics.median(ts))
    base = statistics.median(rmed[0])
    gb = (B * H * W * 4 + B * P * E * outs[0].element_size()) / 1e9
    print(f"# patch embedding ({B}, {H}, {W}) -> ({B}, {P}, {E}) {a.out}, {a.compute} products; {gb:.3f} GB algorithmic per launch")
    print(f"{'build':<20} {'median ms':>10} {'min ms':>9} {'med of round medians':>21} {'vs first':>9} {'TB/s':>7} {'max|d| vs first':>16}")
    for i, name in enumerate(names):
        m = statistics.median(rmed[i])
        print(f"{name:<20} {statistics.median(per[i]):>10.4f} {min(per[i]):>9.4f} {m:>21.4f} {m / base:>9.3f} {gb / m:>7.2f} {diffs[i]:>16.3e}", flush=True)


if __name__ == "__main__":
    main()
