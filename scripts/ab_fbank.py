#!/usr/bin/env python3
"""A/B timing of several builds of libmmfe.so on one benchmark configuration, in ONE process with
interleaved rounds (separate processes add variance that looks like a kernel property).

    python scripts/ab_fbank.py [--config vgg|logmel|har|resample] [--batch N] [--rounds 6] [--launches 20] name=path.so[@VAR=VALUE,...] [name=path.so ...]

  vgg     VGGSound log-mel, 4096 clips of 160 000 samples (kernel fbank512)
  logmel  long-utterance log-mel, 1024 utterances of 480 000 samples (kernel rfft400 + clamp pass)
  har     IMU STFT, 262 144 windows of 6 x 192 samples (kernel stft32)

Every build gets its own plan; round r times `launches` launches of every build in turn with HIP events.
Prints per build: median / min of the per-launch times, the median of the per-round medians, and the maximum
difference of its output from the first build's (a variant that changes the arithmetic shows up here).
Variant libraries live under build/, which .gpurunignore lists (they are not part of the product): take the line out for an A/B run
on the GPU box.
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multi_modal_amd import _native as N          # noqa: E402
from multi_modal_amd import kaldi, plans, vggsound  # noqa: E402


def build_plan(path, device, in_dtype=N.IN_F32, config="vgg", ratio=(160, 441)):
    """The configuration's plan, created through the library at `path`."""
    N._lib = N.bind_library(path)
    plans._plans.clear()            # plans of the previous library stay alive in the caller's list
    if config == "logmel":
        from multi_modal_amd import logmel
        return logmel._plan(device, 400, 160, 80, 16000, False, stream=len(plans._plans) + id(path) % 1000)
    if config == "har":
        from multi_modal_amd import har
        return har._stft_plan(device, 32, 28)
    if config == "resample":
        from multi_modal_amd import resample
        return resample._plan(device, ratio[0], ratio[1], 0, in_dtype)
    kw = dict(vggsound._KW)
    plan = kaldi._fbank_plan(device, sample_frequency=kw["sample_frequency"], frame_length=25.0, frame_shift=kw["frame_shift"],
                             num_mel_bins=kw["num_mel_bins"], window_type=kw["window_type"], blackman_coeff=0.42,
                             preemphasis_coefficient=0.97, remove_dc_offset=True, round_to_power_of_two=True, low_freq=20.0,
                             high_freq=0.0, use_log_fbank=True, out_frames=kw["out_frames"], pad_value=kw["pad_value"],
                             out_shift=kw["out_shift"], out_scale=kw["out_scale"], in_dtype=in_dtype)
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("vgg", "logmel", "har", "resample"), default="vgg")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--samples", type=int, default=160000)
    ap.add_argument("--pcm", action="store_true", help="vgg / logmel / resample: 16-bit PCM input")
    ap.add_argument("--ratio", default="160/441", help="resample: up/down (rows of --samples samples unless 160/441: ten seconds at 44.1 kHz)")
    ap.add_argument("builds", nargs="+", help="name=path pairs")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    if args.config == "vgg":
        rows, n, oshape = args.batch or 4096, args.samples, (384, 128)
    elif args.config == "logmel":
        rows, n, oshape = args.batch or 1024, 480000, (80, 3000)
    elif args.config == "resample":
        ratio = tuple(int(v) for v in args.ratio.split("/"))
        n = 441000 if ratio == (160, 441) else args.samples
        rows, oshape = args.batch or 1024, (-(-n * ratio[0] // ratio[1]),)
    else:
        n = args.samples if args.samples != 160000 else 192          # --samples: window length (default 192)
        rows, oshape = 6 * (args.batch or 262144), (17, 1 + (n + 3) // 4)
    x = torch.empty(rows, n, device=dev)
    slab = max(1, (1 << 28) // n)
    for lo in range(0, rows, slab):
        x[lo:lo + slab] = (0.1 * torch.randn(min(slab, rows - lo), n, device=dev, generator=g)).clamp_(-1, 1)
    if args.pcm:
        x = (x * 32767.0).round().to(torch.int16)
    names, plist, outs = [], [], []
    for b in args.builds:
        name, path = b.split("=", 1)
        path, _, envs = path.partition("@")            # name=path@VAR=VALUE,VAR=VALUE: tuning variables read at plan creation
        saved = {}
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=", 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        names.append(name)
        plist.append(build_plan(os.path.abspath(path), 0, in_dtype=N.IN_S16 if args.pcm else N.IN_F32, config=args.config,
                                ratio=ratio if args.config == "resample" else None))
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
        outs.append(torch.empty((rows,) + oshape, device=dev))
    stream = torch.cuda.current_stream().cuda_stream

    def launch(i):
        plist[i].run_ptr(x.data_ptr(), x.stride(0), rows, n, outs[i].data_ptr(), stream)

    for i in range(len(plist)):
        launch(i)
    torch.cuda.synchronize()
    diffs = [float((outs[i] - outs[0]).abs().max()) for i in range(len(plist))]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:       # leave the idle clock state
        for i in range(len(plist)):
            for _ in range(10):
                launch(i)
        torch.cuda.synchronize()
    per = [[] for _ in plist]
    round_med = [[] for _ in plist]
    for r in range(args.rounds):
        order = list(range(len(plist)))
        if r % 2:
            order.reverse()
        for i in order:
            for _ in range(3):
                launch(i)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
            for s, e in evs:
                s.record()
                launch(i)
                e.record()
            torch.cuda.synchronize()
            ts = [s.elapsed_time(e) for s, e in evs]
            per[i] += ts
            round_med[i].append(statistics.median(ts))
    base = statistics.median(round_med[0])
    print(f"{'build':<24} {'median ms':>10} {'min ms':>9} {'med of round medians':>21} {'vs first':>9} {'max|d| vs first':>16}")
    for i, name in enumerate(names):
        m = statistics.median(round_med[i])
        print(f"{name:<24} {statistics.median(per[i]):>10.4f} {min(per[i]):>9.4f} {m:>21.4f} {m / base:>9.3f} {diffs[i]:>16.3e}", flush=True)


if __name__ == "__main__":
    main()
