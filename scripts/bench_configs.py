#!/usr/bin/env python3
"""Throughput of the secondary configurations (BASELINE.json configs[2], the EPIC parity configuration,
configs[4] per GPU and the headline configuration fed with 16-bit PCM) on one GPU: units/s, achieved algorithmic GB/s and fraction of the 8 TB/s HBM peak.

    python scripts/bench_configs.py [har] [epic] [logmel] [pcm] [resample] [patch] [patch16] [tokens] [--iters 20]

bench.py imports `run_config` to append the same lines to its JSON under "secondary".
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402


def timeit(fn, iters, warmup=3, precondition=0.3):
    """Average and minimum launch time in ms over `iters` launches, after `precondition` seconds of untimed launches: the first
    launches after an idle period or after another configuration run at a different clock and say little about the steady rate."""
    import time
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < precondition:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    if os.environ.get("MMFE_BENCH_VERBOSE"):
        print("sorted ms:", " ".join(f"{m:.3f}" for m in ms), file=sys.stderr)
    return sum(ms) / len(ms), ms[0]


def _line(config, unit, n, per, avg, mn):
    rate = n / (avg * 1e-3)
    return {"config": config, "value": round(rate, 1), "unit": unit, "ms": round(avg, 4), "min_ms": round(mn, 4),
            "algorithmic_bytes_per_unit": per, "GB/s": round(per * rate / 1e9, 1), "frac_of_8TBs": round(per * rate / 8e12, 4)}


def run_config(name: str, dev: torch.device, iters: int = 20) -> dict:
    g = torch.Generator(device=dev).manual_seed(4321)
    if name == "har":
        B = 262144
        x = torch.randn(B, 6, 192, device=dev, generator=g)
        out = torch.empty(B, 6, 17, 49, device=dev)
        avg, mn = timeit(lambda: mm.stft_mag(x, out=out), iters)
        return _line("HAR (262144, 6, 192) -> (262144, 6, 17, 49), kernel stft32", "windows/s", B, 6 * 192 * 4 + 6 * 17 * 49 * 4, avg, mn)
    if name == "epic":
        B = 4096
        x = (0.1 * torch.randn(B, 30672, device=dev, generator=g)).clamp_(-1, 1)
        out = torch.empty(B, 256, 256, device=dev)
        avg, mn = timeit(lambda: mm.log_specgram_batch(x, div255=True, out=out), iters)
        line = _line("EPIC (4096, 30672) -> (4096, 256, 256), n_fft 511, kernel dftgen, float64 matrix-pipe form (compute-bound)",
                     "clips/s", B, 30672 * 4 + 256 * 256 * 4, avg, mn)
        # 256 frames x 256 bins x 120 mirrored tap pairs x 2 products x 2 flop per clip, against the 78.6 TFLOP/s float64 peak
        flop = 256 * 256 * 120 * 2 * 2
        line["TFLOP/s_f64"] = round(flop * line["value"] / 1e12, 1)
        line["frac_of_78.6_TFLOPs"] = round(flop * line["value"] / 78.6e12, 4)
        return line
    if name == "logmel":
        B = 1024
        x = (0.1 * torch.randn(B, 480000, device=dev, generator=g)).clamp_(-1, 1)
        out = torch.empty(B, 80, 3000, device=dev)
        avg, mn = timeit(lambda: mm.logmel.log_mel_spectrogram(x, out=out), iters)
        return _line("long-utterance log-mel (1024, 480000) -> (1024, 80, 3000), n_fft 400, kernel " + mm.plans.last_kernel(),
                     "utterances/s", B, 480000 * 4 + 80 * 3000 * 4, avg, mn)
    if name == "pcm":
        B = 4096
        x = (0.1 * torch.randn(B, 160000, device=dev, generator=g)).clamp_(-1, 1).mul_(32767.0).round_().to(torch.int16)
        out = torch.empty(B, 384, 128, device=dev)
        avg, mn = timeit(lambda: mm.preprocessing_audio_batch(x, out=out), iters)
        return _line("VGGSound log-mel from 16-bit PCM (4096, 160000) int16 -> (4096, 384, 128), kernel " + mm.plans.last_kernel(),
                     "clips/s", B, 153600 * 2 + 384 * 128 * 4, avg, mn)
    if name == "resample":
        B = 1024
        x = (0.1 * torch.randn(B, 441000, device=dev, generator=g)).clamp_(-1, 1).mul_(32767.0).round_().to(torch.int16)
        out = torch.empty(B, 160000, device=dev)
        avg, mn = timeit(lambda: mm.resample_batch(x, 44100, 16000, out=out), iters)
        return _line("resampler 44.1 -> 16 kHz from 16-bit PCM (1024, 441000) int16 -> (1024, 160000), kernel resample",
                     "clips/s", B, 441000 * 2 + 160000 * 4, avg, mn)
    if name in ("patch", "patch16", "tokens"):
        # the consumer's first layer on the 4096 spectrograms of the headline configuration: (4096, 1, 384, 128) -> (4096, 192, 768)
        from multi_modal_amd.patch_embed import PatchEmbed
        B, E = 4096, 768
        w = torch.randn(E, 1, 16, 16, generator=torch.Generator().manual_seed(7)) * 0.02
        bias = torch.randn(E, generator=torch.Generator().manual_seed(8)) * 0.1
        odt = torch.float32 if name == "patch" else torch.float16
        pe = PatchEmbed((384, 128), w, bias, compute_dtype=torch.float16, out_dtype=odt)
        out = torch.empty(B, 192, E, device=dev, dtype=odt)
        flop = 2.0 * 192 * 256 * E
        if name == "tokens":       # waveform -> float16 spectrogram -> float16 tokens, two launches
            x = (0.1 * torch.randn(B, 160000, device=dev, generator=g)).clamp_(-1, 1)
            spec = torch.empty(B, 384, 128, device=dev, dtype=torch.float16)

            def both():
                mm.preprocessing_audio_batch(x, out=spec, dtype=torch.float16)
                pe(spec, out=out)
            avg, mn = timeit(both, iters)
            line = _line("VGGSound waveform -> float16 log-mel -> float16 tokens (4096, 160000) -> (4096, 192, 768), kernels fbank512 + patch_embed",
                         "clips/s", B, 153600 * 4 + 2 * 384 * 128 * 2 + 192 * E * 2, avg, mn)
        else:
            spec = torch.randn(B, 384, 128, device=dev, generator=g)
            avg, mn = timeit(lambda: pe(spec, out=out), iters)
            line = _line(f"patch embedding (4096, 1, 384, 128) float32 -> (4096, 192, 768) {str(odt).split('.')[-1]}, float16 products, kernel patch_embed",
                         "clips/s", B, 384 * 128 * 4 + 192 * E * out.element_size(), avg, mn)
        line["TFLOP/s"] = round(flop * B / (avg * 1e-3) / 1e12, 1)
        return line
    raise ValueError(name)


def main():
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    names = ("har", "epic", "logmel", "pcm", "resample", "patch", "patch16", "tokens")
    which = [a for a in sys.argv[1:] if a in names] or list(names)
    dev = torch.device("cuda", 0)
    for name in which:
        print(json.dumps(run_config(name, dev, iters)), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
