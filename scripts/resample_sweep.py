#!/usr/bin/env python3
"""Resampler throughput over the sample-rate pairs of the reference's corpora (and their inverses), 256 clips a launch, 16-bit and
float32 rows.  With --pad it times every ratio three times in one process: the plan's own choice of LDS layout, the padded layout
forced off and forced on (MMFE_RESAMPLE_PAD, read at plan creation).

    python scripts/resample_sweep.py [--pad] [--only ORIG:TARGET]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_modal_amd as mm  # noqa: E402

PAIRS = [(44100, 16000, 10), (48000, 16000, 10), (48000, 24000, 10), (22050, 16000, 10), (44100, 24000, 10), (16000, 44100, 4),
         (8000, 16000, 10), (16000, 22050, 8), (48000, 44100, 4), (44100, 48000, 4), (16000, 48000, 4), (32000, 16000, 10)]


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def main():
    settings = (None, "0", "1") if "--pad" in sys.argv else (None,)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for orig, target, secs in PAIRS:
        if only is not None and only != f"{orig}:{target}":
            continue
        for dt in (torch.int16, torch.float32):
            n = orig * secs
            x = (torch.randn(256, n, device="cuda") * 3000).to(dt) if dt == torch.int16 else torch.randn(256, n, device="cuda") * 0.1
            n_out = -(-n * target // orig)
            out = torch.empty(256, n_out, device="cuda")
            by = 256 * (n * x.element_size() + n_out * 4)
            cells = []
            for pad in settings:
                mm.plans.clear_plans()
                if pad is None:
                    os.environ.pop("MMFE_RESAMPLE_PAD", None)
                else:
                    os.environ["MMFE_RESAMPLE_PAD"] = pad
                ms = t(lambda: mm.resample_batch(x, orig, target, out=out))
                cells.append(f"{'plan' if pad is None else 'pad=' + pad} {ms:.3f} ms {by / ms / 1e9:.2f} TB/s")
            os.environ.pop("MMFE_RESAMPLE_PAD", None)
            print(f"{orig}->{target} {str(dt):14s} " + "   ".join(cells), flush=True)


if __name__ == "__main__":
    main()
