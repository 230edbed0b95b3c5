#!/usr/bin/env python3
"""Front-end throughput benchmark: VGGSound 10 s 16 kHz clips -> (384, 128) normalised kaldi log-mel.

    python bench.py --gpus 1 --steps 100 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one pass of the fused front-end kernel over one device-resident batch of 4096 clips per GPU
(BASELINE.json configs[1]); inputs are synthetic (0.1*N(0,1) clipped to +-1, seed 1234+rank) and already
in HBM when the timed region starts.  Clips are independent, so ranks shard the batch with no data-path
collective (weak scaling); the optional all-gather of features is timed separately.  Rank 0 prints ONE
JSON line.  The warm-up / timed-iteration idiom mirrors classification/vggsound/latency.py:38-53.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIP_SAMPLES = 160000
OUT_FRAMES, N_MEL = 384, 128
# algorithmic HBM bytes per clip: the 384 kept frames read 384*400 samples once, 384*128 floats are written
BYTES_PER_CLIP = 384 * 400 * 4 + OUT_FRAMES * N_MEL * 4          # 811008
HBM_PEAK_GBS = 8000.0                                             # MI355X HBM3E peak


def cpu_baseline(budget_s: float = 12.0, max_clips: int = 200000):
    """The CPU restatement of the reference's per-clip front-end (oracle, kind 'port'), driven the way
    the reference drives it: one clip per call, one thread (a DataLoader worker is one process)."""
    from oracle import frontend_oracle as O
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1234)
    clips = (0.1 * torch.randn(32, 1, CLIP_SAMPLES, generator=g)).clamp_(-1, 1)
    O.preprocessing_audio(clips[0])          # warm-up
    n = 0
    t0 = time.perf_counter()
    while True:
        O.preprocessing_audio(clips[n % 32])
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_clips:
            break
    return {"value": round(n / el, 2), "unit": "clips/s", "cores": 1, "kind": "port",
            "sample": f"{n} clips of {CLIP_SAMPLES} samples, one clip per call (oracle.preprocessing_audio), "
                      f"{el:.1f} s on 1 of {os.cpu_count()} host cores"}


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/fbank_pmc_latest.json), or None."""
    p = os.path.join(ROOT, "profiles", "fbank_pmc_latest.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true", help="multi-GPU: skip the separately timed all-gather of features")
    ap.add_argument("--no-secondary", action="store_true", help="skip the HAR / long-utterance / EPIC / 16-bit PCM configurations")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--precondition", type=float, default=0.5, help="seconds of untimed launches before the warm-up steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("MMFE_BENCH_DIST", "0") == "1")
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    import multi_modal_amd as mm
    mm.load_library()

    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = (0.1 * torch.randn(B, CLIP_SAMPLES, device=dev, generator=g)).clamp_(-1, 1)
    out = torch.empty(B, OUT_FRAMES, N_MEL, device=dev)

    def step():
        mm.preprocessing_audio_batch(x, out=out)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-conditioning (not part of the W warm-up steps): a step lasts well under a millisecond, so a
    # handful of warm-up steps ends before the GPU has left its idle clock state; keep it busy for ~0.5 s first
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.precondition:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    # ---- timed region: exactly K steps ------------------------------------------------------------
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()          # recorded on torch's current stream, the stream the kernel is launched on
        step()
        ends[i].record()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)

    # ---- optional: all-gather of features (multi-GPU only), timed separately -------------------------
    gather = None
    if use_dist and not args.no_allgather:
        try:
            full = torch.empty(world * B, OUT_FRAMES, N_MEL, device=dev)
            for _ in range(2):
                dist.all_gather_into_tensor(full, out)
            fence()
            k = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(k):
                step()
                dist.all_gather_into_tensor(full, out)
            fence()
            el = time.perf_counter() - t0
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            recv = (world - 1) * out.numel() * 4
            gather = {"clips_per_s_with_allgather": round(world * B * k / el, 1), "ms_per_step": round(1e3 * el / k, 3),
                      "received_bytes_per_rank": recv, "gathered_GB_per_s_per_rank": round(recv * k / el / 1e9, 1), "steps": k}
            del full
        except Exception as exc:      # the headline measurement above is complete: report it whatever happens here
            gather = {"error": repr(exc)}

    if rank == 0:
        achieved = BYTES_PER_CLIP * B / (kernel_avg_ms * 1e-3) / 1e9
        traffic = load_traffic() if B == 4096 else None
        res = {
            "metric": "clips/sec (10 s 16 kHz -> log-mel) per node",
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "VGGSound 10 s 16 kHz mono clips -> kaldi 128-bin log-mel, cut to 384 frames, "
                                   "normalised (BASELINE configs[1])",
                       "batch_per_gpu": B, "clip_samples": CLIP_SAMPLES, "n_fft": 512, "win": 400, "hop": 400,
                       "n_mel": N_MEL, "out": [OUT_FRAMES, N_MEL], "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "fbank512_kernel", "kernel_avg_ms": round(kernel_avg_ms, 4),
                         "kernel_min_ms": round(kernel_ms[0], 4), "kernel_median_ms": round(kernel_ms[len(kernel_ms) // 2], 4),
                         "algorithmic_bytes_per_clip": BYTES_PER_CLIP, "clips_per_launch": B},
        }
        if gather is not None:
            res["allgather"] = gather
        if world == 1 and not args.no_secondary:
            # the other benchmark configurations that fit one GPU, a few launches each (not part of `value`)
            del x, out
            torch.cuda.empty_cache()
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from bench_configs import run_config
                res["secondary"] = [run_config(name, dev, 10) for name in ("har", "logmel", "epic", "pcm")]
            except Exception as exc:  # never let a secondary configuration take the headline line down
                res["secondary"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
