#!/usr/bin/env python3
"""Front-end throughput benchmark.

    python bench.py --gpus 1 --steps 100 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config vgg4096|vgg200k|logmel]

A step is one pass of the fused front-end over one device-resident batch per GPU; inputs are synthetic
(0.1*N(0,1) clipped to +-1, seed 1234+rank) and already in HBM when the timed region starts.  Rows are
independent, so ranks shard the batch with no data-path collective (weak scaling); the all-gather of features
(RCCL over xGMI) is timed separately and reported under "allgather".  Rank 0 prints ONE JSON line.

  --config vgg4096  (default) VGGSound 10 s 16 kHz clips -> (384, 128) normalised kaldi log-mel, 4096 clips per GPU:
                    the configuration the headline metric is quoted on (BASELINE.json configs[1])
  --config vgg200k  the same transform on 25 000 clips per GPU: with --gpus 8 the 200 000-clip set of configs[3],
                    compute-only and compute + all-gather rates, gather GB/s against the 7 x 153 GB/s of xGMI
  --config logmel   30 s utterances, n_fft 400 / hop 160 / 80 mel, 1024 utterances per GPU (configs[4])

The warm-up / timed-iteration idiom mirrors classification/vggsound/latency.py:38-53.
"""
import argparse
import json
import os
import subprocess
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
UTT_SAMPLES, UTT_FRAMES, UTT_MEL = 480000, 3000, 80
BYTES_PER_UTT = UTT_SAMPLES * 4 + UTT_MEL * UTT_FRAMES * 4       # 2880000
HBM_PEAK_GBS = 8000.0                                             # MI355X HBM3E peak
XGMI_PEAK_GBS = 7 * 153.0                                         # 7 point-to-point links per GPU

CONFIGS = {
    "vgg4096": dict(batch=4096, samples=CLIP_SAMPLES, out=(OUT_FRAMES, N_MEL), bytes_per_unit=BYTES_PER_CLIP, unit="clips/s",
                    kernel="fbank512_kernel", metric="clips/sec (10 s 16 kHz -> log-mel) per node",
                    workload="VGGSound 10 s 16 kHz mono clips -> kaldi 128-bin log-mel, cut to 384 frames, normalised "
                             "(BASELINE configs[1])"),
    "vgg200k": dict(batch=25000, samples=CLIP_SAMPLES, out=(OUT_FRAMES, N_MEL), bytes_per_unit=BYTES_PER_CLIP, unit="clips/s",
                    kernel="fbank512_kernel", metric="clips/sec (10 s 16 kHz -> log-mel) per node",
                    workload="VGGSound-size set, 25 000 clips per GPU (200 000 over 8 GPUs), same transform, features "
                             "all-gathered over RCCL (BASELINE configs[3])"),
    "logmel": dict(batch=1024, samples=UTT_SAMPLES, out=(UTT_MEL, UTT_FRAMES), bytes_per_unit=BYTES_PER_UTT, unit="utterances/s",
                   kernel="rfft400_kernel (+ per-utterance clamp pass)", metric="utterances/sec (30 s 16 kHz -> 80-bin log-mel) per node",
                   workload="30 s utterances, n_fft 400 / hop 160 / 80 Slaney mel bands, log10, per-utterance clamp "
                            "(BASELINE configs[4])"),
}


# ---- CPU baseline: the oracle driven the way the reference drives its front-end -----------------------------------
def _cpu_loop(budget_s: float, max_clips: int = 10 ** 9):
    """One worker: the per-clip CPU restatement (oracle.preprocessing_audio), one thread, for `budget_s` seconds."""
    from oracle import frontend_oracle as O
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1234)
    clips = (0.1 * torch.randn(8, 1, CLIP_SAMPLES, generator=g)).clamp_(-1, 1)
    O.preprocessing_audio(clips[0])          # warm-up
    n = 0
    t0 = time.perf_counter()
    while True:
        O.preprocessing_audio(clips[n % 8])
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_clips:
            break
    return n, el


def _cpu_workers(n_workers: int, budget_s: float):
    """`n_workers` fresh single-threaded worker processes (the reference's DataLoader workers are processes,
    classification/vggsound/dataset/vggsound.py:87), each looping for `budget_s` seconds after its own start-up.
    Fresh processes, not forks: the parent has an initialised GPU context."""
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget_s)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(n_workers)]
    total, rate = 0, 0.0
    for p in procs:
        out = p.communicate(timeout=budget_s + 180)[0].decode()
        n, el = out.strip().splitlines()[-1].split()
        total += int(n)
        rate += int(n) / float(el)
    return total, rate


def host_core_share() -> int:
    """Cores this job may use: the box exports its share as OMP_NUM_THREADS (os.cpu_count() shows the whole machine)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, int(os.environ.get("OMP_NUM_THREADS", n)))
    except ValueError:
        pass
    return max(1, min(n, 16))


def _cpu_c_openmp(budget_s: float, threads: int, max_clips: int = 10 ** 9):
    """The compiled leg: oracle/fbank_ref.c (plain C, OpenMP over clips) on batches of 4 clips per thread, `threads` threads,
    for `budget_s` seconds.  The strongest CPU form of this path that exists here."""
    import numpy as np
    from oracle import fbank_ref
    rng = np.random.default_rng(1234)
    batch = min(4 * threads, max(1, max_clips))
    clips = np.clip(0.1 * rng.standard_normal((batch, CLIP_SAMPLES)), -1, 1).astype(np.float32)
    fbank_ref.preprocessing_audio_batch(clips[: min(batch, threads)], threads=threads)       # warm-up
    n = 0
    t0 = time.perf_counter()
    while True:
        fbank_ref.preprocessing_audio_batch(clips, threads=threads)
        n += batch
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_clips:
            break
    return n, el


def cpu_baseline(budget_s: float = 12.0, max_clips: int = 200000, legs=("1_thread", "4_workers", "all_cores", "c_openmp_all_cores")):
    """The CPU restatement of the reference's per-clip front-end (oracle, kind 'port') timed four ways: one thread,
    4 worker processes (how the reference runs it: DataLoader(num_workers=4), vggsound.py:87), one worker per
    available core, and the compiled C / OpenMP restatement on every available core (the best CPU number, batch calls).
    `value` / `cores` at the top level are the 4-worker leg when it ran, else the single thread."""
    per_leg = budget_s / max(1, len(legs))
    res = {"unit": "clips/s", "kind": "port", "legs": {}}
    if "1_thread" in legs:
        n, el = _cpu_loop(per_leg, max_clips)
        res["legs"]["1_thread"] = {"value": round(n / el, 2), "cores": 1, "clips": n, "seconds": round(el, 2)}
    if "4_workers" in legs:
        n, rate = _cpu_workers(4, per_leg)
        res["legs"]["4_workers"] = {"value": round(rate, 2), "cores": 4, "clips": n, "seconds": round(per_leg, 2)}
    if "all_cores" in legs:
        c = host_core_share()
        n, rate = _cpu_workers(c, per_leg)
        res["legs"]["all_cores"] = {"value": round(rate, 2), "cores": c, "clips": n, "seconds": round(per_leg, 2)}
    if "c_openmp_all_cores" in legs:
        c = host_core_share()
        try:
            n, el = _cpu_c_openmp(per_leg, c, max_clips)
            res["legs"]["c_openmp_all_cores"] = {"value": round(n / el, 2), "cores": c, "clips": n, "seconds": round(el, 2),
                                                 "what": "oracle/fbank_ref.c, gcc -O3 -mavx2 -fopenmp, batches of 4 clips per thread"}
        except Exception as exc:      # the library is built by __graft_entry__.build(); report rather than fail the line
            res["legs"]["c_openmp_all_cores"] = {"error": repr(exc)}
    head = "4_workers" if "4_workers" in res["legs"] else next(iter(res["legs"]))
    res["value"] = res["legs"][head]["value"]
    res["cores"] = res["legs"][head]["cores"]
    res["sample"] = (f"{res['legs'][head]['clips']} clips of {CLIP_SAMPLES} samples, one clip per call (oracle.preprocessing_audio), "
                     f"{res['legs'][head]['seconds']} s on {res['cores']} single-threaded worker process(es); host shows "
                     f"{os.cpu_count()} cores, share {host_core_share()}")
    return res


def load_traffic(kernel_avg_ms: float):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/fbank_pmc_latest.json).  Refused (None
    plus the reason) when that profile is of another kernel or its average launch time is more than 10 % away from
    this run's: the counters then describe a different build."""
    p = os.path.join(ROOT, "profiles", "fbank_pmc_latest.json")
    try:
        with open(p) as f:
            prof = json.load(f)
    except Exception as exc:
        return None, f"no profile: {exc!r}"
    if "fbank512_kernel" not in prof.get("kernel_name", ""):
        return None, "profile is of another kernel"
    prof_ms = float(prof.get("avg_ns", 0.0)) * 1e-6
    if prof_ms <= 0 or abs(prof_ms - kernel_avg_ms) > 0.10 * kernel_avg_ms:
        return None, f"profile is stale: its kernel averaged {prof_ms:.4f} ms, this run {kernel_avg_ms:.4f} ms"
    return prof.get("hbm_bytes_per_launch"), None


TRAFFIC_SOURCE = "profiles/fbank_pmc_latest.json (rocprofv3 --pmc passes of an earlier run of this command, replayed; not measured in this run)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vgg4096")
    ap.add_argument("--batch", type=int, default=0, help="units per GPU (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true", help="multi-GPU: skip the separately timed all-gather of features")
    ap.add_argument("--no-secondary", action="store_true", help="skip the HAR / long-utterance / EPIC / 16-bit PCM / resampler / patch-embedding configurations")
    ap.add_argument("--cpu-budget", type=float, default=16.0)
    ap.add_argument("--precondition", type=float, default=0.5, help="seconds of untimed launches before the warm-up steps")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker > 0:            # a worker process of the CPU baseline: never touches the GPU
        n, el = _cpu_loop(args.cpu_worker)
        print(n, el, flush=True)
        return

    cfg = CONFIGS[args.config]
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

    B = args.batch if args.batch > 0 else cfg["batch"]
    n = cfg["samples"]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.empty(B, n, device=dev)
    slab = max(1, (1 << 30) // n)                   # generated on the device in 4 GB slabs
    for lo in range(0, B, slab):
        hi = min(B, lo + slab)
        x[lo:hi] = (0.1 * torch.randn(hi - lo, n, device=dev, generator=g)).clamp_(-1, 1)
    out = torch.empty((B,) + cfg["out"], device=dev)

    if args.config == "logmel":
        def step():
            mm.logmel.log_mel_spectrogram(x, out=out)
    else:
        def step():
            mm.preprocessing_audio_batch(x, out=out)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-conditioning (not part of the W warm-up steps): a step lasts about a millisecond, so a
    # handful of warm-up steps ends before the GPU has left its idle clock state; keep it busy for ~0.5 s first
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.precondition:
        for _ in range(20):
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

    # ---- the all-gather of features (RCCL), timed separately: compute + gather, and the gather alone ------------
    gather = None
    if use_dist and not args.no_allgather:
        try:
            from multi_modal_amd import partition
            n_items = world * B
            full = torch.empty((n_items,) + cfg["out"], device=dev)
            recv = (world - 1) * out.numel() * 4
            k = max(3, min(args.steps, 10))
            modes = {}
            for mode in ("collective", "direct"):          # the library's all-gather, and shards sent point-to-point to every peer
                def gather_once(mode=mode):
                    partition.all_gather_features(out, n_items, out=full, mode=mode)
                for _ in range(2):
                    gather_once()
                fence()
                t0 = time.perf_counter()
                for _ in range(k):
                    step()
                    gather_once()
                fence()
                el = time.perf_counter() - t0
                t0 = time.perf_counter()
                for _ in range(k):
                    gather_once()
                fence()
                el_g = time.perf_counter() - t0
                t = torch.tensor([el, el_g], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el, el_g = (float(v) for v in t.tolist())
                modes[mode] = {cfg["unit"].replace("/s", "_per_s_with_allgather"): round(world * B * k / el, 1),
                               "ms_per_step_with_allgather": round(1e3 * el / k, 3), "allgather_ms": round(1e3 * el_g / k, 3),
                               "gathered_GB_per_s_per_rank": round(recv * k / el_g / 1e9, 1),
                               "frac_of_xgmi_7x153GBs": round(recv * k / el_g / 1e9 / XGMI_PEAK_GBS, 4)}
            # the same shard transformed and exchanged in 4 pieces, each piece's exchange under the next piece's transform
            try:
                fn = (lambda xs: mm.logmel.log_mel_spectrogram(xs)) if args.config == "logmel" else (lambda xs: mm.preprocessing_audio_batch(xs))
                for _ in range(2):
                    partition._gather_chunked(fn, x, n_items, rank, world, 4, None)
                fence()
                t0 = time.perf_counter()
                for _ in range(k):
                    partition._gather_chunked(fn, x, n_items, rank, world, 4, None)
                fence()
                t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
                modes["direct_4_pieces_overlapped"] = {cfg["unit"].replace("/s", "_per_s_with_allgather"): round(world * B * k / el, 1),
                                                       "ms_per_step_with_allgather": round(1e3 * el / k, 3)}
            except Exception as exc:
                modes["direct_4_pieces_overlapped"] = {"error": repr(exc)}
            picked = partition.pick_gather_mode("auto", world, True, True)
            gather = dict(modes[picked])
            gather.update({"received_bytes_per_rank": recv, "steps": k, "auto_picks": picked, "modes": modes,
                           "clips_per_s_with_allgather": modes[picked][cfg["unit"].replace("/s", "_per_s_with_allgather")]})
            del full
        except Exception as exc:      # the headline measurement above is complete: report it whatever happens here
            gather = {"error": repr(exc)}

    if rank == 0:
        achieved = cfg["bytes_per_unit"] * B / (kernel_avg_ms * 1e-3) / 1e9
        traffic, traffic_note = (None, "no PMC profile for this configuration")
        if args.config == "vgg4096" and B == 4096:
            traffic, traffic_note = load_traffic(kernel_avg_ms)
        res = {
            "metric": cfg["metric"],
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": cfg["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config, "batch_per_gpu": B, "samples_per_unit": n,
                       "out": list(cfg["out"]), "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": cfg["kernel"], "kernel_avg_ms": round(kernel_avg_ms, 4),
                         "kernel_min_ms": round(kernel_ms[0], 4), "kernel_median_ms": round(kernel_ms[len(kernel_ms) // 2], 4),
                         "algorithmic_bytes_per_unit": cfg["bytes_per_unit"], "units_per_launch": B},
        }
        if args.config in ("vgg4096", "vgg200k"):
            res["roofline"]["bound_note"] = ("priced against HBM as the metric asks; measured, the kernel takes the same shader cycles with and without its memory "
                                             "traffic and is held at about 1.98 GHz by the power limit (profiles/r03_fbank512_clock.txt)")
        if traffic is not None:
            res["roofline"]["traffic_source"] = TRAFFIC_SOURCE
        if traffic_note:
            res["roofline"]["traffic_note"] = traffic_note
        if gather is not None:
            res["allgather"] = gather
        if world == 1 and not args.no_secondary and args.config == "vgg4096":
            # the other benchmark configurations that fit one GPU, a few launches each (not part of `value`)
            del x, out
            torch.cuda.empty_cache()
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from bench_configs import run_config
                res["secondary"] = [run_config(name, dev, 10) for name in ("har", "logmel", "epic", "pcm", "resample", "patch", "patch16", "tokens")]
            except Exception as exc:  # never let a secondary configuration take the headline line down
                res["secondary"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
