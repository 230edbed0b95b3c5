"""The multi-GPU path on the one GPU a test box has: a FRESH child process (started before anything in it touches
the GPU) forms a 1-rank RCCL group (torch backend "nccl") and runs what every rank of the 8-GPU configuration runs:

  * the per-GPU shard of the 200 000-clip configuration (25 000 clips generated on the device) through
    partition.sharded_apply(preprocessing_audio_batch, gather=True): the RCCL all_gather_into_tensor executes;
  * partition.all_gather_features on a shard with a padded tail shape;
  * stats.RunningNorm.finalize(): the RCCL all-reduce of (sum, sum of squares, count).

Checks are the size-independent properties of the path: finiteness, position independence (a clip's features do not
depend on where in the batch it sits: bit-equal to a small batch of the same clips), gathered == local for one rank,
and the statistics against a float64 reduction by torch.  The reference has no counterpart (it is single-GPU:
classification/vggsound/vanilla_multi.py:113); uneven shards over several ranks are covered on CPU with gloo
(tests/test_host_cpu.py).
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
n_clips = int(sys.argv[2])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
import multi_modal_amd as mm
from multi_modal_amd import partition, stats

g = torch.Generator(device=dev).manual_seed(1234)
x = torch.empty(n_clips, 160000, device=dev)
for lo in range(0, n_clips, 5000):          # generated on the device in slabs (randn of 4e9 elements at once needs 2x the memory)
    hi = min(n_clips, lo + 5000)
    x[lo:hi] = (0.1 * torch.randn(hi - lo, 160000, device=dev, generator=g)).clamp_(-1, 1)

full = partition.sharded_apply(mm.preprocessing_audio_batch, x, gather=True, mode="collective")
torch.cuda.synchronize()
assert tuple(full.shape) == (n_clips, 384, 128), full.shape
assert bool(torch.isfinite(full).all())
local = partition.sharded_apply(mm.preprocessing_audio_batch, x, gather=False)
assert torch.equal(full, local)                              # one rank: the gathered tensor is the local shard
direct = partition.sharded_apply(mm.preprocessing_audio_batch, x, gather=True, mode="direct")
assert direct.data_ptr() != local.data_ptr() and torch.equal(direct, local)      # the point-to-point form: own shard copied into its slot
del direct
pick = torch.tensor([0, 1, 4095, 4096, n_clips // 2, n_clips - 2, n_clips - 1], device=dev)
small = mm.preprocessing_audio_batch(x[pick].contiguous())
assert torch.equal(full[pick], small), "features depend on the position in the batch"
print("shard ok", n_clips, float(full.mean()), flush=True)

# all_gather_features: explicit call, output buffer reuse, and the single-rank shape check
shard = full[:1001].contiguous()
out = partition.all_gather_features(shard, 1001)
assert torch.equal(out, shard)
buf = torch.empty(1001, 384, 128, device=dev)
for mode in ("collective", "direct", "auto"):
    buf.fill_(-1.0)
    out2 = partition.all_gather_features(shard, 1001, out=buf, mode=mode)
    assert out2.data_ptr() == buf.data_ptr() and torch.equal(out2, shard), mode
try:
    partition.all_gather_features(shard, 1002)
    raise SystemExit("a shard of the wrong size was accepted")
except ValueError:
    pass

# cal_norm over the shard: two float64 sums on the device, all-reduced over RCCL
r = stats.RunningNorm(dev)
for lo in range(0, n_clips, 4096):
    r.update(full[lo:lo + 4096])
mean, std = r.finalize()
ref = full[:4096 * 2].double()
r2 = stats.RunningNorm(dev)
r2.update(full[:4096 * 2])
m2, s2 = r2.finalize()
assert abs(m2 - float(ref.mean())) < 1e-9 and abs(s2 - float(ref.std(unbiased=False))) < 1e-9, (m2, s2)
assert abs(mean - m2) < 0.05 and abs(std - s2) < 0.05          # same distribution, more clips
print("norm ok", mean, std, flush=True)
dist.barrier()
dist.destroy_process_group()
print("child ok", flush=True)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_child(tmp_path, n_clips, timeout):
    script = tmp_path / "multi_child.py"
    script.write_text(_CHILD)
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-u", str(script), ROOT, str(n_clips)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=timeout)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-4000:]
    assert "shard ok" in out and "norm ok" in out and "child ok" in out, out[-4000:]
    return out


def test_rccl_one_rank_full_shard(tmp_path):
    """25 000 clips = the per-GPU shard of the 200 000-clip / 8-GPU configuration (16 GB in, 4.9 GB out, 4.9 GB gathered)."""
    out = _run_child(tmp_path, 25000, 300)
    print(out[-600:])


def test_bench_configs_under_one_rank_group(tmp_path):
    """bench.py --config vgg200k / logmel through a 1-rank RCCL group (the launch shape the driver uses for N > 1):
    one JSON line with the gather timed separately."""
    import json
    for config, extra in (("vgg200k", ["--batch", "2048"]), ("logmel", ["--batch", "64"])):
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                   MMFE_BENCH_DIST="1")
        p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", config, "--steps", "3",
                            "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--precondition", "0.05"] + extra,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
        res = json.loads(line)
        assert res["n_gpus"] == 1 and res["value"] > 0 and res["roofline"]["frac"] > 0
        assert "allgather" in res and "error" not in res["allgather"], res.get("allgather")
        assert res["allgather"]["clips_per_s_with_allgather"] > 0
        for mode in ("collective", "direct"):
            assert res["allgather"]["modes"][mode]["allgather_ms"] > 0, res["allgather"]
        assert res["allgather"]["auto_picks"] in ("collective", "direct")
        assert res["allgather"]["modes"]["direct_4_pieces_overlapped"].get("ms_per_step_with_allgather", 0) > 0, res["allgather"]
