"""CPU tests of the host side: C-ABI export/parse checks (no compute without a GPU), shape rules, table
construction, wrapper argument handling, batch partitioner (world_size 2, gloo)."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def N():
    from multi_modal_amd import _native
    _native.load_library()
    return _native


def test_library_exports_every_declared_symbol(N):
    header = open(os.path.join(ROOT, "include", "mmfe.h")).read()
    declared = set(re.findall(r"\b(mmfe_[a-z0-9_]+)\s*\(", header))
    declared -= {"mmfe_spec", "mmfe_plan"}
    assert declared == set(N.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(N.library_path())
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.mmfe_abi_version() == N.ABI_VERSION


def test_spec_struct_matches_header(N):
    header = open(os.path.join(ROOT, "include", "mmfe.h")).read()
    body = header[header.index("typedef struct mmfe_spec {"):header.index("} mmfe_spec;")]
    fields = re.findall(r"^\s*(int32_t|float)\s+([a-z0-9_]+);", body, re.M)
    assert [f[1] for f in fields] == [f[0] for f in N.MmfeSpec._fields_]
    for (ctype, name), (pname, ptype) in zip(fields, N.MmfeSpec._fields_):
        assert (ptype is ctypes.c_int32) == (ctype == "int32_t"), name
    assert ctypes.sizeof(N.MmfeSpec) == 4 * len(fields)


def test_no_cpu_fallback(N):
    """Without a device, plan creation fails loudly (there is no CPU path in the product)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400, n_mel=2, power=2, log_kind=N.LOG_LN_MAX)
    with pytest.raises(N.MmfeError, match="no HIP device|CPU"):
        N.Plan(spec, np.ones(400, np.float32), [1, 2], [1, 1], np.ones(2, np.float32))
    import multi_modal_amd as mm
    with pytest.raises(mm.MmfeError):
        mm.preprocessing_audio(torch.zeros(1, 160000))


def test_bad_spec_rejected(N):
    spec = N.make_spec(kind=99, n_fft=512, win_len=400, hop=400)
    fr, ft = ctypes.c_int64(), ctypes.c_int64()
    rc = N.load_library().mmfe_spec_out_shape(ctypes.byref(spec), 1000, ctypes.byref(fr), ctypes.byref(ft))
    assert rc == -1 and "kind" in N.last_error()
    spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=0)
    assert N.load_library().mmfe_spec_out_shape(ctypes.byref(spec), 1000, ctypes.byref(fr), ctypes.byref(ft)) == -1
    spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400)
    spec.struct_size = 8
    assert N.load_library().mmfe_spec_out_shape(ctypes.byref(spec), 1000, ctypes.byref(fr), ctypes.byref(ft)) == -1
    # input element type: float32 or 16-bit PCM, the latter for the log-mel / log-spectrogram kinds only
    spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400, in_dtype=7)
    assert N.load_library().mmfe_spec_out_shape(ctypes.byref(spec), 1000, ctypes.byref(fr), ctypes.byref(ft)) == -1
    assert "in_dtype" in N.last_error()
    spec = N.make_spec(kind=N.KIND_FRAME, win_len=16, hop=8, frames_mode=N.FRAMES_CEIL, in_dtype=N.IN_S16)
    assert N.load_library().mmfe_spec_out_shape(ctypes.byref(spec), 1000, ctypes.byref(fr), ctypes.byref(ft)) == -1
    assert "PCM" in N.last_error()
    spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400, n_mel=128, in_dtype=N.IN_S16)
    assert N.spec_out_shape(spec, 160000) == (400, 128)


def test_shape_rules_match_the_oracle(N):
    from oracle import frontend_oracle as O
    # kaldi snip-edges, then pad/cut to 384
    for n in [399, 400, 401, 799, 800, 12345, 160000, 200000]:
        spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400, n_mel=128)
        m = 0 if n < 400 else 1 + (n - 400) // 400
        assert N.spec_out_shape(spec, n) == (m, 128)
        spec = N.make_spec(kind=N.KIND_FBANK, n_fft=512, win_len=400, hop=400, n_mel=128, out_frames=384)
        assert N.spec_out_shape(spec, n) == (384, 128)
    # librosa centred framing with n_fft 511
    spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=511, win_len=240, win_off=135, hop=120, pad_mode=N.PAD_ZERO,
                       frames_mode=N.FRAMES_CENTER, layout=N.LAYOUT_FEAT_MAJOR)
    for n in [30672, 30696, 1, 119, 120, 121, 24000]:
        assert N.spec_out_shape(spec, n) == (O.log_specgram(np.zeros(n, np.float32)).shape[1], 256)
    # scipy framing
    spec = N.make_spec(kind=N.KIND_STFT_MAG, n_fft=32, win_len=32, hop=4, pad_mode=N.PAD_ZERO, frames_mode=N.FRAMES_SCIPY,
                       power=1, layout=N.LAYOUT_FEAT_MAJOR)
    for n in [32, 33, 35, 36, 192, 1003]:
        assert N.spec_out_shape(spec, n) == (O.stft_mag(np.zeros((1, n), np.float32)).shape[-1], 17)
    # frame_signal
    for n, fl, hop in [(10, 4, 3), (160000, 400, 160), (3, 8, 2), (100, 16, 7), (5, 5, 2), (7, 5, 3)]:
        spec = N.make_spec(kind=N.KIND_FRAME, win_len=fl, hop=hop, frames_mode=N.FRAMES_CEIL)
        assert N.spec_out_shape(spec, n) == tuple(O.frame_signal(torch.zeros(n), fl, hop).shape)
    # long-utterance log-mel: last frame dropped
    spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=400, win_len=400, hop=160, pad_mode=N.PAD_REFLECT, frames_mode=N.FRAMES_CENTER,
                       n_mel=80, drop_last=1, layout=N.LAYOUT_FEAT_MAJOR)
    assert N.spec_out_shape(spec, 480000) == (3000, 80)


def test_mel_tables_match_the_oracle():
    from multi_modal_amd import melbank
    from oracle import frontend_oracle as O
    for kw in [dict(num_bins=128, padded_window=512, sample_freq=16000.0), dict(num_bins=23, padded_window=512, sample_freq=16000.0),
               dict(num_bins=40, padded_window=256, sample_freq=8000.0, low_freq=0.0, high_freq=-200.0)]:
        assert torch.equal(melbank.kaldi_mel_banks(**kw), O.kaldi_mel_banks(**kw))
    for wt in ["hanning", "hamming", "povey", "rectangular", "blackman"]:
        assert np.array_equal(melbank.kaldi_window(wt, 400), O.kaldi_window(wt, 400).numpy())
    with pytest.raises(ValueError):
        melbank.kaldi_window("kaiser", 400)
    assert np.allclose(melbank.slaney_mel_filters(201, 80, 16000), O.slaney_mel_filters(201, 80, 16000).T, atol=1e-12)
    import scipy.signal
    assert np.array_equal(melbank.periodic_hann(240), scipy.signal.get_window("hann", 240, fftbins=True))
    assert np.array_equal(melbank.periodic_hann(32), scipy.signal.get_window("hann", 32))


def test_mel_bank_against_an_independent_float64_bank():
    """The product's kaldi mel bank against transformers.audio_utils.mel_filter_bank (float64, kaldi mel scale, triangles in mel
    space): code that shares nothing with melbank.py or the oracle.  2e-5 is the float32 rounding of the product's bank."""
    from multi_modal_amd import melbank
    audio_utils = pytest.importorskip("transformers.audio_utils")
    for n_mel, n_fft, sr, lo, hi in [(128, 512, 16000, 20.0, 8000.0), (23, 512, 16000, 20.0, 8000.0), (40, 256, 8000, 0.0, 3800.0)]:
        hf = audio_utils.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mel, min_frequency=lo, max_frequency=hi,
                                         sampling_rate=sr, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
        mine = torch.nn.functional.pad(melbank.kaldi_mel_banks(n_mel, n_fft, float(sr), lo, hi if hi < sr / 2 else 0.0), (0, 1)).numpy()
        assert mine.shape == hf.T.shape
        assert np.abs(hf.T - mine).max() <= 2e-5


def test_dense_to_bands_round_trip():
    from multi_modal_amd import melbank
    bank = torch.nn.functional.pad(melbank.kaldi_mel_banks(128, 512, 16000.0), (0, 1)).numpy()
    bs, bl, bw = melbank.dense_to_bands(bank)
    assert bl.sum() == bw.size == 504 and bl[3] == 0 and bl.max() == 10
    rebuilt = np.zeros_like(bank)
    off = 0
    for m in range(128):
        rebuilt[m, bs[m]:bs[m] + bl[m]] = bw[off:off + bl[m]]
        off += bl[m]
    assert np.array_equal(rebuilt, bank)
    # a filter with an interior zero keeps its span
    w = np.zeros((2, 10), np.float32)
    w[0, [2, 4]] = 1.0
    bs, bl, bw = melbank.dense_to_bands(w)
    assert (bs[0], bl[0], bl[1]) == (2, 3, 0) and np.array_equal(bw, [1, 0, 1])


def test_wrapper_argument_errors_need_no_gpu():
    import multi_modal_amd as mm
    x = torch.zeros(1, 16000)
    for kw in [dict(dither=1.0), dict(use_energy=True), dict(vtln_warp=1.1), dict(snip_edges=False), dict(subtract_mean=True),
               dict(use_power=False)]:
        with pytest.raises(NotImplementedError):
            mm.fbank(x, **kw)
    with pytest.raises(ValueError):
        mm.fbank(torch.zeros(16000))
    with pytest.raises(ValueError):
        mm.fbank(x, channel=3)
    assert mm.fbank(torch.zeros(1, 100)).shape == (0, 0)          # shorter than one window: empty, like the reference
    with pytest.raises(ValueError):
        mm.frame_signal(torch.zeros(10), 4, 2, torch.ones(5))
    with pytest.raises(ValueError):
        mm.stft_mag(np.zeros((2, 31), np.float32))
    from multi_modal_amd import epic
    from oracle import frontend_oracle as O
    for args in [(0, 0, 60.0, 240000), (10 ** 6, 0, 60.0, 240000), (120, 30, 60.0, 240000), (77, 3, 50.0, 100000)]:
        assert epic.sound_slice_bounds(*args) == O.sound_slice_bounds(*args)
    assert torch.equal(mm.spec_to_tensor(np.full((4, 5), 255.0, np.float32)), torch.ones(1, 4, 5))
    with pytest.raises(ValueError):
        mm.logmel.log_mel_spectrogram(torch.zeros(1, 16000), precision="float16")
    assert isinstance(mm.plans.last_kernel(), str)


def test_mel_stream_shapes_cover_the_common_banks():
    """The n_fft = 400 radix kernel has template instances for a fixed list of mel stream shapes
    (csrc/rfft400.hip: kShapes).  Recompute the shape of the common banks the way the library does and
    check that each fits an instance; a bank that fits none silently takes the slower any-length kernel."""
    from multi_modal_amd import melbank
    shapes = [((5, 19, 17, 0), True), ((8, 20, 20, 0), True), ((4, 8, 12, 24), False), ((12, 44, 0, 0), False), ((36, 32, 0, 0), True)]

    def stream_shape(fb):
        bs, bl, _ = melbank.dense_to_bands(np.asarray(fb, dtype=np.float32))
        n_mel = len(bs)
        nfb = (n_mel + 3) // 4
        units = 2 * nfb
        ns = (units + 15) // 16
        run = []
        for f in range(nfb):
            ms = [m for m in range(4 * f, min(4 * f + 4, n_mel)) if bl[m] > 0]
            run.append(max(bs[m] + bl[m] for m in ms) - min(bs[m] for m in ms) if ms else 0)
        split = units - 16 * (ns - 1) <= 8
        steps = [0, 0, 0, 0]
        for u in range(units):
            st, f = u // 16, u >> 1
            need = (run[f] + 1) // 2 if (split and st == ns - 1) else run[f]
            steps[st] = max(steps[st], need, 1)
        return tuple(int(v) for v in steps), split

    def fits(steps, split):
        return any(sp == split and all((a > 0) == (b > 0) and a >= b for a, b in zip(sh, steps)) for sh, sp in shapes)

    for n_mels in (40, 64, 80, 128):
        assert fits(*stream_shape(melbank.slaney_mel_filters(201, n_mels, 16000))), n_mels
    for n_mels in (40, 64, 80, 128):
        assert fits(*stream_shape(melbank.kaldi_mel_banks(n_mels, 400, 16000.0).numpy())), n_mels


def test_shard_ranges():
    from multi_modal_amd.partition import shard_range, shard_size
    for n, w in [(200000, 8), (4096, 8), (10, 4), (3, 8), (0, 2), (7, 1)]:
        ranges = [shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
            assert b == c and a <= b
        assert max(b - a for a, b in ranges) <= shard_size(n, w)
    assert shard_range(200000, 3, 8) == (75000, 100000)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from multi_modal_amd.partition import all_gather_features, pick_gather_mode, shard_range, sharded_apply
world = int(sys.argv[4])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=world)
rank = dist.get_rank()
for n in (7, 8, 1, 12, 2):
    full = torch.arange(n * 6, dtype=torch.float32).reshape(n, 2, 3)
    lo, hi = shard_range(n, rank, world)
    for mode in ("collective", "direct", "auto"):
        got = all_gather_features(full[lo:hi].clone(), n, mode=mode)
        assert got.shape == full.shape and torch.equal(got, full), (n, rank, mode)       # bit-equal to the single-process tensor
    # caller-provided output of the direct form: shards land in their final slots, no other element is touched
    out = torch.full((n + 2, 2, 3), -1.0)
    got = all_gather_features(full[lo:hi].clone(), n, out=out, mode="direct")
    assert got.data_ptr() == out.data_ptr() and torch.equal(out[:n], full) and bool((out[n:] == -1).all())
    # a stand-in front-end (the real one needs a GPU): the partitioner only sees a (B, n) -> (B, f, m) map
    fn = lambda x: torch.stack([x.sum(1, keepdim=True), x.max(1, keepdim=True).values], 1)
    batch = torch.arange(n * 5, dtype=torch.float32).reshape(n, 5)
    for mode in ("collective", "direct"):
        assert torch.equal(sharded_apply(fn, batch, mode=mode), fn(batch))
    assert sharded_apply(fn, batch, gather=False).shape[0] == hi - lo
    for chunks in (2, 3, 5):           # transformed and exchanged piece by piece: the same tensor
        assert torch.equal(sharded_apply(fn, batch, chunks=chunks), fn(batch)), (n, rank, chunks)
assert pick_gather_mode("auto", 8, True, True) == "direct" and pick_gather_mode("auto", 2, True, True) == "collective"
assert pick_gather_mode("auto", 2, False, False) == "direct" and pick_gather_mode("auto", 3, True, False) == "collective"
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("world", [2, 3])
def test_partition_all_gather_gloo(tmp_path, world):
    """Both forms of the feature all-gather (the collective, and shards sent point-to-point into their final slots) over gloo with
    2 and 3 ranks, even and uneven splits, empty tail shards: bit-equal to the single-process tensor."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_bench_cpu_baseline_fields():
    """The CPU leg of bench.py (the oracle timed per clip) on a tiny budget: the fields the JSON line carries."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.BYTES_PER_CLIP == 811008 and bench.BYTES_PER_UTT == 2880000
    r = bench.cpu_baseline(budget_s=0.6, max_clips=3, legs=("1_thread", "4_workers", "c_openmp_all_cores"))
    assert r["unit"] == "clips/s" and r["kind"] == "port" and "clips" in r["sample"]
    assert r["legs"]["1_thread"]["cores"] == 1 and r["legs"]["1_thread"]["value"] > 0
    assert r["legs"]["4_workers"]["cores"] == 4 and r["legs"]["4_workers"]["value"] > 0
    assert r["cores"] == 4 and r["value"] == r["legs"]["4_workers"]["value"]       # the leg the reference's DataLoader corresponds to
    c = r["legs"]["c_openmp_all_cores"]                                            # the compiled C / OpenMP restatement, every core of the share
    assert c["cores"] == bench.host_core_share() and c["value"] > 0 and c["clips"] >= 1
    assert 1 <= bench.host_core_share() <= 16
    assert set(bench.CONFIGS) == {"vgg4096", "vgg200k", "logmel"}


def test_resampler_taps_and_output_lengths(N):
    """The low-pass the resampler uploads is the one scipy.signal.resample_poly designs for float32 rows, and the output
    length rule is scipy's ceil(n * up / down)."""
    import scipy.signal
    from multi_modal_amd import resample
    for up, down in [(160, 441), (1, 3), (1, 2), (441, 160), (2, 1), (3, 7)]:
        m = max(up, down)
        h = scipy.signal.firwin(20 * m + 1, 1.0 / m, window=("kaiser", 5.0)).astype(np.float32)
        h *= up
        assert np.array_equal(resample.lowpass_taps(up, down), h)
        spec = N.make_spec(kind=N.KIND_RESAMPLE, n_fft=up, hop=down, win_len=20 * m + 1)
        for n in [1, 7, 440, 441, 442, 44100, 441000]:
            assert N.spec_out_shape(spec, n) == (-(-n * up // down), 1)
        spec = N.make_spec(kind=N.KIND_RESAMPLE, n_fft=up, hop=down, win_len=20 * m + 1, out_frames=160000)
        assert N.spec_out_shape(spec, 441000) == (160000, 1)
    assert resample._ratio(44100, 16000) == (160, 441) and resample._ratio(48000, 24000) == (1, 2)
    with pytest.raises(ValueError):
        resample._ratio(0, 16000)
