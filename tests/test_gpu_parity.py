"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (floating point; shapes and indices are exact):
  kaldi log-mel (raw):      p99.9 |d| <= 1e-4, max |d| <= 2e-3  (near-empty 1-2 bin mel filters amplify fp32
                            FFT rounding; two correct fp32 FFTs cannot agree to 1e-4 max-abs)
  kaldi log-mel normalised: max |d| <= 5e-4
  EPIC log spectrogram raw: max |d| <= 1e-4
  HAR magnitudes:           max |d| <= 1e-6 (values O(1))
  frame_signal:             bit exact
"""
import numpy as np
import pytest
import torch

from conftest import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mm():
    import multi_modal_amd as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


@pytest.fixture(scope="module")
def O():
    from oracle import frontend_oracle
    return frontend_oracle


KINDS = ["noise0.1", "uniform", "sine+noise", "silence", "impulse", "dc", "quiet"]
HIGH_DYNAMIC_RANGE = ("sine+noise", "dc")


def _stats(d):
    d = np.abs(np.asarray(d, dtype=np.float64)).ravel()
    return d.max(), np.percentile(d, 99.9)


@pytest.mark.parametrize("kind", KINDS)
def test_vggsound_raw_fbank(mm, O, kind):
    x = synth(kind, 160000, 1234)
    ref = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=16000, window_type="hanning", num_mel_bins=128,
                        frame_shift=25).numpy()
    got = mm.fbank(torch.from_numpy(x)[None].cuda(), htk_compat=True, sample_frequency=16000, use_energy=False,
                   window_type="hanning", num_mel_bins=128, dither=0.0, frame_shift=25).cpu().numpy()
    assert got.shape == ref.shape == (400, 128)
    mx, p999 = _stats(got - ref)
    print(kind, "raw fbank max", mx, "p99.9", p999)
    if kind in HIGH_DYNAMIC_RANGE:
        # A strong tone / offset next to a 1e-3 noise floor puts most bins at the rounding-noise level of
        # ANY fp32 FFT of the frame; two fp32 evaluations then differ by more than 1e-4 from each other.
        # Judge both against the float64 evaluation of the same formula: the HIP path must be as close
        # to the exact value as the fp32 CPU restatement is (factor 1.5), and still within 2e-3 / 5e-4 of it.
        exact = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=16000, window_type="hanning",
                              num_mel_bins=128, frame_shift=25, compute_dtype=torch.float64).numpy()
        g_mx, g_p = _stats(got - exact)
        r_mx, r_p = _stats(ref - exact)
        print(kind, "vs float64: hip max", g_mx, "p99.9", g_p, "| cpu fp32 max", r_mx, "p99.9", r_p)
        assert g_p <= max(1e-4, 1.5 * r_p) and g_mx <= max(2e-3, 1.5 * r_mx)
        assert p999 <= 5e-4 and mx <= 4e-3
    else:
        assert p999 <= 1e-4 and mx <= 2e-3
    # mel filter 3 is empty for this bank: exact floor ln(2^-23) in every frame
    assert np.all(got[:, 3] == np.float32(np.log(np.float32(1.1920929e-07))))


def test_vggsound_preprocessing_batch(mm, O):
    clips = np.stack([synth(k, 160000, 1234 + i) for i, k in enumerate(KINDS)])
    ref = np.stack([O.preprocessing_audio(torch.from_numpy(c)[None]).numpy() for c in clips])
    got = mm.preprocessing_audio_batch(torch.from_numpy(clips).cuda()).cpu().numpy()
    assert got.shape == ref.shape == (len(KINDS), 384, 128)
    mx, p999 = _stats(got - ref)
    print("normalised max", mx, "p99.9", p999)
    assert mx <= 5e-4
    single = mm.preprocessing_audio(torch.from_numpy(clips[0])[None])       # CPU tensor in -> CPU tensor out
    assert single.device.type == "cpu" and single.shape == (384, 128)
    assert np.array_equal(single.numpy(), got[0])


@pytest.mark.parametrize("n", [399, 400, 401, 799, 800, 12345, 153600, 200000])
def test_vggsound_lengths(mm, O, n):
    """Ragged clip lengths: fewer than 384 frames -> padded rows, more -> cut, n < window -> all padding."""
    x = synth("noise0.1", n, 7 + n)
    got = mm.preprocessing_audio(torch.from_numpy(x)[None].cuda()).cpu().numpy()
    assert got.shape == (384, 128)
    if n < 400:
        assert np.allclose(got, 5.1991 / 4.4049, atol=1e-6)
        return
    ref = O.preprocessing_audio(torch.from_numpy(x)[None]).numpy()
    mx, _ = _stats(got - ref)
    assert mx <= 5e-4
    m = 1 + (n - 400) // 400
    if m < 384:
        assert np.all(got[m:] == got[m, 0])


def test_fbank_other_parameters(mm, O):
    """Other fbank parameter sets: AST-style (povey window, 10 ms shift), 23-bin default, odd offsets."""
    x = synth("noise0.1", 48000, 99)
    for kw in [dict(num_mel_bins=128, frame_shift=10.0, window_type="povey"),
               dict(num_mel_bins=23),
               dict(num_mel_bins=80, window_type="hamming", frame_length=20.0, frame_shift=7.5, low_freq=0.0, high_freq=-400.0),
               dict(num_mel_bins=40, remove_dc_offset=False, preemphasis_coefficient=0.0, window_type="rectangular")]:
        okw = dict(kw)
        ref = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=16000, **okw).numpy()
        got = mm.fbank(torch.from_numpy(x)[None].cuda(), sample_frequency=16000, **kw).cpu().numpy()
        assert got.shape == ref.shape
        mx, p999 = _stats(got - ref)
        print(kw, mx, p999)
        assert p999 <= 1e-4 and mx <= 2e-3


def test_fbank_generic_path_non_512(mm, O):
    """A window that does not round to 512 (8 kHz, 25 ms -> 200 samples -> n_fft 256) goes through the
    any-length DFT kernel."""
    x = synth("noise0.1", 16000, 5)
    ref = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=8000, num_mel_bins=40).numpy()
    got = mm.fbank(torch.from_numpy(x)[None].cuda(), sample_frequency=8000, num_mel_bins=40).cpu().numpy()
    assert got.shape == ref.shape
    mx, p999 = _stats(got - ref)
    assert p999 <= 1e-4 and mx <= 2e-3


def test_fbank_unaligned_and_strided_inputs(mm, O):
    """Rows that start at odd sample offsets (4-byte aligned only), odd row strides and row views of a wider
    buffer take the scalar-load kernel variants; results must not depend on alignment."""
    base = torch.from_numpy(np.stack([synth("noise0.1", 160001, 50 + i) for i in range(3)])).cuda()
    ref = torch.stack([O.preprocessing_audio(base[i:i + 1, 1:].cpu()) for i in range(3)]).numpy()
    got_view = mm.preprocessing_audio_batch(base[:, 1:]).cpu().numpy()            # odd offset, odd stride (160001)
    got_copy = mm.preprocessing_audio_batch(base[:, 1:].contiguous()).cpu().numpy()
    assert np.abs(got_view - ref).max() <= 5e-4 and np.abs(got_copy - ref).max() <= 5e-4
    # two kernel variants (scalar vs 8-byte loads): same algorithm, separately compiled, so fp32 rounding of the
    # FFT may differ in the last bit and near-empty mel cells amplify that like any other fp32 difference
    assert np.abs(got_view - got_copy).max() <= 5e-4
    wide = torch.zeros(3, 200000, device="cuda")
    wide[:, :160000] = base[:, :160000]
    assert torch.equal(mm.preprocessing_audio_batch(wide[:, :160000]), mm.preprocessing_audio_batch(base[:, :160000].contiguous()))
    # non-unit inner stride falls back to a contiguous copy
    inter = torch.zeros(2, 320000, device="cuda")
    inter[:, ::2] = base[:2, :160000]
    assert torch.equal(mm.preprocessing_audio_batch(inter[:, ::2]), mm.preprocessing_audio_batch(base[:2, :160000].contiguous()))


def test_fbank_odd_window_and_hop(mm, O):
    """22.05 kHz, 25 ms -> 551-tap window (odd), n_fft 1024, hop 220: served by the any-length DFT kernel;
    11.025 kHz -> 275 taps, n_fft 512, hop 110: the 512 kernel with an odd window (scalar loads)."""
    x = synth("noise0.1", 30000, 77)
    for sf, nm in [(22050.0, 40), (11025.0, 64)]:
        ref = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=sf, num_mel_bins=nm).numpy()
        got = mm.fbank(torch.from_numpy(x)[None].cuda(), sample_frequency=sf, num_mel_bins=nm).cpu().numpy()
        assert got.shape == ref.shape
        mx, p999 = _stats(got - ref)
        assert p999 <= 1e-4 and mx <= 2e-3


def test_fbank_valu_mel_form_matches(mm, O):
    """The general (VALU) form of the banded mel projection and the matrix-pipe form give the same features."""
    import os
    x = torch.from_numpy(np.stack([synth("noise0.1", 160000, 90), synth("uniform", 160000, 91)])).cuda()
    a = mm.preprocessing_audio_batch(x).clone()
    from multi_modal_amd import plans
    os.environ["MMFE_MEL_FORM"] = "valu"
    try:
        plans.clear_plans()
        b = mm.preprocessing_audio_batch(x).clone()
    finally:
        del os.environ["MMFE_MEL_FORM"]
        plans.clear_plans()
    assert (a - b).abs().max().item() <= 2e-6


def test_har_unaligned_rows(mm, O):
    rng = np.random.default_rng(3)
    buf = rng.standard_normal((5, 6, 195)).astype(np.float32)
    t = torch.from_numpy(buf).cuda()
    got = mm.stft_mag(t[:, :, 1:194]).cpu().numpy()          # T = 193 (odd), rows start at odd offsets
    ref = O.stft_mag(buf[:, :, 1:194])
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-6


def test_fbank_unsupported_raise(mm):
    x = torch.zeros(1, 16000)
    for kw in [dict(dither=1.0), dict(use_energy=True), dict(vtln_warp=1.1), dict(snip_edges=False), dict(subtract_mean=True)]:
        with pytest.raises(NotImplementedError):
            mm.fbank(x, **kw)


@pytest.mark.parametrize("n", [30672, 30696])
@pytest.mark.parametrize("kind", ["noise0.1", "quiet", "silence", "sine+noise"])
def test_epic_log_specgram(mm, O, n, kind):
    x = synth(kind, n, 4242 + n)
    ref = O.log_specgram(x)
    got = mm.log_specgram(x)
    assert isinstance(got, np.ndarray) and got.dtype == np.float32
    assert got.shape == ref.shape == (256, 256)
    mx, p999 = _stats(got - ref)
    print(kind, n, "epic raw max", mx, "p99.9", p999)
    assert mx <= 1e-4
    t = mm.spec_to_tensor(got)
    assert t.shape == (1, 256, 256)
    assert np.allclose(t.numpy(), O.spec_to_tensor(ref).numpy(), atol=1e-6)


def test_epic_batch_div255_and_slices(mm, O):
    rng = np.random.default_rng(5)
    recs = [synth("noise0.1", 24000 * 5, 1), synth("uniform", 24000 * 3 + 17, 2)]
    rec_index = [0, 0, 1, 1, 0]
    start_frame = [100, 130, 30, 60, 170]
    idx = [3, 0, 7, 2, 1]
    fps = [60.0, 60.0, 50.0, 50.0, 60.0]
    bounds = [O.sound_slice_bounds(start_frame[i], idx[i], fps[i], len(recs[rec_index[i]])) for i in range(5)]
    assert len({r - l for l, r in bounds}) == 1
    ref = np.stack([O.spec_to_tensor(O.log_specgram(recs[rec_index[i]][l:r])).numpy() for i, (l, r) in enumerate(bounds)])
    got = mm.epic.extract_sound_features([torch.from_numpy(r) for r in recs], rec_index, start_frame, idx, fps).cpu().numpy()
    assert got.shape == ref.shape == (5, 1, 256, 256)
    assert np.abs(got - ref).max() <= 1e-4 / 255 + 1e-7
    # edge slices (clamped to the first / last 30696 samples)
    b0 = O.sound_slice_bounds(0, 0, 60.0, len(recs[0]))
    b1 = O.sound_slice_bounds(10 ** 6, 0, 60.0, len(recs[0]))
    assert b0 == (0, 30696) and b1 == (len(recs[0]) - 30696, len(recs[0]))
    got = mm.epic.extract_sound_features([torch.from_numpy(recs[0])], [0, 0], [0, 10 ** 6], [0, 0], [60.0, 60.0]).cpu().numpy()
    for i, (l, r) in enumerate([b0, b1]):
        ref = O.spec_to_tensor(O.log_specgram(recs[0][l:r])).numpy()
        assert np.abs(got[i] - ref).max() <= 1e-4 / 255 + 1e-7


def test_epic_slices_against_the_reference_function(mm, O):
    """extract_sound_features against the slice ranges the reference's own _extract_sound_feature cut
    (tests/golden/epic_slices.npz): the features of every case must equal, bit for bit, those of the fixture's
    [left, right) range cut on the host and passed as a plain row - the same kernel on the same samples exactly when
    the device-side slice indices are the fixture's.  Then the whole chain against the oracle on those ranges."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "epic_slices.npz"))
    cases, bounds = z["cases"], z["bounds"]
    totals = sorted(set(int(v) for v in cases[:, 3]))
    recs = {n: synth("noise0.1", n, 900 + i) for i, n in enumerate(totals)}
    rec_list = [torch.from_numpy(recs[n]) for n in totals]
    keep = [i for i in range(len(cases)) if bounds[i, 1] - bounds[i, 0] >= 30672]          # whole slices: 256 frames each
    assert len(keep) >= 50
    rec_index = [totals.index(int(cases[i, 3])) for i in keep]
    got = mm.epic.extract_sound_features(rec_list, rec_index, [int(cases[i, 0]) for i in keep], [int(cases[i, 1]) for i in keep],
                                         [float(cases[i, 2]) for i in keep])
    assert got.shape == (len(keep), 1, 256, 256)
    lens = [int(bounds[i, 1] - bounds[i, 0]) for i in keep]
    rows = np.zeros((len(keep), max(lens)), np.float32)
    for j, i in enumerate(keep):
        rows[j, : lens[j]] = recs[int(cases[i, 3])][bounds[i, 0]:bounds[i, 1]]
    want = mm.log_specgram_batch(torch.from_numpy(rows).cuda(), div255=True, lengths=lens)
    assert torch.equal(got[:, 0], want)
    for j in (0, len(keep) // 2, len(keep) - 1):
        i = keep[j]
        ref = O.spec_to_tensor(O.log_specgram(recs[int(cases[i, 3])][bounds[i, 0]:bounds[i, 1]])).numpy()
        assert np.abs(got[j].cpu().numpy() - ref).max() <= 1e-4 / 255 + 1e-7
    # recordings shorter than one slice: the reference takes all of them
    short = [i for i in range(len(cases)) if 120 < bounds[i, 1] - bounds[i, 0] < 30672]
    assert short
    i = short[0]
    n = int(cases[i, 3])
    g = mm.epic.extract_sound_features([torch.from_numpy(recs[n])], [0], [int(cases[i, 0])], [int(cases[i, 1])], [float(cases[i, 2])])
    ref = O.spec_to_tensor(O.log_specgram(recs[n][bounds[i, 0]:bounds[i, 1]])).numpy()
    assert g.shape[1:] == ref.shape and np.abs(g[0].cpu().numpy() - ref).max() <= 1e-4 / 255 + 1e-7


def test_epic_spec_transform_against_the_reference_classes(mm, O):
    """The /255 of the kernel epilogue against the reference's own transform chain (tests/golden/spec_transform.npz is its output
    on a (256, 256) mode-'F' image): spec_to_tensor on a device tensor is a true division, bit exact; the fused epilogue folds
    ln 2 / 255 into the one multiplication behind its base-2 logarithm, so it is one rounding (of the natural logarithm) short of
    the two-step value and one (of the reciprocal) away from the division: |d| <= 2 ulp of the result."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spec_transform.npz"))
    seed, h, w = (int(v) for v in z["c0_seed_shape"])
    spec = (np.random.default_rng(seed).integers(-(1 << 22), 1 << 22, size=(h, w)).astype(np.float64) / float(1 << 18)).astype(np.float32)
    t = mm.spec_to_tensor(torch.from_numpy(spec).cuda())
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), z["c0_out"])
    x = synth("noise0.1", 30672, 77)
    raw = mm.log_specgram_batch(torch.from_numpy(x)[None].cuda())
    fused = mm.log_specgram_batch(torch.from_numpy(x)[None].cuda(), div255=True)
    chain = mm.spec_to_tensor(raw[0])                  # the reference's order: log spectrogram, then the transform chain
    d = (fused - chain).abs()
    assert (d <= 2 * torch.finfo(torch.float32).eps * chain.abs() + 1e-12).all()


def test_epic_mixed_slice_lengths_one_launch(mm, O):
    """A batch that mixes interior (30 672 +- 1 samples) and edge (30 696) slices, as a DataLoader batch of the reference
    does (epic_kitchen.py:143-154): ONE call, each row equal to its own per-slice oracle within 1e-4 / 255."""
    recs = [synth("noise0.1", 24000 * 5, 11), synth("uniform", 24000 * 4 + 333, 12), synth("quiet", 24000 * 3, 13)]
    rec_index = [0, 0, 1, 1, 2, 2, 0, 1]
    start_frame = [0, 100, 30, 10 ** 6, 60, 0, 170, 95]
    idx = [0, 3, 7, 0, 2, 0, 1, 4]
    fps = [60.0, 60.0, 50.0, 50.0, 60.0, 60.0, 60.0, 50.0]
    bounds = [O.sound_slice_bounds(start_frame[i], idx[i], fps[i], len(recs[rec_index[i]])) for i in range(len(idx))]
    sizes = {r - l for l, r in bounds}
    assert 30696 in sizes and len(sizes) >= 2
    ref = np.stack([O.spec_to_tensor(O.log_specgram(recs[rec_index[i]][l:r])).numpy() for i, (l, r) in enumerate(bounds)])
    got = mm.epic.extract_sound_features([torch.from_numpy(r) for r in recs], rec_index, start_frame, idx, fps).cpu().numpy()
    assert got.shape == ref.shape == (len(idx), 1, 256, 256)
    assert np.abs(got - ref).max() <= 1e-4 / 255 + 1e-7
    # the same through the strided ragged entry: rows padded to the longest slice, garbage in the padding
    n = max(sizes)
    rows = np.full((len(idx), n), 7.0, np.float32)
    for i, (l, r) in enumerate(bounds):
        rows[i, : r - l] = recs[rec_index[i]][l:r]
    got2 = mm.log_specgram_batch(torch.from_numpy(rows).cuda(), div255=True, lengths=[r - l for l, r in bounds]).cpu().numpy()
    assert np.array_equal(got2[:, None], got)
    # 16-bit PCM recordings, mixed lengths
    pcm = [torch.from_numpy(np.clip(np.round(r.astype(np.float64) * 32768), -32768, 32767).astype(np.int16)) for r in recs]
    dec = [p.to(torch.float32) / 32768 for p in pcm]
    a = mm.epic.extract_sound_features(pcm, rec_index, start_frame, idx, fps)
    b = mm.epic.extract_sound_features(dec, rec_index, start_frame, idx, fps)
    assert torch.equal(a, b)


def test_vggsound_ragged_batch_one_launch(mm, O):
    """Clips of different lengths in one batch tensor (padded with garbage to the longest): every clip's features equal
    those of the clip passed alone, bit for bit, through both load variants and the C ABI's mmfe_run_ragged directly."""
    lens = [160000, 399, 400, 401, 12345, 153600, 0, 80000, 159999]
    n = max(lens)
    rows = np.full((len(lens), n), 0.9, np.float32)
    clips = [synth("noise0.1", max(l, 1), 50 + i)[:l] for i, l in enumerate(lens)]
    for i, c in enumerate(clips):
        rows[i, : len(c)] = c
    x = torch.from_numpy(rows).cuda()
    got = mm.preprocessing_audio_batch(x, lengths=lens)
    assert mm.plans.last_kernel() == "fbank512" and tuple(got.shape) == (len(lens), 384, 128)
    for i, c in enumerate(clips):
        if len(c) < 400:
            assert torch.all(got[i] == got[1, 0, 0]) and abs(float(got[i, 0, 0]) - 5.1991 / 4.4049) < 1e-6
            continue
        # the clip alone, in a buffer of even row stride so that it takes the same load variant of the kernel as the
        # batch (the variants round the pre-emphasis differently in the last bit: equal to tolerance, not to the bit)
        buf = torch.zeros(1, len(c) + (len(c) & 1), device="cuda")
        buf[0, : len(c)] = torch.from_numpy(c).cuda()
        alone = mm.preprocessing_audio_batch(buf[:, : len(c)])[0]
        assert torch.equal(got[i], alone), i
        ref = O.preprocessing_audio(torch.from_numpy(c)[None]).numpy()
        assert np.abs(got[i].cpu().numpy() - ref).max() <= 5e-4
    # odd row stride -> the scalar-load variant; lengths given as a device tensor
    xo = torch.empty(len(lens), n + 1, device="cuda")[:, :n]
    xo.copy_(x)
    got_o = mm.preprocessing_audio_batch(xo, lengths=torch.tensor(lens, device="cuda"))
    assert float((got_o - got).abs().max()) <= 5e-4
    i = lens.index(159999)                      # an odd-length clip alone has an odd row stride: the same (scalar-load) variant
    assert torch.equal(got_o[i], mm.preprocessing_audio_batch(torch.from_numpy(clips[i])[None].cuda())[0])
    # bfloat16 output, ragged
    got_h = mm.preprocessing_audio_batch(x, lengths=lens, dtype=torch.bfloat16)
    assert torch.equal(got_h, got.to(torch.bfloat16))
    # lengths beyond the tensor are refused on the host when they are host data
    with pytest.raises(ValueError):
        mm.preprocessing_audio_batch(x, lengths=[n + 1] * len(lens))
    # other kinds say so instead of ignoring the lengths
    from multi_modal_amd import _native as N
    with pytest.raises(N.MmfeError, match="per-row lengths"):
        mm.plans.run_rows(mm.har._stft_plan(0, 32, 28), torch.zeros(4, 64, device="cuda"), lengths=[64, 64, 32, 64])


def test_logmel_ragged_rows(mm, O):
    """The n_fft 400 kernel (reflect padding at each row's own end) and the any-length kernel on ragged rows; the
    long-utterance configuration with its per-row clamp refuses them."""
    from multi_modal_amd import _native as N
    from multi_modal_amd import melbank
    lens = [48000, 16000, 401, 47999, 32160]
    n = max(lens)
    rows = np.full((len(lens), n), -0.7, np.float32)
    clips = [synth("noise0.1", l, 70 + i) for i, l in enumerate(lens)]
    for i, c in enumerate(clips):
        rows[i, : len(c)] = c
    fb = melbank.slaney_mel_filters(201, 80, 16000)
    bs, bl, bw = melbank.dense_to_bands(fb.astype(np.float32))
    for f64 in (False, True):
        spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=400, win_len=400, hop=160, pad_mode=N.PAD_REFLECT, frames_mode=N.FRAMES_CENTER,
                           n_mel=80, power=2, log_kind=N.LOG_LOG10_MAX, log_floor=1e-10, drop_last=1, layout=N.LAYOUT_FEAT_MAJOR,
                           pad_value=-3.0, window_f64=1 if f64 else 0)
        plan = N.Plan(spec, melbank.periodic_hann(400), bs, bl, bw, device=0)
        assert plan.kernel == ("dftgen" if f64 else "rfft400")
        x = torch.from_numpy(rows).cuda()
        got = mm.plans.run_rows(plan, x, lengths=lens)
        frames_max = n // 160
        assert tuple(got.shape) == (len(lens), 80, frames_max)
        for i, c in enumerate(clips):
            alone = mm.plans.run_rows(plan, torch.from_numpy(c)[None].cuda())[0]
            fr = alone.shape[1]
            assert fr == len(c) // 160
            assert torch.equal(got[i, :, :fr], alone), (f64, i)
            assert torch.all(got[i, :, fr:] == -3.0)
    with pytest.raises(N.MmfeError, match="clamp_below_max"):
        mm.plans.run_rows(mm.logmel._plan(0, 400, 160, 80, 16000, False), torch.zeros(2, 16000, device="cuda"), lengths=[16000, 8000])


def test_har_short_input_rejected(mm):
    for shape in [(1, 31), (2, 4)]:
        with pytest.raises(ValueError):
            mm.stft_mag(np.zeros(shape, np.float32))


def test_har_golden_scipy(mm, golden_dir):
    """Against stored outputs of scipy.signal.stft itself (the call at adbox.py:20)."""
    g = np.load(f"{golden_dir}/har_stft.npz")
    for name in ("b6x192", "c15x1003", "exact32", "t33"):
        got = mm.stft_mag(g[name + "_x"])
        assert got.shape == g[name + "_y"].shape
        assert np.abs(got - g[name + "_y"]).max() <= 1e-6 * max(1.0, float(np.abs(g[name + "_y"]).max()))


def test_frame_signal_golden_reference(mm, golden_dir):
    """Against stored outputs of the reference's own frame_signal: bit exact."""
    g = np.load(f"{golden_dir}/frame_signal.npz")
    i = 0
    while f"c{i}_signal" in g:
        fl, hop = (int(v) for v in g[f"c{i}_params"])
        s, w = torch.from_numpy(g[f"c{i}_signal"]), torch.from_numpy(g[f"c{i}_window"])
        assert np.array_equal(mm.frame_signal(s.cuda(), fl, hop, w.cuda()).cpu().numpy(), g[f"c{i}_out"])
        assert np.array_equal(mm.frame_signal(s, fl, hop).numpy(), g[f"c{i}_out_ones"])
        i += 1
    assert i == 7


def test_kaldi_golden_cells(mm, golden_dir):
    """Stored oracle cells (frames 0, 1, 191, 383) and the independent float64 cross-check."""
    g = np.load(f"{golden_dir}/kaldi_fbank.npz")
    frames = g["frames"]
    for i, kind in enumerate(KINDS):
        x = torch.from_numpy(synth(kind, 160000, 1234 + i))[None].cuda()
        norm = mm.preprocessing_audio(x).cpu().numpy()
        assert np.abs(norm[frames] - g[f"{kind}_norm"]).max() <= (5e-4 if kind not in HIGH_DYNAMIC_RANGE else 1e-3)
        raw = mm.fbank(x, sample_frequency=16000, window_type="hanning", num_mel_bins=128, frame_shift=25).cpu().numpy()
        d = np.abs(raw[frames] - g[f"{kind}_hf64"])
        assert np.percentile(d, 99) <= 1e-3 and np.median(d) <= 2e-5      # independent fp64 implementation, fp64-built weights
        s = g[f"{kind}_sums"]
        assert abs(norm.astype(np.float64).sum() - s[2]) <= 1e-5 * 384 * 128 * 4


@pytest.mark.parametrize("shape", [(6, 192), (15, 1003), (3, 2, 64), (2, 32), (5, 33)])
def test_har_stft_mag(mm, O, shape):
    rng = np.random.default_rng(4321)
    x = rng.standard_normal(shape).astype(np.float32)
    ref = O.stft_mag(x)
    got = mm.stft_mag(x)
    assert isinstance(got, np.ndarray) and got.dtype == np.float32
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-6


def test_har_many_groups_per_block(mm, O):
    """More row groups than resident blocks, a last group that is not full and a row count that is not a multiple of 4: every
    block of the wave-specialised STFT kernel alternates between its two LDS buffer sets several times and ends on a partial
    group; the phase-serial kernel (rows that are not 16-byte aligned) must give the same bits."""
    g = np.random.default_rng(77)
    x = g.standard_normal((12301, 192)).astype(np.float32)
    ref = O.stft_mag(x)
    got = mm.stft_mag(torch.from_numpy(x).cuda())
    assert got.shape == ref.shape
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-6
    shifted = torch.empty(12301 * 192 + 1, device="cuda")[1:].view(12301, 192)      # 4-byte aligned only
    shifted.copy_(torch.from_numpy(x))
    assert torch.equal(mm.stft_mag(shifted), got)


@pytest.mark.parametrize("t", [190, 191, 193, 50, 34])
def test_har_row_lengths_not_multiple_of_4(mm, O, t):
    """Packed rows whose length is not a multiple of 4 samples: the 16-byte pieces a group's samples arrive in straddle rows, the
    batch's last piece is partial (read float by float: nothing beyond the batch is touched), the last group is not full; a
    strided view of the same rows goes through the phase-serial kernel and must give the same bits."""
    g = np.random.default_rng(1000 + t)
    n = 4099
    x = g.standard_normal((n, t)).astype(np.float32)
    ref = O.stft_mag(x)
    guard = torch.full((n * t + 64,), float("nan"), device="cuda")          # NaNs right behind the batch: a read past its end shows
    packed = guard[: n * t].view(n, t)
    packed.copy_(torch.from_numpy(x))
    got = mm.stft_mag(packed)
    assert got.shape == ref.shape
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-6
    wide = torch.zeros(n, t + 3, device="cuda")
    wide[:, :t] = packed
    assert torch.equal(mm.stft_mag(wide[:, :t]), got)


def test_har_against_scipy(mm):
    import scipy.signal
    rng = np.random.default_rng(11)
    x = rng.standard_normal((64, 6, 192)).astype(np.float32)
    ref = np.abs(scipy.signal.stft(x, fs=100, nperseg=32, noverlap=28)[-1])
    got = mm.stft_mag(torch.from_numpy(x).cuda(), fs=100, nperseg=32, noverlap=28).cpu().numpy()
    assert got.shape == ref.shape == (64, 6, 17, 49)
    assert np.abs(got - ref).max() <= 1e-6


@pytest.mark.parametrize("shape,fl,hop", [((1, 10), 4, 3), ((2, 16000), 400, 160), ((3,), 8, 2), ((2, 3, 100), 16, 7),
                                          ((5,), 5, 2), ((7,), 5, 3)])
def test_frame_signal(mm, O, shape, fl, hop):
    rng = np.random.default_rng(3)
    s = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal(fl).astype(np.float32))
    ref = O.frame_signal(s, fl, hop, w)
    got = mm.frame_signal(s.cuda(), fl, hop, w.cuda()).cpu()
    assert got.shape == ref.shape
    assert torch.equal(got, ref)
    assert torch.equal(mm.frame_signal(s, fl, hop).cpu(), O.frame_signal(s, fl, hop))
    with pytest.raises(ValueError):
        mm.frame_signal(s, fl, hop, torch.ones(fl + 1))


@pytest.mark.parametrize("precision", ["float32", "float64"])
def test_long_utterance_logmel(mm, O, precision):
    """30 s utterances, n_fft 400 / hop 160 / 80 mel: the radix kernel (float32) and the any-length kernel with
    float64 accumulation, both against the float64 CPU restatement.  Tolerance 1e-4 abs on the normalised
    log-mel; the float32 CPU evaluation of the same formula sits within 2e-5 of the float64 one."""
    x = np.stack([synth("noise0.1", 480000, 99), synth("sine+noise", 480000, 100), synth("quiet", 480000, 101)])
    ref = np.stack([O.whisper_logmel(c) for c in x])
    got = mm.logmel.log_mel_spectrogram(torch.from_numpy(x).cuda(), precision=precision).cpu().numpy()
    assert got.shape == ref.shape == (3, 80, 3000)
    assert mm.plans.last_kernel() == ("rfft400" if precision == "float32" else "dftgen")
    mx, p999 = _stats(got - ref)
    print("logmel", precision, "max", mx, "p99.9", p999)
    assert mx <= 1e-4


def test_logmel_clamp_active(mm, O):
    """Utterances whose dynamic range exceeds 8 decades, so the (max - 8) clamp really changes cells: an impulse,
    half silence / half noise, pure silence (every cell at the floor) and a DC offset."""
    n = 32000
    half = synth("noise0.1", n, 41)
    half[: n // 2] = 0.0
    x = np.stack([synth("impulse", n, 40), half, synth("silence", n, 42), synth("dc", n, 43)])
    ref = np.stack([O.whisper_logmel(c) for c in x])
    for precision in ("float32", "float64"):
        got = mm.logmel.log_mel_spectrogram(torch.from_numpy(x).cuda(), precision=precision).cpu().numpy()
        floor_cells = int((ref[1] == ref[1].min()).sum())
        assert floor_cells > 1000                      # the clamp is active in the half-silent clip
        mx, p999 = _stats(got - ref)
        print("clamp", precision, "max", mx, "p99.9", p999)
        assert mx <= 1e-4


@pytest.mark.parametrize("n", [160, 161, 399, 400, 401, 1279, 1280, 1281, 4000, 16000, 48001])
def test_logmel_lengths(mm, O, n):
    """Ragged utterance lengths: partial last octets, rows shorter than one window (reflection wraps more than once)."""
    x = synth("noise0.1", n, 300 + n)
    ref = O.whisper_logmel(x)
    got = mm.logmel.log_mel_spectrogram(torch.from_numpy(x)[None].cuda()).cpu().numpy()[0]
    assert got.shape == ref.shape == (80, n // 160)
    mx, _ = _stats(got - ref)
    assert mx <= 1e-4


@pytest.mark.parametrize("hop,off", [(160, 0), (160, 1), (100, 0), (150, 0), (128, 2), (7, 0)])
def test_logmel_hops_and_alignment(mm, O, hop, off):
    """Other hops (generic staging layout, scalar loads when hop % 4 != 0) and rows that start off a 16-byte boundary."""
    n = 24000
    base = np.stack([synth("noise0.1", n + 4, 11), synth("uniform", n + 4, 12), synth("sine+noise", n + 4, 13)])
    x = base[:, off:off + n]
    ref = np.stack([O.whisper_logmel(c, hop=hop) for c in x])
    xg = torch.from_numpy(base).cuda()[:, off:off + n]
    got = mm.logmel.log_mel_spectrogram(xg, hop_length=hop).cpu().numpy()
    assert got.shape == ref.shape
    assert mm.plans.last_kernel() == "rfft400"
    mx, _ = _stats(got - ref)
    print("hop", hop, "off", off, "max", mx)
    assert mx <= 1e-4


@pytest.mark.parametrize("n_mels,kernel", [(40, "rfft400"), (64, "rfft400"), (80, "rfft400"), (128, "rfft400"), (23, "dftgen")])
def test_logmel_filter_bank_sizes(mm, O, n_mels, kernel):
    """Every instantiated mel stream shape of the radix kernel (2, 3 and 4 streams, with and without a split last
    stream), and a bank that fits none of them and therefore takes the any-length kernel."""
    x = np.stack([synth("noise0.1", 16000, 60 + n_mels), synth("sine+noise", 16000, 61 + n_mels)])
    ref = np.stack([O.whisper_logmel(c, n_mels=n_mels) for c in x])
    got = mm.logmel.log_mel_spectrogram(torch.from_numpy(x).cuda(), n_mels=n_mels).cpu().numpy()
    assert mm.plans.last_kernel() == kernel
    assert got.shape == ref.shape == (2, n_mels, 100)
    mx, _ = _stats(got - ref)
    print("n_mels", n_mels, kernel, "max", mx)
    assert mx <= 1e-4


def test_logmel_kernels_agree(mm):
    """The radix kernel and the any-length float64 kernel serve the same plan parameters: same shapes, same values
    to float32 FFT accuracy, on a batch large enough to wrap the persistent grid several times."""
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (0.1 * torch.randn(600, 16000, device="cuda", generator=g)).clamp_(-1, 1)
    a = mm.logmel.log_mel_spectrogram(x, precision="float32")
    b = mm.logmel.log_mel_spectrogram(x, precision="float64")
    assert a.shape == b.shape == (600, 80, 100)
    assert (a - b).abs().max().item() <= 1e-4
    # equal rows give bit-identical features wherever they sit in the batch
    y = x[:3].repeat(200, 1)
    c = mm.logmel.log_mel_spectrogram(y)
    assert torch.equal(c.view(200, 3, 80, 100), a[:3].unsqueeze(0).expand(200, -1, -1, -1))


def test_rfft400_time_major_no_clamp(mm):
    """Direct plan: centred zero-padded framing, 128 Slaney bands, ln(x + eps), time-major output, no clamp pass."""
    from multi_modal_amd import _native as N, melbank
    n, hop = 20000, 160
    x = np.stack([synth("noise0.1", n, 21), synth("uniform", n, 22)])
    fb = melbank.slaney_mel_filters(201, 128, 16000).astype(np.float32)
    bs, bl, bw = melbank.dense_to_bands(fb)
    win = melbank.periodic_hann(400).astype(np.float32)
    spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=400, win_len=400, win_off=0, hop=hop, pad_mode=N.PAD_ZERO,
                       frames_mode=N.FRAMES_CENTER, n_mel=128, power=2, log_kind=N.LOG_LN_ADD, log_floor=1e-6,
                       out_shift=1.5, out_scale=0.5, layout=N.LAYOUT_TIME_MAJOR)
    plan = N.Plan(spec, win, bs, bl, bw, device=0)
    assert plan.kernel == "rfft400"
    xg = torch.from_numpy(x).cuda()
    frames = 1 + n // hop
    out = torch.empty(2, frames, 128, device="cuda")
    plan.run_ptr(xg.data_ptr(), xg.stride(0), 2, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xp = np.pad(x.astype(np.float64), ((0, 0), (200, 200)))
    idx = hop * np.arange(frames)[:, None] + np.arange(400)[None, :]
    P = np.abs(np.fft.rfft(xp[:, idx] * win.astype(np.float64), axis=-1)) ** 2
    ref = (np.log(P @ fb.T.astype(np.float64) + 1e-6) + 1.5) * 0.5
    mx, _ = _stats(out.cpu().numpy() - ref)
    print("time-major ln_add max", mx)
    assert mx <= 1e-4


def test_fbank_n400_radix(mm, O):
    """kaldi fbank without power-of-two rounding, DC removal or pre-emphasis: 400-point spectrum, snip-edge framing."""
    x = synth("noise0.1", 48000, 31)
    kw = dict(num_mel_bins=80, round_to_power_of_two=False, remove_dc_offset=False, preemphasis_coefficient=0.0,
              window_type="hanning")
    ref = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=16000, **kw).numpy()
    got = mm.fbank(torch.from_numpy(x)[None].cuda(), sample_frequency=16000, **kw).cpu().numpy()
    assert got.shape == ref.shape
    assert mm.plans.last_kernel() == "rfft400"
    mx, p999 = _stats(got - ref)
    print("fbank n_fft=400 max", mx, "p99.9", p999)
    assert p999 <= 1e-4 and mx <= 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vggsound_half_precision_output(mm, dtype):
    """16-bit output of the fused log-mel kernel = its float32 output rounded to nearest even, bit for bit (the
    arithmetic is the same float32 pipeline), including padded rows, ragged lengths and a non-multiple-of-4 bank."""
    clips = torch.from_numpy(np.stack([synth(k, 160000, 77 + i) for i, k in enumerate(KINDS)])).cuda()
    ref = mm.preprocessing_audio_batch(clips)
    got = mm.preprocessing_audio_batch(clips, dtype=dtype)
    assert got.dtype == dtype and got.shape == ref.shape
    assert torch.equal(got, ref.to(dtype))
    short = clips[:, :12345]                     # 30 frames, the other 354 rows are padding
    assert torch.equal(mm.preprocessing_audio_batch(short, dtype=dtype), mm.preprocessing_audio_batch(short).to(dtype))
    kw = dict(sample_frequency=16000, num_mel_bins=23, frame_shift=10.0)
    assert torch.equal(mm.fbank_batch(clips[:, :48000], dtype=dtype, **kw), mm.fbank_batch(clips[:, :48000], **kw).to(dtype))
    # n_fft 400 with DC removal and pre-emphasis runs on the any-length DFT kernel, which rounds in its stores as well
    kw400 = dict(kw, round_to_power_of_two=False)
    assert torch.equal(mm.fbank_batch(clips[:, :48000], dtype=dtype, **kw400), mm.fbank_batch(clips[:, :48000], **kw400).to(dtype))
    kwr = dict(kw400, num_mel_bins=80, remove_dc_offset=False, preemphasis_coefficient=0.0)      # the radix n_fft 400 kernel
    assert torch.equal(mm.fbank_batch(clips[:, :48000], dtype=dtype, **kwr), mm.fbank_batch(clips[:, :48000], **kwr).to(dtype))
    assert mm.plans.last_kernel() == "rfft400"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_epic_half_precision_output(mm, dtype):
    """EPIC log-spectrograms stored as float16 / bfloat16 (the (B, 1, 256, 256) tensor an encoder under autocast would cast):
    bit for bit the rounding of the float32 result, for the matrix-pipe form, the vector form (mel bank), mixed slice lengths
    from packed recordings, and PCM input."""
    x = torch.from_numpy(np.stack([synth(k, 30696, 300 + i) for i, k in enumerate(("noise0.1", "quiet", "sine+noise"))])).cuda()
    ref = mm.log_specgram_batch(x, div255=True)
    got = mm.log_specgram_batch(x, div255=True, dtype=dtype)
    assert got.dtype == dtype and torch.equal(got, ref.to(dtype))
    lens = [30672, 30696, 30672]
    assert torch.equal(mm.log_specgram_batch(x, lengths=lens, dtype=dtype), mm.log_specgram_batch(x, lengths=lens).to(dtype))
    recs = [torch.from_numpy(synth("noise0.1", 24000 * 8, 310)), torch.from_numpy(synth("uniform", 24000 * 5, 311))]
    args = (recs, [0, 1, 0, 1], [0, 30, 150, 90], [0, 5, 20, 40], [60.0, 60.0, 60.0, 60.0])
    a = mm.epic.extract_sound_features(*args)
    b = mm.epic.extract_sound_features(*args, dtype=dtype)
    assert b.shape == a.shape == (4, 1, 256, 256) and torch.equal(b, a.to(dtype))
    pcm = (x * 32767.0).round().to(torch.int16)
    assert torch.equal(mm.log_specgram_batch(pcm, dtype=dtype), mm.log_specgram_batch(pcm).to(dtype))
    with pytest.raises(TypeError):
        mm.log_specgram_batch(x, dtype=torch.float64)


def test_cal_norm(mm):
    rng = np.random.default_rng(0)
    f = torch.from_numpy(rng.standard_normal((7, 384, 128)).astype(np.float32) * 4.4 - 5.2)
    mean, std = mm.stats.cal_norm(f.cuda())
    assert abs(mean - f.double().mean().item()) < 1e-9 * 1e3
    assert abs(std - f.double().std(unbiased=False).item()) < 1e-6


def test_epic_batch_averaged_statistics(mm):
    """The EPIC statistics pass (epic_kitchen.py:275-290): mean of per-batch means, sqrt of the mean of per-batch UNBIASED variances,
    against the same formula in torch float64; the pooled RunningNorm figure differs when the batch means drift."""
    g = torch.Generator(device="cuda").manual_seed(8)
    batches = [torch.randn(48, 1, 64, 64, device="cuda", generator=g) * (1.0 + 0.1 * i) + 0.5 * i for i in range(5)]
    norm = mm.stats.BatchAveragedNorm()
    pooled = mm.stats.RunningNorm()
    for bt in batches:
        norm.update(bt)
        pooled.update(bt)
    mean, std = norm.finalize()
    ref_mean = sum(bt.double().mean().item() for bt in batches) / len(batches)
    ref_std = (sum(bt.double().std().item() ** 2 for bt in batches) / len(batches)) ** 0.5
    assert abs(mean - ref_mean) <= 1e-9 and abs(std - ref_std) <= 1e-9
    assert pooled.finalize()[1] > std + 0.1          # pooled variance also holds the drift of the batch means


def test_full_size_properties(mm):
    """The headline benchmark configuration at full size (4096 VGGSound clips): properties that do not need the oracle.
    Every clip is one of 8 distinct waveforms, so equal clips must give bit-identical features wherever
    they sit in the batch, and the 8 distinct ones are checked against the small-batch result."""
    base = torch.from_numpy(np.stack([synth("noise0.1", 160000, 1234 + i) for i in range(8)])).cuda()
    small = mm.preprocessing_audio_batch(base)
    big_in = base.repeat(512, 1)
    big = mm.preprocessing_audio_batch(big_in)
    assert big.shape == (4096, 384, 128)
    assert torch.equal(big.view(512, 8, 384, 128), small.unsqueeze(0).expand(512, -1, -1, -1))
    assert torch.isfinite(big).all()
    # linearity of the pre-log pipeline: scaling the waveform by 2 adds ln(4) to every un-floored cell
    a = mm.fbank_batch(base[:2], sample_frequency=16000, window_type="hanning", num_mel_bins=128, frame_shift=25)
    b = mm.fbank_batch(base[:2] * 2.0, sample_frequency=16000, window_type="hanning", num_mel_bins=128, frame_shift=25)
    mask = a > -15.0
    assert torch.allclose((b - a)[mask], torch.full_like(a[mask], float(np.log(4.0))), atol=2e-5)


def test_full_size_properties_har(mm):
    """The IMU benchmark configuration at full size (262 144 windows x 6 axes): equal windows give bit-identical spectrograms
    wherever they sit in the batch; scaling the input by a power of two scales the magnitudes exactly."""
    g = torch.Generator(device="cuda").manual_seed(4321)
    base = torch.randn(64, 6, 192, device="cuda", generator=g)
    small = mm.stft_mag(base)
    big = mm.stft_mag(base.repeat(4096, 1, 1))
    assert big.shape == (262144, 6, 17, 49)
    assert torch.equal(big.view(4096, 64, 6, 17, 49), small.unsqueeze(0).expand(4096, -1, -1, -1, -1))
    assert torch.equal(mm.stft_mag(base * 4.0), small * 4.0)
    assert torch.isfinite(big).all()


def test_full_size_properties_logmel(mm):
    """The long-utterance benchmark configuration per GPU (1024 utterances of 30 s): position independence of the radix kernel, and
    the clamp invariant of every utterance (no cell below max - 2 after the (x + 4) / 4 map)."""
    g = torch.Generator(device="cuda").manual_seed(99)
    base = (0.1 * torch.randn(8, 480000, device="cuda", generator=g)).clamp_(-1, 1)
    small = mm.logmel.log_mel_spectrogram(base)
    big = mm.logmel.log_mel_spectrogram(base.repeat(128, 1))
    assert big.shape == (1024, 80, 3000)
    assert torch.equal(big.view(128, 8, 80, 3000), small.unsqueeze(0).expand(128, -1, -1, -1))
    mx = big.amax(dim=(1, 2))
    mn = big.amin(dim=(1, 2))
    assert torch.all(mn >= mx - 2.0 - 1e-6)
    assert torch.isfinite(big).all()


@pytest.mark.parametrize("which", ["fbank512", "rfft400"])
def test_radix_kernels_read_slices(mm, which):
    """mmfe_run_slices through the radix kernels: rows given as (recording, offset) slices of packed recordings give
    the same features as the same rows handed over as a strided batch (odd offsets: scalar-load variants)."""
    from multi_modal_amd import kaldi, logmel
    recs = [torch.from_numpy(synth("noise0.1", n, 500 + i)).cuda() for i, n in enumerate((50000, 33333, 41001))]
    packed = torch.cat(recs)
    starts = torch.tensor([0, 50000, 83333], dtype=torch.int64, device="cuda")
    rec_index = [0, 2, 1, 1, 0, 2, 0]
    offsets = [0, 1, 7, 1600, 12345, 17000, 25999]
    n = 24000
    rows = torch.stack([recs[r][o:o + n] for r, o in zip(rec_index, offsets)])
    if which == "fbank512":
        plan = kaldi._fbank_plan(0, sample_frequency=16000.0, frame_length=25.0, frame_shift=10.0, num_mel_bins=80,
                                 window_type="povey", blackman_coeff=0.42, preemphasis_coefficient=0.97, remove_dc_offset=True,
                                 round_to_power_of_two=True, low_freq=20.0, high_freq=0.0, use_log_fbank=True)
    else:
        plan = logmel._plan(0, 400, 160, 80, 16000, False)
    assert plan.kernel == which
    ref = mm.plans.run_rows(plan, rows)
    got = torch.empty_like(ref)
    d_index = torch.tensor(rec_index, dtype=torch.int32, device="cuda")
    d_off = torch.tensor(offsets, dtype=torch.int64, device="cuda")
    plan.run_slices_ptr(packed.data_ptr(), starts.data_ptr(), d_index.data_ptr(), d_off.data_ptr(), len(offsets), n,
                        got.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if which == "rfft400":
        assert torch.equal(got, ref)          # same arithmetic, only the loads differ
    else:
        # slices take the scalar-load variant of the kernel, whose 16-slice frame walk sums the DC offset in another
        # order than the 13-slice 8-byte-load variant: equal to float32 rounding, not bit for bit
        assert (got - ref).abs().max().item() <= 5e-4



def test_plans_shared_between_threads_and_streams(mm):
    """One cached plan serves several host threads, each launching on a stream of its own (ctypes drops the GIL for the
    call, so the launches really interleave): every thread gets the bits the serial calls gave."""
    import threading
    clips = [torch.from_numpy(np.stack([synth("noise0.1", 160000, 900 + 4 * t + i) for i in range(4)])).cuda() for t in range(4)]
    g = torch.Generator(device="cuda").manual_seed(5)
    imu = [torch.randn(512, 6, 192, device="cuda", generator=g) for _ in range(4)]
    want = [(mm.preprocessing_audio_batch(c), mm.stft_mag(x)) for c, x in zip(clips, imu)]
    torch.cuda.synchronize()
    got, errs = [None] * 4, []

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(25):
                    y = mm.preprocessing_audio_batch(clips[t])
                    z = mm.stft_mag(imu[t])
                s.synchronize()
            got[t] = (y, z)
        except Exception as e:      # surfaced by the assert below
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t in range(4):
        assert torch.equal(got[t][0], want[t][0]) and torch.equal(got[t][1], want[t][1])


def test_clamp_plans_are_per_stream(mm):
    """The long-utterance log-mel (clamp_below_max: two passes that exchange per-row extremes through a scratch the plan
    owns) called from two host threads on two streams: the wrapper keeps one plan per stream, so the calls may overlap
    on the device and still give the bits of the serial calls."""
    import threading
    g = torch.Generator(device="cuda").manual_seed(11)
    xs = [torch.randn(64, 160000, device="cuda", generator=g) * torch.logspace(-3, 0, 64, device="cuda")[:, None] for _ in range(2)]
    want = [mm.logmel.log_mel_spectrogram(x) for x in xs]
    torch.cuda.synchronize()
    got, errs, plans = [None, None], [], [None, None]

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(40):
                    y = mm.logmel.log_mel_spectrogram(xs[t])
                plans[t] = mm.logmel._plan(0, 400, 160, 80, 16000, False, stream=s.cuda_stream)
                s.synchronize()
            got[t] = y
        except Exception as e:      # surfaced by the assert below
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    assert plans[0] is not plans[1]
    for t in range(2):
        assert torch.equal(got[t], want[t])


def test_plan_creation_rejects_what_no_launch_could_serve(mm):
    """Specs that used to be accepted and then failed or were silently computed differently at run time."""
    from multi_modal_amd import _native as N
    from multi_modal_amd import melbank
    # a 50 ms kaldi window at 16 kHz (800 taps, n_fft 1024) needs more LDS than a CU has: refused at plan creation
    with pytest.raises(N.MmfeError, match="LDS"):
        mm.fbank(torch.zeros(1, 16000, device="cuda"), frame_length=50.0, sample_frequency=16000)
    # the 32-point STFT kernel implements zero extension without clamp / dropped frame only
    win = melbank.periodic_hann(32)
    for bad in (dict(pad_mode=N.PAD_REFLECT), dict(clamp_below_max=1.0), dict(drop_last=1)):
        kw = dict(kind=N.KIND_STFT_MAG, n_fft=32, win_len=32, hop=4, pad_mode=N.PAD_ZERO, frames_mode=N.FRAMES_SCIPY, power=1,
                  spec_scale=1.0 / 16.0, layout=N.LAYOUT_FEAT_MAJOR, window_f64=1)
        kw.update(bad)
        with pytest.raises(N.MmfeError, match="stft32"):
            N.Plan(N.make_spec(**kw), win, device=0)


def test_cal_norm_odd_offsets_and_devices(mm):
    """cal_norm on a contiguous view that starts 4 bytes into its storage (not 16-byte aligned) and on a CPU tensor."""
    g = torch.Generator(device="cuda").manual_seed(3)
    f = torch.randn(100003, device="cuda", generator=g)
    v = f[1:]
    mean, std = mm.stats.cal_norm(v)
    assert abs(mean - float(v.double().mean())) < 1e-9 and abs(std - float(v.double().std(unbiased=False))) < 1e-9
    mean_c, std_c = mm.stats.cal_norm(v.cpu())
    assert abs(mean_c - mean) < 1e-12 and abs(std_c - std) < 1e-12      # float64 atomics: the order of the adds varies


def _pcm(kind, n, seed):
    """A synthetic waveform quantised the way a 16-bit FLAC / pcm_s16le file holds it."""
    return np.clip(np.round(synth(kind, n, seed).astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


def test_vggsound_pcm_input(mm, O):
    """16-bit PCM clips decoded inside the n_fft 512 kernel: bit for bit the features of the decoded float32
    waveform (s / 32768), for the 4-byte-load variant, the odd-offset variant, ragged lengths and 16-bit output;
    and the oracle's values on the decoded waveform."""
    s = np.stack([_pcm(k, 160002, 300 + i) for i, k in enumerate(KINDS)])
    f = s.astype(np.float32) / np.float32(32768.0)
    d_s, d_f = torch.from_numpy(s).cuda(), torch.from_numpy(f).cuda()
    got = mm.preprocessing_audio_batch(d_s[:, :160000].contiguous())
    assert mm.plans.last_kernel() == "fbank512" and got.dtype == torch.float32
    assert torch.equal(got, mm.preprocessing_audio_batch(d_f[:, :160000].contiguous()))
    ref = np.stack([O.preprocessing_audio(torch.from_numpy(c[:160000])[None]).numpy() for c in f])
    assert np.abs(got.cpu().numpy() - ref).max() <= 5e-4
    # rows that start on an odd sample (2-byte aligned only) and an odd row stride: the 2-byte-load variant against
    # the float variant that also loads sample by sample
    assert torch.equal(mm.preprocessing_audio_batch(d_s[:, 1:160001]), mm.preprocessing_audio_batch(d_f[:, 1:160001]))
    # even offset, even stride, but a view: 4-byte loads
    assert torch.equal(mm.preprocessing_audio_batch(d_s[:, 2:160002]), mm.preprocessing_audio_batch(d_f[:, 2:160002]))
    for n in (399, 400, 12345):
        assert torch.equal(mm.preprocessing_audio_batch(d_s[:, :n].contiguous()), mm.preprocessing_audio_batch(d_f[:, :n].contiguous()))
    assert torch.equal(mm.preprocessing_audio_batch(d_s[:, :160000].contiguous(), dtype=torch.float16), got.to(torch.float16))
    # CPU int16 tensor in -> CPU float32 tensor out
    one = mm.preprocessing_audio(torch.from_numpy(s[0:1, :160000]))
    assert one.device.type == "cpu" and np.array_equal(one.numpy(), got[0].cpu().numpy())


def test_fbank_pcm_other_parameters(mm, O):
    """PCM input through other log-mel parameter sets: a 512-point plan without DC removal / pre-emphasis, an odd
    window (2-byte loads), and windows served by the any-length kernel (n_fft 256 and 400)."""
    s = torch.from_numpy(np.stack([_pcm("noise0.1", 48000, 31), _pcm("uniform", 48000, 32)])).cuda()
    f = s.to(torch.float32) / 32768.0
    cases = [(dict(sample_frequency=16000, num_mel_bins=40, remove_dc_offset=False, preemphasis_coefficient=0.0, window_type="rectangular"), "fbank512"),
             (dict(sample_frequency=16000, num_mel_bins=23), "fbank512"),
             (dict(sample_frequency=11025.0, num_mel_bins=64), "fbank512"),
             (dict(sample_frequency=8000, num_mel_bins=40), "dftgen"),
             (dict(sample_frequency=16000, num_mel_bins=80, round_to_power_of_two=False), "dftgen")]
    for kw, kernel in cases:
        got = mm.fbank_batch(s, **kw)
        assert mm.plans.last_kernel() == kernel, (kw, mm.plans.last_kernel())
        if kernel == "fbank512":
            assert torch.equal(got, mm.fbank_batch(f, **kw)), kw
        ref = O.kaldi_fbank(f[:1].cpu(), **kw).numpy()
        mx, p999 = _stats(got[0].cpu().numpy() - ref)
        assert p999 <= 1e-4 and mx <= 2e-3, (kw, mx, p999)


def test_epic_pcm_input(mm, O):
    """pcm_s16le recordings (what extract_audio.py writes) straight into the log-spectrogram kernel: the bits of the
    decoded float32 path, per slice and through the packed-recordings entry point."""
    s = _pcm("noise0.1", 30672, 8)
    f = s.astype(np.float32) / np.float32(32768.0)
    got = mm.log_specgram(s)
    assert got.dtype == np.float32 and np.array_equal(got, mm.log_specgram(f))
    assert np.abs(got - O.log_specgram(f)).max() <= 1e-4
    recs = [_pcm("noise0.1", 24000 * 5, 1), _pcm("uniform", 24000 * 3 + 17, 2)]
    args = ([0, 0, 1, 1, 0], [100, 130, 30, 60, 170], [3, 0, 7, 2, 1], [60.0, 60.0, 50.0, 50.0, 60.0])
    a = mm.epic.extract_sound_features([torch.from_numpy(r) for r in recs], *args)
    b = mm.epic.extract_sound_features([torch.from_numpy(r.astype(np.float32) / np.float32(32768.0)) for r in recs], *args)
    assert a.shape == (5, 1, 256, 256) and torch.equal(a, b)


def test_logmel_pcm_input(mm):
    """16-bit PCM utterances through the n_fft 400 radix kernel (8-byte loads, scalar loads at odd offsets, row-end
    octets) and through the float64 any-length kernel: the bits of the decoded float32 path."""
    s16 = torch.from_numpy(np.stack([_pcm("noise0.1", 48004, 41), _pcm("sine+noise", 48004, 42)])).cuda()
    f32 = s16.to(torch.float32) / 32768.0
    for off, n, hop in [(0, 48000, 160), (4, 48000, 160), (1, 47999, 160), (0, 20000, 100), (3, 5000, 37)]:
        a = mm.logmel.log_mel_spectrogram(s16[:, off:off + n], hop_length=hop)
        assert mm.plans.last_kernel() == "rfft400"
        assert torch.equal(a, mm.logmel.log_mel_spectrogram(f32[:, off:off + n], hop_length=hop)), (off, n, hop)
    a = mm.logmel.log_mel_spectrogram(s16[:, :16000], precision="float64")
    assert mm.plans.last_kernel() == "dftgen"
    assert torch.equal(a, mm.logmel.log_mel_spectrogram(f32[:, :16000], precision="float64"))
    kw = dict(sample_frequency=16000, num_mel_bins=80, round_to_power_of_two=False, remove_dc_offset=False, preemphasis_coefficient=0.0)
    b = mm.fbank_batch(s16[:, :48000].contiguous(), **kw)
    assert mm.plans.last_kernel() == "rfft400"
    assert torch.equal(b, mm.fbank_batch(f32[:, :48000].contiguous(), **kw))


def test_pcm_rejected_where_not_implemented(mm):
    x = torch.zeros(2, 6, 192, dtype=torch.int16, device="cuda")
    with pytest.raises(TypeError):
        mm.stft_mag(x)
    from multi_modal_amd import _native as N
    spec = N.make_spec(kind=N.KIND_STFT_MAG, n_fft=32, win_len=32, hop=4, frames_mode=N.FRAMES_SCIPY, pad_mode=N.PAD_ZERO,
                       power=1, in_dtype=N.IN_S16)
    with pytest.raises(mm.MmfeError, match="16-bit PCM"):
        N.Plan(spec, np.ones(32, np.float32), device=0)
    # framing moves values: int16 in, int16 out, exact
    sig = torch.arange(-50, 50, dtype=torch.int16)
    fr = mm.frame_signal(sig.cuda(), 16, 8)
    assert fr.dtype == torch.int16 and fr.shape[-1] == 16


@pytest.mark.parametrize("n_fft,win,hop,pad", [(511, 240, 120, "constant"), (511, 239, 120, "constant"), (256, 120, 50, "reflect"),
                                              (64, 48, 16, "constant"), (400, 33, 7, "reflect"), (1024, 200, 100, "constant"),
                                              (2048, 200, 100, "constant")])
def test_plain_spectrogram_matrix_form(mm, O, n_fft, win, hop, pad):
    """Plain spectrograms with up to 120 mirrored tap pairs take the float64 matrix-pipe form of the any-length DFT
    kernel (even and odd windows, 8 / 16 / 30 k-step variants, both layouts, |X| and |X|^2, PCM input, ragged frame
    tiles); checked against the oracle's centred STFT and against the vector form of the same kernel."""
    import os
    from multi_modal_amd import _native as N, melbank, plans
    x = np.stack([synth("noise0.1", 5000, 11), synth("sine+noise", 5000, 12)])
    stft = [O.centred_stft(r, n_fft, hop, win, pad_mode=pad) for r in x]
    for layout, power, log_kind in [(N.LAYOUT_FEAT_MAJOR, 2, N.LOG_LN_ADD), (N.LAYOUT_TIME_MAJOR, 1, N.LOG_NONE)]:
        spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=n_fft, win_len=win, win_off=(n_fft - win) // 2, hop=hop,
                           pad_mode=N.PAD_ZERO if pad == "constant" else N.PAD_REFLECT, frames_mode=N.FRAMES_CENTER, power=power,
                           log_kind=log_kind, log_floor=1e-6, layout=layout, window_f64=1)
        plan = N.Plan(spec, melbank.periodic_hann(win), device=0)
        assert plan.kernel == "dftgen"
        got = plans.run_rows(plan, torch.from_numpy(x).cuda()).cpu().numpy()
        os.environ["MMFE_DFT_FORM"] = "valu"
        try:
            alt = plans.run_rows(plan, torch.from_numpy(x).cuda()).cpu().numpy()
        finally:
            del os.environ["MMFE_DFT_FORM"]
        plan.close()
        if power == 2:
            ref = np.stack([np.log(np.real(s * np.conj(s)) + np.float32(1e-6)) for s in stft])
            tol = 1e-4
        else:
            ref = np.stack([np.abs(s).T for s in stft])
            tol = 2e-6 * max(1.0, float(np.abs(ref).max()))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= tol, (layout, np.abs(got - ref).max())
        assert np.abs(got - alt).max() <= tol
    # 16-bit PCM rows through the same form: bits of the decoded float path
    s16 = torch.from_numpy(np.stack([_pcm("noise0.1", 5000, 13)])).cuda()
    for in_dtype, rows in ((N.IN_S16, s16), (N.IN_F32, s16.to(torch.float32) / 32768.0)):
        spec = N.make_spec(kind=N.KIND_LOGSPEC, n_fft=n_fft, win_len=win, win_off=(n_fft - win) // 2, hop=hop, pad_mode=N.PAD_ZERO,
                           frames_mode=N.FRAMES_CENTER, power=2, log_kind=N.LOG_LN_ADD, log_floor=1e-6, layout=N.LAYOUT_FEAT_MAJOR,
                           window_f64=1, in_dtype=in_dtype)
        plan = N.Plan(spec, melbank.periodic_hann(win), device=0)
        out = plans.run_rows(plan, rows).clone()
        plan.close()
        if in_dtype == N.IN_S16:
            first = out
        else:
            assert torch.equal(first, out)


def test_runs_are_capturable_in_a_hip_graph(mm):
    """mmfe_run neither allocates nor synchronises, so a warmed-up call can be recorded into a HIP graph and replayed
    on new data in the same buffers (all kernel families, including the two-pass clamp of the long-utterance log-mel)."""
    g = torch.Generator(device="cuda").manual_seed(17)
    cases = [
        (lambda x, o: mm.preprocessing_audio_batch(x, out=o), (0.1 * torch.randn(8, 160000, device="cuda", generator=g)), (8, 384, 128)),
        (lambda x, o: mm.stft_mag(x, out=o), torch.randn(64, 6, 192, device="cuda", generator=g), (64, 6, 17, 49)),
        (lambda x, o: mm.log_specgram_batch(x, div255=True, out=o), (0.1 * torch.randn(4, 30672, device="cuda", generator=g)), (4, 256, 256)),
        (lambda x, o: mm.logmel.log_mel_spectrogram(x, out=o), (0.1 * torch.randn(3, 48000, device="cuda", generator=g)), (3, 80, 300)),
    ]
    for fn, x, shape in cases:
        out = torch.empty(shape, device="cuda")
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(x, out)                               # warm-up ON THE CAPTURE STREAM: plan creation (the log-mel wrapper keeps one
            side.synchronize()                       # plan per stream) and per-kernel launch attributes happen here
            with torch.cuda.graph(graph, stream=side):
                fn(x, out)
        torch.cuda.current_stream().wait_stream(side)
        x2 = torch.roll(x, shifts=7, dims=-1) * 0.5
        want = fn(x2, torch.empty(shape, device="cuda")).clone()
        x.copy_(x2)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), shape


@pytest.mark.parametrize("nperseg,noverlap,t", [(16, 12, 100), (64, 48, 500), (100, 75, 1000), (33, 5, 400), (256, 128, 3000),
                                               (300, 299, 700), (8, 0, 64)])
def test_stft_mag_other_segment_lengths(mm, O, nperseg, noverlap, t):
    """Segment lengths other than the reference's 32 run on the any-length DFT kernel with scipy's framing (matrix-pipe
    form up to 240 taps, vector form beyond): against the oracle and against scipy itself."""
    import scipy.signal
    rng = np.random.default_rng(nperseg)
    x = rng.standard_normal((3, 2, t)).astype(np.float32)
    got = mm.stft_mag(torch.from_numpy(x).cuda(), nperseg=nperseg, noverlap=noverlap).cpu().numpy()
    assert mm.plans.last_kernel() == "dftgen"
    ref = O.stft_mag(x, nperseg=nperseg, noverlap=noverlap)
    live = np.abs(scipy.signal.stft(x, fs=100, nperseg=nperseg, noverlap=noverlap)[-1])
    assert got.shape == ref.shape == live.shape
    assert np.abs(got - ref).max() <= 1e-6 and np.abs(got - live).max() <= 1e-6


def test_log_stage_against_libm(mm):
    """The log stage alone: the same plan once without the log (the kernel's own mel energies E) and once with it; every cell
    must equal the libm value of ln(max(E, eps)) to 4e-6 absolute while E sweeps more than 2^-23 ... 2^10 (the hardware
    base-2 logarithm plus one fused multiply-add; floored cells carry the host-libm value exactly)."""
    base = synth("noise0.1", 16000, 77)
    amps = 2.0 ** np.arange(-14, 3.5, 0.5)                     # energies scale with the square: 2^-28 ... 2^6 times the base level
    x = torch.from_numpy(np.stack([np.clip(a * base * 4.0, -8.0, 8.0) for a in amps]).astype(np.float32)).cuda()
    kw = dict(sample_frequency=16000.0, window_type="hanning", num_mel_bins=128, frame_shift=25.0)
    lin = mm.kaldi.fbank_batch(x, use_log_fbank=False, **kw).cpu().numpy().astype(np.float64)
    got = mm.kaldi.fbank_batch(x, use_log_fbank=True, **kw).cpu().numpy().astype(np.float64)
    eps = float(np.finfo(np.float32).eps)
    live = lin[:, :, [m for m in range(128) if m != 3]]        # filter 3 of this bank is empty
    assert live.min() < 2.0 ** -23 and live.max() > 2.0 ** 10, (live.min(), live.max())
    want = np.log(np.maximum(lin, eps))
    d = np.abs(got - want)
    print("log stage: energies", live.min(), "...", live.max(), "max |d|", d.max())
    assert d.max() <= 4e-6
    floored = lin <= eps
    assert floored.any() and np.all(got[floored] == np.float64(np.float32(np.log(np.float32(eps)))))


def test_mel_weights_are_the_kaldi_triangles(mm):
    """The mel weights the kernels are fed, against an implementation that shares no code with them: the float64 kaldi bank of
    transformers.audio_utils (when importable) and literal cells of that bank."""
    from multi_modal_amd import melbank
    bank = torch.nn.functional.pad(melbank.kaldi_mel_banks(128, 512, 16000.0, 20.0, 0.0), (0, 1)).numpy()
    cells = {(0, 1): 0.8026399861583207, (127, 250): 0.8720851867439546, (127, 255): 0.1862875061163922, (0, 2): 0.0, (100, 150): 0.0}
    for (m, k), v in cells.items():
        assert abs(float(bank[m, k]) - v) <= 2e-5
    try:
        from transformers.audio_utils import mel_filter_bank
    except Exception:
        pytest.skip("transformers not importable")
    hf = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20.0, max_frequency=8000.0, sampling_rate=16000,
                         norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    assert np.abs(hf.T - bank).max() <= 2e-5
    # and through the kernel: linear mel energies of white noise equal (float64 power spectrum of the oracle's windowed frames) x bank
    x = torch.from_numpy(synth("noise0.1", 4000, 5))[None]
    lin = mm.kaldi.fbank_batch(x.cuda(), use_log_fbank=False, sample_frequency=16000.0, window_type="hanning", num_mel_bins=128,
                               frame_shift=25.0).cpu().numpy()[0].astype(np.float64)
    fr = x[0].double().unfold(0, 400, 400)
    fr = fr - fr.mean(1, keepdim=True)
    fr = fr - 0.97 * torch.cat([fr[:, :1], fr[:, :-1]], 1)
    fr = fr * torch.hann_window(400, periodic=False, dtype=torch.float64)
    pw = torch.fft.rfft(torch.nn.functional.pad(fr, (0, 112)), dim=1).abs().pow(2).numpy()
    want = pw @ hf
    live = want > 1e-3
    assert np.abs(lin[live] / want[live] - 1.0).max() <= 2e-4


def test_epic_golden_fixture(mm, golden_dir):
    """Stored cells of the EPIC log-spectrogram fixture (rows of the CPU restatement and of the float64 numpy evaluation)."""
    g = np.load(f"{golden_dir}/epic_logspec.npz")
    rows = g["rows"]
    for n in (30672, 30696):
        for kind in ("noise0.1", "quiet", "silence", "sine+noise"):
            y = mm.log_specgram(synth(kind, n, 4242 + n))
            assert np.abs(y[rows] - g[f"{kind}_{n}_rows"]).max() <= 1e-4
            assert np.abs(y[rows] - g[f"{kind}_{n}_np64_rows"]).max() <= 1e-4
            assert np.abs(y[:, [0, 1, 128, 255]] - g[f"{kind}_{n}_cols"]).max() <= 1e-4


def test_logmel_golden_fixture(mm, golden_dir):
    """Stored columns of the long-utterance fixture: the CPU restatement and transformers' WhisperFeatureExtractor output."""
    g = np.load(f"{golden_dir}/logmel.npz")
    x = torch.from_numpy(synth("noise0.1", 480000, 99))[None].cuda()
    for precision in ("float32", "float64"):
        y = mm.logmel.log_mel_spectrogram(x, precision=precision).cpu().numpy()[0]
        assert np.abs(y[:, g["cols"]] - g["oracle"]).max() <= 1e-4
        assert np.abs(y[:, g["cols"]] - g["hf"]).max() <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vggsound_feature_major_layout(mm, dtype):
    """(B, 1, 128, 384) written directly equals the transpose of the time-major output, bit for bit: whole batches, short and
    ragged clips (frame counts that are not multiples of 4), the any-length and n_fft 400 kernels behind kaldi.fbank_batch."""
    x = torch.from_numpy(np.stack([synth("noise0.1", 160000, 1), synth("sine+noise", 160000, 2), synth("quiet", 160000, 3)])).cuda()
    tm = mm.preprocessing_audio_batch(x, dtype=dtype)
    fm = mm.preprocessing_audio_batch(x, dtype=dtype, nchw=True)
    assert fm.shape == (3, 1, 128, 384) and fm.is_contiguous()
    assert torch.equal(fm[:, 0], tm.transpose(1, 2))
    lengths = [160000, 50123, 801]
    tm = mm.preprocessing_audio_batch(x, dtype=dtype, lengths=lengths)
    fm = mm.preprocessing_audio_batch(x, dtype=dtype, lengths=lengths, nchw=True)
    assert torch.equal(fm[:, 0], tm.transpose(1, 2))
    short = x[:, :50123]                                       # 125 frames: the last quad holds one frame
    tm = mm.preprocessing_audio_batch(short, dtype=dtype)
    fm = mm.preprocessing_audio_batch(short, dtype=dtype, nchw=True)
    assert torch.equal(fm[:, 0], tm.transpose(1, 2))
    if dtype == torch.float32:
        for kw in (dict(frame_length=25.0, frame_shift=10.0, num_mel_bins=80, round_to_power_of_two=False, remove_dc_offset=False,
                        preemphasis_coefficient=0.0),            # n_fft 400 radix kernel
                   dict(frame_length=20.0, frame_shift=10.0, num_mel_bins=40, round_to_power_of_two=False),      # any-length kernel
                   dict(frame_length=25.0, frame_shift=10.0, num_mel_bins=23)):                                  # n_fft 512, VALU mel form
            a = mm.kaldi.fbank_batch(short, sample_frequency=16000.0, **kw)
            b = mm.kaldi.fbank_batch(short, sample_frequency=16000.0, feature_major=True, **kw)
            print(kw, mm.plans.last_kernel())
            assert b.shape == (a.shape[0], a.shape[2], a.shape[1]) and torch.equal(b, a.transpose(1, 2))


# ---- polyphase resampler (the step in front of the path: librosa.load(sr=16000) / ffmpeg -ar 24000) ----------------
# Checker: scipy.signal.resample_poly itself (float32 rows).  Parity with librosa's own default resampler (soxr, absent
# here) is unpinned.  Tolerance: 2e-6 abs for |x| <= 1 (float32 FIR sums of up to 60 taps per output in another order
# than scipy's upfirdn).

def test_resample_golden_scipy(mm, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "resample.npz"))
    for tag in sorted(k[:-2] for k in g.files if k.endswith("_y")):
        n, seed, up, down = (int(v) for v in g[tag + "_params"])
        x = synth("noise0.1", n, seed)
        got = mm.resample_batch(torch.from_numpy(x)[None].cuda(), down, up).cpu().numpy()[0]
        assert got.shape == g[tag + "_y"].shape
        mx = np.abs(got - g[tag + "_y"]).max()
        print(tag, "max", mx)
        assert mx <= 2e-6


@pytest.mark.parametrize("orig,target", [(44100, 16000), (48000, 16000), (48000, 24000), (22050, 16000), (16000, 44100), (8000, 16000), (44100, 24000)])
@pytest.mark.parametrize("n", [1, 7, 440, 441, 4410, 44101])
def test_resample_against_scipy(mm, O, orig, target, n):
    x = np.stack([synth(k, n, 800 + i + n) for i, k in enumerate(("noise0.1", "uniform", "impulse", "dc"))])
    ref = O.resample_poly(x, orig, target)
    got = mm.resample_batch(torch.from_numpy(x).cuda(), orig, target).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-6


def test_resample_load_duration_pcm_and_fixed_length(mm, O):
    """librosa.load(sr=16000, duration=10) on a 44.1 kHz track (vggsound.py:114): 441 000 samples -> exactly 160 000; 16-bit PCM
    rows give the bits of their decoded floats; out_samples cuts and zero-pads; strided rows and `out=`."""
    rng = np.random.default_rng(3)
    pcm = rng.integers(-20000, 20000, size=(3, 470000), dtype=np.int16)
    x = pcm.astype(np.float32) / 32768.0
    ref = O.resample_poly(x[:, :441000], 44100, 16000)
    got_f = mm.load_resampled(torch.from_numpy(x).cuda(), 44100, sr=16000, duration=10)
    got_p = mm.load_resampled(torch.from_numpy(pcm).cuda(), 44100, sr=16000, duration=10)
    assert got_f.shape == (3, 160000)
    assert np.abs(got_f.cpu().numpy() - ref).max() <= 2e-6
    assert torch.equal(got_f, got_p)
    one = mm.load_resampled(torch.from_numpy(x[1]), 44100, sr=16000, duration=10)       # CPU in -> CPU out, 1-D
    assert one.device.type == "cpu" and torch.equal(one, got_f[1].cpu())
    full = O.resample_poly(x, 44100, 16000)
    for out_samples in (1000, 160000, full.shape[1], full.shape[1] + 777):
        got = mm.resample_batch(torch.from_numpy(x).cuda(), 44100, 16000, out_samples=out_samples).cpu().numpy()
        want = np.zeros((3, out_samples), np.float32)
        k = min(out_samples, full.shape[1])
        want[:, :k] = full[:, :k]
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6
        assert np.all(got[:, k:] == 0)
    out = torch.full((3, 160000), 7.0, device="cuda")
    ret = mm.resample_batch(torch.from_numpy(x).cuda()[:, :441000], 44100, 16000, out=out)
    assert ret is out and torch.equal(out, got_f)
    with pytest.raises(ValueError):
        mm.resample_batch(torch.from_numpy(x).cuda(), 44100, 16000, out=torch.empty(3, 5, device="cuda"))
    same = mm.resample_batch(torch.from_numpy(pcm[:, :100]).cuda(), 16000, 16000)
    assert torch.equal(same.cpu(), torch.from_numpy(x[:, :100]))


def test_resample_feeds_the_fbank_path(mm, O):
    """44.1 kHz 16-bit track -> resampler -> VGGSound features, all on the device, against the CPU chain
    scipy.resample_poly -> oracle fbank."""
    rng = np.random.default_rng(11)
    pcm = (3000 * rng.standard_normal((2, 441000))).clip(-32768, 32767).astype(np.int16)
    y = mm.load_resampled(torch.from_numpy(pcm).cuda(), 44100, sr=16000, duration=10)
    got = mm.preprocessing_audio_batch(y).cpu().numpy()
    ref_y = O.resample_poly(pcm.astype(np.float32) / 32768.0, 44100, 16000)
    ref = np.stack([O.preprocessing_audio(torch.from_numpy(r)[None]).numpy() for r in ref_y])
    assert got.shape == ref.shape == (2, 384, 128)
    assert np.abs(got - ref).max() <= 5e-4


def test_resample_full_size_properties(mm):
    """512 ten-second 44.1 kHz tracks in one launch: position independence, linearity in a power of two, and the DC gain of
    the low-pass (a constant input comes out as the same constant away from the clip ends)."""
    g = torch.Generator(device="cuda").manual_seed(21)
    base = (0.1 * torch.randn(4, 441000, device="cuda", generator=g)).clamp_(-1, 1)
    small = mm.resample_batch(base, 44100, 16000)
    big = mm.resample_batch(base.repeat(128, 1), 44100, 16000)
    assert big.shape == (512, 160000)
    assert torch.equal(big.view(128, 4, 160000), small.unsqueeze(0).expand(128, -1, -1))
    assert torch.equal(mm.resample_batch(base * 0.25, 44100, 16000), small * 0.25)
    dc = mm.resample_batch(torch.full((1, 441000), 0.5, device="cuda"), 44100, 16000)
    assert (dc[0, 100:-100] - 0.5).abs().max().item() <= 2e-4
    assert torch.isfinite(big).all()


def test_resample_rejects_bad_plans(mm):
    from multi_modal_amd import _native as N
    taps = mm.resample.lowpass_taps(160, 441)
    with pytest.raises(mm.MmfeError, match="coprime"):
        N.Plan(N.make_spec(kind=N.KIND_RESAMPLE, n_fft=320, hop=882, win_len=len(taps)), taps, device=0)
    with pytest.raises(mm.MmfeError, match="odd"):
        N.Plan(N.make_spec(kind=N.KIND_RESAMPLE, n_fft=160, hop=441, win_len=len(taps) - 1), taps[:-1], device=0)
    with pytest.raises(mm.MmfeError, match="mel|log"):
        N.Plan(N.make_spec(kind=N.KIND_RESAMPLE, n_fft=160, hop=441, win_len=len(taps), log_kind=N.LOG_LN_MAX), taps, device=0)
    plan = N.Plan(N.make_spec(kind=N.KIND_RESAMPLE, n_fft=160, hop=441, win_len=len(taps)), taps, device=0)
    assert plan.kernel == "resample"
    plan.close()


def test_resample_both_kernels_agree(mm, O, monkeypatch):
    """44.1 -> 16 kHz runs on the wave-uniform-phase kernel; MMFE_RESAMPLE_FORM=general (read at plan creation) forces the general
    kernel, which otherwise only serves ratios such as 441/160.  Same filter, same alignment: both within 2e-6 of scipy and within
    1e-6 of each other (the accumulation order differs)."""
    x = np.stack([synth("noise0.1", 50001, 4242), synth("uniform", 50001, 4243)])
    ref = O.resample_poly(x, 44100, 16000)
    mm.plans.clear_plans()
    fast = mm.resample_batch(torch.from_numpy(x).cuda(), 44100, 16000).cpu().numpy()
    mm.plans.clear_plans()
    monkeypatch.setenv("MMFE_RESAMPLE_FORM", "general")
    slow = mm.resample_batch(torch.from_numpy(x).cuda(), 44100, 16000).cpu().numpy()
    monkeypatch.delenv("MMFE_RESAMPLE_FORM")
    mm.plans.clear_plans()
    assert np.abs(fast - ref).max() <= 2e-6 and np.abs(slow - ref).max() <= 2e-6
    assert np.abs(fast - slow).max() <= 1e-6


@pytest.mark.parametrize("orig,target", [(16000, 44100), (16000, 22050), (48000, 44100), (44100, 48000), (22050, 16000), (32000, 44100)])
@pytest.mark.parametrize("dtype", ["f32", "s16"])
def test_resample_even_lane_strides_and_round_by_round_tiles(mm, O, orig, target, dtype):
    """Ratios whose lanes sit an even number of samples apart (x/160, x/320: staged with one unused float per lane stride) and
    22.05 -> 16 kHz, whose tile outputs leave round by round because they do not fit LDS beside the staged input: whole tiles,
    a partial last tile, rows shorter than a tile, lengths that are not a multiple of 4, many rows, and data behind the batch."""
    for rows, n in ((3, 120000), (2, 100003), (37, 9001), (1, 5)):
        rng = np.random.default_rng(rows * 7 + n + orig)
        guard = torch.full((rows * n + 64,), 9999, dtype=torch.int16 if dtype == "s16" else torch.float32, device="cuda")
        x = guard[: rows * n].view(rows, n)
        if dtype == "s16":
            pcm = rng.integers(-32768, 32768, size=(rows, n), dtype=np.int16)
            x.copy_(torch.from_numpy(pcm))
            xf = pcm.astype(np.float32) / 32768.0
        else:
            xf = rng.uniform(-1, 1, size=(rows, n)).astype(np.float32)
            x.copy_(torch.from_numpy(xf))
        ref = O.resample_poly(xf, orig, target)
        got = mm.resample_batch(x, orig, target).cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 4e-6, (rows, n, mm.plans.last_kernel())


@pytest.mark.parametrize("orig,target", [(16000, 44100), (44100, 16000), (8000, 16000), (48000, 16000), (16000, 48000)])
def test_resample_padded_staging_is_bit_identical(mm, monkeypatch, orig, target):
    """MMFE_RESAMPLE_PAD (read at plan creation) forces the padded LDS layout on or off.  The unused floats carry weight zero, so
    both layouts add the same products in the same order: the outputs must be the same bits - for lane strides of 1 and 2 samples
    too, where a run's window crosses an unused float every one or two samples."""
    g = torch.Generator(device="cuda").manual_seed(orig + target)
    x = torch.rand(5, 70001, device="cuda", generator=g) * 2 - 1
    outs = []
    for pad in ("0", "1"):
        mm.plans.clear_plans()
        monkeypatch.setenv("MMFE_RESAMPLE_PAD", pad)
        outs.append(mm.resample_batch(x, orig, target))
        outs.append(mm.resample_batch(x[:, :70000].contiguous(), orig, target))
    monkeypatch.delenv("MMFE_RESAMPLE_PAD")
    mm.plans.clear_plans()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])


@pytest.mark.parametrize("rows,n", [(3, 441000), (5, 100004), (2, 30000), (70, 61740), (1, 8), (2, 50002), (3, 50001)])
def test_resample_pcm_two_block_form_is_bit_identical(mm, O, monkeypatch, rows, n):
    """16-bit rows of aligned batches take the form that keeps the samples 16-bit in LDS and runs two blocks per CU;
    MMFE_RESAMPLE_FORM=wide (read at plan creation) keeps them on the one-block form.  The 1 / 32768 of the decode moves from the
    samples to the weights, a power-of-two scale: the outputs must be the same bits.  Shapes: whole tiles plus a partial last tile,
    rows shorter than one tile, many rows per block, a row of 8 samples, row lengths that are not a multiple of 4 (both settings
    then run the sample-by-sample staging); full-scale samples included; the batch is followed by
    other data (a read past its end would show)."""
    rng = np.random.default_rng(rows * 1000 + n)
    pcm = rng.integers(-32768, 32768, size=(rows, n), dtype=np.int16)
    pcm[0, :4] = [32767, -32768, 32767, -32768]
    guard = torch.full((rows * n + 64,), 12345, dtype=torch.int16, device="cuda")
    x = guard[: rows * n].view(rows, n)
    x.copy_(torch.from_numpy(pcm))
    mm.plans.clear_plans()
    two = mm.resample_batch(x, 44100, 16000)
    mm.plans.clear_plans()
    monkeypatch.setenv("MMFE_RESAMPLE_FORM", "wide")
    one = mm.resample_batch(x, 44100, 16000)
    monkeypatch.delenv("MMFE_RESAMPLE_FORM")
    mm.plans.clear_plans()
    assert torch.equal(two, one)
    ref = O.resample_poly(pcm.astype(np.float32) / 32768.0, 44100, 16000)
    assert two.shape == ref.shape and np.abs(two.cpu().numpy() - ref).max() <= 4e-6


def test_device_prefetcher_overlapped_batches(mm, O):
    """DataLoader-shaped batches (waveforms + labels) through the pinned-buffer prefetcher with the fused transform: every batch
    equals the direct call on the same clips, in order, for several depths, batch counts around the pipeline depth, a ragged last
    batch, int16 PCM clips, and dict batches; tensors already on the device pass through."""
    from multi_modal_amd.loader import DevicePrefetcher
    rng = np.random.default_rng(5)
    clips = [torch.from_numpy((0.1 * rng.standard_normal((3 if i == 6 else 4, 1, 16000))).astype(np.float32)) for i in range(7)]
    labels = [torch.arange(c.shape[0]) + 10 * i for i, c in enumerate(clips)]
    want = [mm.preprocessing_audio_batch(c.cuda()) for c in clips]
    for depth in (2, 3, 4):
        for n_batches in (0, 1, 2, 3, 7):
            got = list(DevicePrefetcher(zip(clips[:n_batches], labels[:n_batches]), depth=depth,
                                        transform=lambda b: (mm.preprocessing_audio_batch(b[0]), b[1])))
            assert len(got) == n_batches
            for (f, l), w, lab in zip(got, want, labels):
                assert f.is_cuda and l.is_cuda and torch.equal(f, w) and torch.equal(l.cpu(), lab)
    pcm = [(c * 32767.0).round().to(torch.int16) for c in clips[:3]]
    got = list(DevicePrefetcher(({"audio": p, "name": "clip%d" % i, "on_device": labels[i].cuda()} for i, p in enumerate(pcm)),
                                transform=lambda b: {**b, "audio": mm.preprocessing_audio_batch(b["audio"])}))
    for i, g in enumerate(got):
        assert g["name"] == "clip%d" % i and torch.equal(g["on_device"].cpu(), labels[i])
        assert torch.equal(g["audio"], mm.preprocessing_audio_batch(pcm[i].cuda()))
    with pytest.raises(ValueError):
        DevicePrefetcher([], depth=1)


def test_clamp_plan_refuses_more_rows_than_its_scratch_holds(mm):
    """A clamp_below_max plan keeps per-row extremes for at most 2^20 rows per call: one row more is refused with a clear error
    before anything is launched (the output keeps its fill), and the same plan still serves a smaller batch afterwards."""
    rows = (1 << 20) + 1
    x = torch.zeros(rows, 160, device="cuda")
    out = torch.full((rows, 80, 1), 7.0, device="cuda")
    with pytest.raises(mm.MmfeError, match="rows"):
        mm.logmel.log_mel_spectrogram(x, out=out)
    torch.cuda.synchronize()
    assert torch.all(out == 7.0)
    ok = mm.logmel.log_mel_spectrogram(x[:5])
    assert ok.shape == (5, 80, 1) and torch.isfinite(ok).all()


def test_frame_signal_typed_cases_golden_reference(mm, golden_dir):
    """float64 / float16 / bfloat16 / mixed-dtype operands against stored outputs of the reference's own frame_signal: the dtype
    torch's promotion gives the product, and every bit of it (CPU and GPU operands)."""
    from test_oracle_cpu import _typed_frame_cases
    for s, w, fl, hop, want in _typed_frame_cases(np.load(f"{golden_dir}/frame_signal.npz")):
        for dev in ("cpu", "cuda"):
            got = mm.frame_signal(s.to(dev), fl, hop) if w is None else mm.frame_signal(s.to(dev), fl, hop, w.to(dev))
            assert got.dtype == want.dtype and got.shape == want.shape and got.device.type == dev
            assert torch.equal(got.cpu().contiguous().view(torch.uint8), want.contiguous().view(torch.uint8)), (s.dtype, dev)
    # integer signal with the default (integer ones) window: integer frames, exact
    xi = torch.arange(-20, 21, dtype=torch.int16)
    gi = mm.frame_signal(xi, 8, 3)
    assert gi.dtype == torch.int16 and torch.equal(gi, O_frame(xi, 8, 3))
    # float64 values that float32 cannot hold survive
    xd = torch.tensor([1.0 + 2.0 ** -40, -3.0 - 2.0 ** -45, 2.0 ** -60, 7.0], dtype=torch.float64)
    assert torch.equal(mm.frame_signal(xd.cuda(), 4, 2).cpu()[0], xd)


def O_frame(s, fl, hop, w=None):
    from oracle import frontend_oracle
    return frontend_oracle.frame_signal(s, fl, hop, w)


def test_frame_signal_window_cache_is_bounded_and_identity_keyed(mm, O):
    """At most transforms._MAX_WINDOWS frame plans stay alive however many windows pass through; a window tensor seen before (same
    object, unmodified) is recognised without reading it back; modifying it in place is noticed."""
    from multi_modal_amd import plans, transforms
    x = torch.from_numpy(synth("noise0.1", 1000, 41)).cuda()
    for i in range(transforms._MAX_WINDOWS + 5):
        w = torch.full((32,), float(i + 1))
        assert torch.equal(mm.frame_signal(x, 32, 8, w.cuda()).cpu(), O.frame_signal(x.cpu(), 32, 8, w))
    assert plans.family_size("frame") == transforms._MAX_WINDOWS
    w = torch.hann_window(32).cuda()
    a = mm.frame_signal(x, 32, 8, w)
    assert transforms._window_key(w, torch.float32)[1] is None          # known by identity: no host copy was made
    w.mul_(2.0)
    b = mm.frame_signal(x, 32, 8, w)
    assert torch.equal(b, 2.0 * a)


def test_frame_signal_plans_are_cached_by_window(mm, O):
    """Repeated frame_signal calls with equal windows share one plan (no table rebuild, no stream synchronisation); another
    window gets another plan and its own values."""
    from multi_modal_amd import plans
    plans.clear_plans()              # (the family of frame plans is bounded: start from an empty one)
    x = torch.from_numpy(synth("noise0.1", 4000, 31)).cuda()
    w1 = torch.hann_window(64)
    w2 = torch.hamming_window(64)
    a = mm.frame_signal(x, 64, 16, w1.cuda())
    n_plans = len(plans._plans)
    b = mm.frame_signal(x, 64, 16, w1.clone().cuda())
    assert len(plans._plans) == n_plans and torch.equal(a, b)
    c = mm.frame_signal(x, 64, 16, w2.cuda())
    assert len(plans._plans) == n_plans + 1
    assert torch.equal(a.cpu(), O.frame_signal(x.cpu(), 64, 16, w1)) and torch.equal(c.cpu(), O.frame_signal(x.cpu(), 64, 16, w2))


def test_vggsound_against_the_c_restatement(mm):
    """The HIP path against the second, independent CPU restatement (oracle/fbank_ref.c: plain C, its own radix-2 FFT): same bar as
    against the Python oracle."""
    from oracle import fbank_ref
    clips = np.stack([synth(k, 160000, 4321 + i) for i, k in enumerate(("noise0.1", "uniform", "quiet", "impulse"))])
    ref = fbank_ref.preprocessing_audio_batch(clips, threads=4)
    got = mm.preprocessing_audio_batch(torch.from_numpy(clips).cuda()).cpu().numpy()
    mx, p999 = _stats(got - ref)
    print("vs C restatement: max", mx, "p99.9", p999)
    assert p999 <= 1e-4 and mx <= 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_logmel_half_precision_output(mm, dtype):
    """Long-utterance log-mel stored as float16 / bfloat16 = the rounding of the float32 features, bit for bit, with the per-row
    clamp active (a loud burst in a quiet utterance), lengths that end in a partial octet, other hops, and PCM input."""
    rng = np.random.default_rng(12)
    x = (1e-5 * rng.standard_normal((5, 48000))).astype(np.float32)
    x[:, 10000:12000] += 0.5 * rng.standard_normal((5, 2000)).astype(np.float32)
    x = torch.from_numpy(x).cuda()
    for n, hop in ((48000, 160), (47999, 160), (12345, 160), (16000, 128), (401, 37)):
        ref = mm.logmel.log_mel_spectrogram(x[:, :n], hop_length=hop)
        got = mm.logmel.log_mel_spectrogram(x[:, :n], hop_length=hop, dtype=dtype)
        assert mm.plans.last_kernel() == "rfft400"
        assert got.dtype == dtype and got.shape == ref.shape
        assert torch.equal(got, ref.to(dtype)), (n, hop)
    ref = mm.logmel.log_mel_spectrogram(x)
    assert (ref.amin(dim=(1, 2)) >= ref.amax(dim=(1, 2)) - 2.0 - 1e-6).all() and (ref == ref.amin(dim=(1, 2), keepdim=True)).float().mean() > 0.2
    pcm = (x * 32767.0).round().clamp(-32768, 32767).to(torch.int16)
    assert torch.equal(mm.logmel.log_mel_spectrogram(pcm, dtype=dtype), mm.logmel.log_mel_spectrogram(pcm).to(dtype))
    with pytest.raises(mm.MmfeError, match="16-bit output"):
        mm.logmel.log_mel_spectrogram(x, precision="float64", dtype=dtype)       # any-length kernel with a clamp pass: float32 only
