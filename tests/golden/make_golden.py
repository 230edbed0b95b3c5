#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run HERE, in the build container; the files travel,
/root/reference and third-party sources do not).

What pins what
  har_stft.npz        outputs of scipy.signal.stft itself, the third-party function the reference calls at
                      classification/har/adbox.py:20 with the literal arguments used there  -> PINNED
  frame_signal.npz    outputs of the reference's own frame_signal
                      (classification/vggsound/utils/transforms.py:19-61), executed from the reference file:
                      the function definition alone is compiled from the file's AST, because importing the
                      whole module needs torchvision, which this image does not have  -> PINNED
  kaldi_fbank.npz     outputs of the oracle's restatement (torchaudio is absent: PARITY UNPINNED) plus the
                      same cells computed by an independent float64 implementation that is present in the
                      image (transformers.audio_utils.spectrogram with kaldi mel scale, the recipe
                      transformers uses as its torchaudio-free fallback) as a cross-check
  epic_logspec.npz    outputs of the oracle's restatement (librosa is absent: PARITY UNPINNED) plus a plain
                      numpy.fft.rfft evaluation on explicitly framed data as a cross-check
  logmel.npz          long-utterance log-mel (not in the reference): oracle restatement vs
                      transformers.WhisperFeatureExtractor
  epic_slices.npz     (left, right) sample ranges chosen by the reference's own EPICKitchen._extract_sound_feature
                      (classification/recognition/dataset/epic_kitchen.py:128-154), executed from the reference file:
                      the method definition alone is compiled from the file's AST (the module import needs librosa,
                      torchvision, h5py and a pickle) and called on a stub `self` whose audio dictionary holds
                      numpy.arange recordings and whose _log_specgram returns its argument, so the returned slice
                      names its own bounds  -> PINNED (integer index work, bit exact)
  spec_transform.npz  outputs of the reference's own Stack -> ToTorchFormatTensor(div=True) -> GroupNormalize([0],[1])
                      (classification/recognition/dataset/transform.py:253-285, 63-76; wired at epic_kitchen.py:62-66),
                      class definitions compiled from the file's AST, fed the mode-'F' PIL image epic_kitchen.py:176
                      builds  -> PINNED
  patch_embed.npz     tokens computed by the reference's own PatchEmbed class (classification/vggsound/models/vit_model.py:179-199),
                      class definition compiled from the file's AST (the module import needs timm), built the way
                      AudioTransformer builds its audio projection (vit_model.py:435: Conv2d(1, E, (16, 16), stride 16)) with
                      seeded weights, run in float32 on seeded (B, 1, H, W) images  -> PINNED (patch order, weight layout,
                      flatten / transpose; the 16-bit rounding of autocast is restated by the oracle on top of it)
Inputs are regenerated from tests/conftest.synth (integer PRNG), so no waveform is stored.
"""
import ast
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import synth  # noqa: E402
from oracle import frontend_oracle as O  # noqa: E402

REF = "/root/reference"
KINDS = ["noise0.1", "uniform", "sine+noise", "silence", "impulse", "dc", "quiet"]
FRAMES = [0, 1, 191, 383]          # first / second / middle / last kept frame of the 384


def har():
    import scipy.signal
    out = {}
    for name, shape, seed in [("b6x192", (4, 6, 192), 4321), ("c15x1003", (15, 1003), 7), ("exact32", (2, 32), 9), ("t33", (3, 33), 11)]:
        rng = np.random.default_rng(seed)
        x = (rng.integers(-(1 << 20), 1 << 20, size=shape).astype(np.float64) / float(1 << 18)).astype(np.float32)
        y = np.abs(scipy.signal.stft(x, fs=100, nperseg=32, noverlap=28)[-1])
        out[name + "_x"] = x
        out[name + "_y"] = y.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "har_stft.npz"), **out)
    print("har_stft.npz", {k: v.shape for k, v in out.items()})


def frame_signal():
    # Only the `frame_signal` function definition of the reference file is compiled and run (its module import needs the
    # absent torchvision).  That definition was read before it was run here: it is index arithmetic and torch tensor
    # operations (torch.zeros / torch.cat / arange / indexing / multiply), with no imports, I/O or calls beyond torch and math.
    src = open(os.path.join(REF, "classification/vggsound/utils/transforms.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "frame_signal"][0]
    ns = {"torch": torch, "math": math}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference:frame_signal", "exec"), ns)
    ref_fn = ns["frame_signal"]
    out = {}
    cases = [((1, 10), 4, 3), ((2, 1600), 400, 160), ((3,), 8, 2), ((2, 3, 100), 16, 7), ((5,), 5, 2), ((7,), 5, 3), ((1, 33), 32, 4)]
    for i, (shape, fl, hop) in enumerate(cases):
        rng = np.random.default_rng(100 + i)
        s = (rng.integers(-1000, 1000, size=shape).astype(np.float32) / 256.0)
        w = (rng.integers(-1000, 1000, size=fl).astype(np.float32) / 512.0)
        y = ref_fn(torch.from_numpy(s), fl, hop, torch.from_numpy(w)).numpy()
        y1 = ref_fn(torch.from_numpy(s), fl, hop).numpy()
        out[f"c{i}_signal"], out[f"c{i}_window"] = s, w
        out[f"c{i}_params"] = np.array([fl, hop])
        out[f"c{i}_out"], out[f"c{i}_out_ones"] = y, y1
    # the product's dtype follows torch's promotion of signal * window (transforms.py:24-25, 58-59): float64, float16, bfloat16 and
    # mixed operands, stored as the bit patterns the reference function returned (16-bit types as uint16 views)
    dt = {"f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
    typed = [("f64", "f64", (2, 50), 16, 5), ("f16", "f16", (3, 41), 8, 3), ("bf16", "bf16", (2, 37), 8, 4), ("f32", "f64", (1, 45), 12, 7),
             ("f16", "f32", (2, 29), 6, 5), ("f64", None, (1, 23), 9, 2)]
    for i, (sd, wd, shape, fl, hop) in enumerate(typed):
        rng = np.random.default_rng(300 + i)
        s = torch.from_numpy(rng.standard_normal(shape)).to(dt[sd])               # values with full mantissas in the signal's type
        w = None if wd is None else torch.from_numpy(rng.standard_normal(fl)).to(dt[wd])
        y = ref_fn(s, fl, hop) if w is None else ref_fn(s, fl, hop, w)
        bits = lambda t: t.view(torch.int16).numpy().view(np.uint16) if t.dtype in (torch.float16, torch.bfloat16) else t.numpy()
        out[f"t{i}_signal"], out[f"t{i}_out"] = bits(s), bits(y)
        if w is not None:
            out[f"t{i}_window"] = bits(w)
        out[f"t{i}_params"] = np.array([fl, hop])
        out[f"t{i}_dtypes"] = np.array([sd, wd or "none", {v: k for k, v in dt.items()}[y.dtype]])
    np.savez_compressed(os.path.join(HERE, "frame_signal.npz"), **out)
    print("frame_signal.npz", len(cases), "float32 cases,", len(typed), "typed cases")


def kaldi():
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    import warnings
    warnings.simplefilter("ignore")
    mf = mel_filter_bank(257, 128, 20, 8000, 16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    win = window_function(400, "hann", periodic=False)
    out = {"frames": np.array(FRAMES)}
    for i, kind in enumerate(KINDS):
        x = synth(kind, 160000, 1234 + i)
        raw = O.kaldi_fbank(torch.from_numpy(x)[None], sample_frequency=16000, window_type="hanning", num_mel_bins=128, frame_shift=25).numpy()
        norm = O.preprocessing_audio(torch.from_numpy(x)[None]).numpy()
        hf = spectrogram(x.astype(np.float64), win, frame_length=400, hop_length=400, fft_length=512, power=2.0, center=False,
                         preemphasis=0.97, mel_filters=mf, log_mel="log", mel_floor=1.192092955078125e-07,
                         remove_dc_offset=True, dtype=np.float64).T
        assert raw.shape == (400, 128) and norm.shape == (384, 128) and hf.shape == (400, 128)
        out[f"{kind}_raw"] = raw[FRAMES].astype(np.float32)
        out[f"{kind}_norm"] = norm[FRAMES].astype(np.float32)
        out[f"{kind}_hf64"] = hf[FRAMES].astype(np.float64)
        out[f"{kind}_sums"] = np.array([raw.astype(np.float64).sum(), (raw.astype(np.float64) ** 2).sum(),
                                        norm.astype(np.float64).sum(), (norm.astype(np.float64) ** 2).sum()])
    bank = torch.nn.functional.pad(O.kaldi_mel_banks(128, 512, 16000.0), (0, 1)).numpy()
    out["bank_nnz_per_row"] = (bank > 0).sum(1).astype(np.int32)
    out["bank_first_rows"] = bank[:8, :8].astype(np.float32)
    out["bank_checksum"] = np.array([bank.astype(np.float64).sum(), (bank.astype(np.float64) ** 2).sum()])
    np.savez_compressed(os.path.join(HERE, "kaldi_fbank.npz"), **out)
    print("kaldi_fbank.npz", len(out), "arrays")


def epic():
    out = {}
    rows = [0, 1, 17, 128, 255]
    for n in (30672, 30696):
        for kind in ("noise0.1", "quiet", "silence", "sine+noise"):
            x = synth(kind, n, 4242 + n)
            y = O.log_specgram(x)
            assert y.shape == (256, 256) and y.dtype == np.float32
            # independent evaluation: explicit framing + numpy rfft of length 511, float64 throughout
            w = np.zeros(511)
            w[135:375] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(240) / 240)
            xp = np.pad(x.astype(np.float64), 255)
            fr = np.stack([xp[120 * t:120 * t + 511] * w for t in range(256)])
            z = np.fft.rfft(fr, n=511, axis=1).T
            chk = np.log(np.abs(z) ** 2 + 1e-6)
            out[f"{kind}_{n}_rows"] = y[rows].astype(np.float32)
            out[f"{kind}_{n}_cols"] = y[:, [0, 1, 128, 255]].astype(np.float32)
            out[f"{kind}_{n}_np64_rows"] = chk[rows]
            out[f"{kind}_{n}_sums"] = np.array([y.astype(np.float64).sum(), (y.astype(np.float64) ** 2).sum()])
    out["rows"] = np.array(rows)
    np.savez_compressed(os.path.join(HERE, "epic_logspec.npz"), **out)
    print("epic_logspec.npz", len(out), "arrays")


def _reference_defs(rel_path, names, ns):
    """Compiles the named top-level functions / classes, or `Class.method` definitions, of one reference file from its AST
    into `ns` (nothing else of the file is executed) and returns them in order."""
    tree = ast.parse(open(os.path.join(REF, rel_path)).read())
    out = []
    for name in names:
        cls, _, meth = name.partition(".")
        node = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == cls][0]
        if meth:
            node = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == meth][0]
        exec(compile(ast.Module(body=[node], type_ignores=[]), "reference:" + name, "exec"), ns)
        out.append(ns[node.name])
    return out


EPIC_SLICE_SR = 24000


def epic_slice_cases():
    """(start_frame, idx, fps, n_total) rows: both edge clamps, the first / last interior positions, .5-rounding ties of
    left_sec * 24000 (fps 50 and 60 put centres on multiples of 1/50, 1/60 s: 0.639 * 24000 = 15336 exactly, so a tie needs
    centre * 24000 = k + 0.5, i.e. odd multiples of 1/48000 s - fps 50 / 60 never give one, fps 29.97 and 23.976 do not
    either; ties are forced with synthetic fps values), recordings shorter than one edge slice, fps 50 / 60."""
    rows = []
    for fps in (50.0, 60.0):
        for n_total in (24000 * 60, 24000 * 7 + 123, 30696, 30697, 40000, 20000, 1):
            dur_frames = int(n_total / EPIC_SLICE_SR * fps)
            for centre in sorted({0, 1, 2, int(0.639 * fps), int(0.639 * fps) + 1, int(0.64 * fps) + 1, dur_frames // 2, dur_frames // 3,
                                  dur_frames - int(0.639 * fps) - 1, dur_frames - int(0.639 * fps), dur_frames - 1, dur_frames, dur_frames + 5}):
                if centre < 0:
                    continue
                rows.append((centre // 2, centre - centre // 2, fps, n_total))
    # forced ties: centre_sec * 24000 = k + 0.5  <=>  (start + idx) / fps = (2k + 1) / 48000
    for k in (20000, 20001, 33333, 100000, 100001):
        rows.append((2 * k + 1, 0, 48000.0, 24000 * 20))
        rows.append((k, k + 1, 48000.0, 24000 * 20))
    # non-integer fps as video_records may hold
    for fps in (29.97, 23.976, 59.94):
        for centre in (0, 19, 20, 500, 1000, 1797, 1798):
            rows.append((centre, 3, fps, 24000 * 61))
    return rows


def epic_slices():
    import types
    (ref_fn,) = _reference_defs("classification/recognition/dataset/epic_kitchen.py", ["EPICKitchen._extract_sound_feature"], {"np": np})
    (ret_fn,) = _reference_defs("classification/retrieval/dataset/epic_kitchen.py", ["EPICKitchen._extract_sound_feature"], {"np": np})
    rows = epic_slice_cases()
    res = []
    for fn in (ref_fn, ret_fn):
        got = []
        for start_frame, idx, fps, n_total in rows:
            stub = types.SimpleNamespace(use_audio_dict=True, audio_path={"rec": np.arange(n_total, dtype=np.int64)},
                                         resampling_rate=EPIC_SLICE_SR, _log_specgram=lambda s: s)
            record = types.SimpleNamespace(start_frame=start_frame, fps={"Spec": fps}, untrimmed_video_name="rec")
            sl = fn(stub, record, idx)
            got.append((int(sl[0]), int(sl[-1]) + 1) if len(sl) else (0, 0))
        res.append(got)
    assert res[0] == res[1], "recognition/ and retrieval/ copies disagree"
    cases = np.array([(a, b, c, d) for a, b, c, d in rows], dtype=np.float64)
    bounds = np.array(res[0], dtype=np.int64)
    lens = bounds[:, 1] - bounds[:, 0]
    print("epic_slices.npz", len(rows), "cases; slice lengths", sorted(set(lens.tolist())))
    np.savez_compressed(os.path.join(HERE, "epic_slices.npz"), cases=cases, bounds=bounds)


def spec_transform():
    from PIL import Image
    ns = {"np": np, "torch": torch}
    Stack, ToTorchFormatTensor, GroupNormalize = _reference_defs(
        "classification/recognition/dataset/transform.py", ["Stack", "ToTorchFormatTensor", "GroupNormalize"], ns)
    chain = [Stack(), ToTorchFormatTensor(), GroupNormalize([0], [1])]       # epic_kitchen.py:62-66 with input_mean / input_std of 'Spec' (:37)
    out = {}
    for i, (shape, seed) in enumerate([((256, 256), 31), ((17, 9), 32), ((1, 1), 33)]):
        rng = np.random.default_rng(seed)
        spec = (rng.integers(-(1 << 22), 1 << 22, size=shape).astype(np.float64) / float(1 << 18)).astype(np.float32)   # ln-power range -16 .. 16
        x = [Image.fromarray(spec)]                                          # epic_kitchen.py:176
        assert x[0].mode == "F"
        for t in chain:
            x = t(x)
        assert tuple(x.shape) == (1,) + shape and x.dtype == torch.float32
        out[f"c{i}_seed_shape"] = np.array([seed, shape[0], shape[1]])
        out[f"c{i}_out"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, "spec_transform.npz"), **out)
    print("spec_transform.npz", {k: v.shape for k, v in out.items()})


def logmel():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    x = synth("noise0.1", 480000, 99)
    ref = fe._np_extract_fbank_features(x[None], "cpu")[0]
    got = O.whisper_logmel(x)
    assert ref.shape == got.shape == (80, 3000)
    np.savez_compressed(os.path.join(HERE, "logmel.npz"), cols=np.array([0, 1, 1500, 2999]), hf=ref[:, [0, 1, 1500, 2999]].astype(np.float32),
                        oracle=got[:, [0, 1, 1500, 2999]], sums=np.array([got.astype(np.float64).sum(), (got.astype(np.float64) ** 2).sum()]))
    print("logmel.npz max|oracle-hf| =", float(np.abs(ref - got).max()))


def resample():
    """Outputs of scipy.signal.resample_poly itself (float32 input): the function the resampler restates."""
    import scipy.signal
    out = {}
    for tag, n, up, down in (("44k1_to_16k", 44100, 160, 441), ("48k_to_16k", 4801, 1, 3), ("48k_to_24k", 5000, 1, 2),
                             ("16k_to_44k1", 1600, 441, 160), ("short", 7, 160, 441)):
        x = synth("noise0.1", n, 700 + n)
        out[tag + "_y"] = scipy.signal.resample_poly(x, up, down).astype(np.float32)
        out[tag + "_params"] = np.array([n, 700 + n, up, down])
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)
    print("resample.npz", len(out), "arrays")


PATCH_CASES = [  # name, B, H, W, E, seed, token stride of the stored subset
    ("vgg", 2, 384, 128, 768, 501, 13),        # (B, 1, 384, 128): audio.unsqueeze(1) of the VGGSound tensor, 192 tokens
    ("epic", 1, 256, 256, 768, 502, 17),       # audio.view(-1, 1, 256, 256), 256 tokens
    ("small", 3, 32, 48, 384, 503, 1),
    ("ragged", 2, 40, 52, 384, 504, 1),        # rows / columns beyond the last whole patch are ignored by the convolution
]


def patch_embed_inputs(B, H, W, E, seed):
    """Dyadic values with few significant bits: exact in float16 and bfloat16, and every partial sum of the 256 products is exact
    in float32, so the float32 convolution has ONE right answer whatever the summation order or operand rounding."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-12, 13, size=(B, 1, H, W)).astype(np.float32) / 4.0
    w = rng.integers(-8, 9, size=(E, 1, 16, 16)).astype(np.float32) / 16.0
    b = rng.integers(-64, 65, size=(E,)).astype(np.float32) / 8.0
    return x, w, b


def patch_embed():
    from collections import abc
    from itertools import repeat
    ns = {"torch": torch, "nn": torch.nn, "abc": abc, "repeat": repeat}
    # Read before it was run here: to_2tuple is a two-line tuple helper, PatchEmbed an nn.Module holding one Conv2d whose forward is
    # `self.proj(x).flatten(2).transpose(1, 2)`; no imports, I/O or calls beyond torch.nn.
    _, PatchEmbed = _reference_defs("classification/vggsound/models/vit_model.py", ["to_2tuple", "PatchEmbed"], ns)
    out = {}
    for name, B, H, W, E, seed, step in PATCH_CASES:
        x, w, b = patch_embed_inputs(B, H, W, E, seed)
        pe = PatchEmbed(img_size=224, patch_size=16, in_chans=3, embed_dim=E)          # as VisionTransformer.__init__ builds it
        proj = torch.nn.Conv2d(1, E, kernel_size=(16, 16), stride=(16, 16))            # vit_model.py:435, fstride = tstride = 16
        with torch.no_grad():
            proj.weight.copy_(torch.from_numpy(w))
            proj.bias.copy_(torch.from_numpy(b))
            pe.proj = proj                                                             # vit_model.py:439
            y = pe(torch.from_numpy(x)).numpy()
        assert y.shape == (B, (H // 16) * (W // 16), E)
        out[name + "_shape"] = np.array([B, H, W, E, seed, step])
        out[name + "_tokens"] = y[:, ::step].copy()
        out[name + "_sums"] = np.array([y.astype(np.float64).sum(), (y.astype(np.float64) ** 2).sum()])
    np.savez_compressed(os.path.join(HERE, "patch_embed.npz"), **out)
    print("patch_embed.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
        raise SystemExit(0)
    har()
    frame_signal()
    kaldi()
    epic()
    logmel()
    resample()
    epic_slices()
    spec_transform()
    patch_embed()
