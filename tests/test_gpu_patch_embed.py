"""Patch embedding (csrc/patch_embed.hip through the C ABI, mmfe_patch_embed_*) against
  * tokens the reference's own PatchEmbed class computed (tests/golden/patch_embed.npz, made by make_golden.py from
    classification/vggsound/models/vit_model.py:179-199 with the projection of vit_model.py:435) on dyadic data whose float32
    convolution has one right answer: bit exact;
  * the oracle's restatement of the autocast evaluation (operands rounded to the 16-bit type, exact sums) on random data:
    float32 tokens within 2e-5 * sum |x w| (float32 accumulation of 256 products), 16-bit tokens within one unit in the last
    place of the 16-bit type.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["vgg", "epic", "small", "ragged"]


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


def dyadic(B, H, W, E, seed):
    """The generator of tests/golden/make_golden.py: values exact in float16 and bfloat16, every partial sum exact in float32."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-12, 13, size=(B, 1, H, W)).astype(np.float32) / 4.0
    w = rng.integers(-8, 9, size=(E, 1, 16, 16)).astype(np.float32) / 16.0
    b = rng.integers(-64, 65, size=(E,)).astype(np.float32) / 8.0
    return torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b)


@pytest.mark.parametrize("compute", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_patch_embed_golden_reference_class(mm, golden_dir, case, compute):
    from multi_modal_amd.patch_embed import PatchEmbed
    z = np.load(os.path.join(golden_dir, "patch_embed.npz"))
    B, H, W, E, seed, step = (int(v) for v in z[case + "_shape"])
    x, w, b = dyadic(B, H, W, E, seed)
    pe = PatchEmbed((H, W), w, b, compute_dtype=compute)
    y = pe(x.cuda()).cpu().numpy()
    assert y.shape == (B, (H // 16) * (W // 16), E)
    assert np.array_equal(y[:, ::step], z[case + "_tokens"])
    sums = np.array([y.astype(np.float64).sum(), (y.astype(np.float64) ** 2).sum()])
    assert np.array_equal(sums, z[case + "_sums"])
    y16 = PatchEmbed((H, W), w, b, compute_dtype=compute, out_dtype=compute)(x.cuda())
    assert y16.dtype == compute and torch.equal(y16.cpu(), torch.from_numpy(y).to(compute))      # exact sums, rounded once


def test_patch_embed_many_steps_per_block_bit_exact(mm, O):
    """300 VGGSound-shaped images = 900 steps of 64 patches over 256 persistent blocks: every block runs its double-buffered
    loop three or four times.  Dyadic data, so a tile read before its DMA has landed (or after the next one has) shows as a wrong
    bit somewhere; the float32 convolution is the checker."""
    from multi_modal_amd.patch_embed import PatchEmbed
    B, H, W, E = 300, 384, 128, 768
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.integers(-12, 13, size=(B, H, W)).astype(np.float32) / 4.0)
    _, w, b = dyadic(1, 16, 16, E, 78)
    ref = O.patch_embed(x, w, b, compute_dtype=None)
    pe = PatchEmbed((H, W), w, b, compute_dtype=torch.bfloat16)
    xd = x.cuda()
    for _ in range(3):
        y = pe(xd)
        assert torch.equal(y.cpu(), ref)
    y16 = PatchEmbed((H, W), w, b, compute_dtype=torch.float16, out_dtype=torch.float16)(xd)
    assert torch.equal(y16.cpu(), ref.to(torch.float16))


@pytest.mark.parametrize("E", [384, 768])
@pytest.mark.parametrize("compute", [torch.float16, torch.bfloat16])
def test_patch_embed_random_against_oracle(mm, O, compute, E):
    from multi_modal_amd.patch_embed import PatchEmbed
    g = torch.Generator().manual_seed(5 + E)
    B, H, W = 5, 128, 384                                           # the (B, 1, 128, 384) image layout: 24 patches per row
    x = torch.randn(B, 1, H, W, generator=g) * 1.5 - 0.3
    w = torch.randn(E, 1, 16, 16, generator=g) * 0.05
    b = torch.randn(E, generator=g) * 0.1
    ref = O.patch_embed(x, w, b, compute_dtype=compute, out_dtype=torch.float64)
    scale = O.patch_embed(x.abs(), w.abs(), None, compute_dtype=compute, out_dtype=torch.float64)      # sum |x w| per token
    y = PatchEmbed((H, W), w, b, compute_dtype=compute)(x.cuda()).cpu().double()
    assert ((y - ref).abs() <= 2e-5 * scale + 1e-6).all()
    for out_dtype in (torch.float16, torch.bfloat16):
        y16 = PatchEmbed((H, W), w, b, compute_dtype=compute, out_dtype=out_dtype)(x.cuda()).cpu()
        ulp = torch.finfo(out_dtype).eps * ref.abs().clamp_min(2.0 ** -14)
        assert ((y16.double() - ref).abs() <= ulp + 2e-5 * scale).all()
    # and against the plain float32 module (no autocast): the 16-bit rounding of the operands is all that separates them
    plain = O.patch_embed(x, w, b, compute_dtype=None).double()
    assert ((y - plain).abs() <= 2.5 * torch.finfo(compute).eps * scale + 1e-4).all()


@pytest.mark.parametrize("compute", [torch.float16, torch.bfloat16])
def test_patch_embed_16_bit_images(mm, O, compute):
    """An image that already holds the products' 16-bit type (what the front-end kernels store with dtype=float16 / bfloat16) gives
    the tokens of its float32 form bit for bit - the rounding on the way in is the identity on such values - over 900 steps
    (pairs of patches per DMA, interleaved pieces), on a batch that ends in a partial step, and from the front-end's own output."""
    from multi_modal_amd.patch_embed import PatchEmbed
    g = torch.Generator().manual_seed(21)
    w = torch.randn(768, 1, 16, 16, generator=g) * 0.03
    b = torch.randn(768, generator=g) * 0.1
    for B, H, W, out_dtype in ((300, 384, 128, torch.float32), (5, 48, 40, compute), (3, 128, 384, torch.float16)):
        x16 = (torch.randn(B, 1, H, W, generator=g) * 1.5 - 0.3).to(compute).cuda()
        pe = PatchEmbed((H, W), w, b, compute_dtype=compute, out_dtype=out_dtype)
        y16 = pe(x16)
        y32 = pe(x16.float())
        assert y16.shape == (B, (H // 16) * (W // 16), 768) and torch.equal(y16, y32)
    ref = O.patch_embed(x16.cpu().float(), w, b, compute_dtype=compute, out_dtype=torch.float64)
    assert ((y16.cpu().double() - ref).abs() <= 2e-3 * ref.abs() + 1e-3).all()
    # dyadic data: exact whatever the path
    xd, wd, bd = dyadic(70, 64, 48, 384, 33)
    refd = O.patch_embed(xd, wd, bd, compute_dtype=None)
    assert torch.equal(PatchEmbed((64, 48), wd, bd, compute_dtype=compute)(xd.to(compute).cuda()).cpu(), refd)
    # the front-end's 16-bit spectrogram straight into the projection
    from conftest import synth
    clips = torch.from_numpy(np.stack([synth("noise0.1", 160000, 400 + i) for i in range(2)])).cuda()
    spec16 = mm.preprocessing_audio_batch(clips, dtype=compute)
    spec32 = mm.preprocessing_audio_batch(clips)
    pe = PatchEmbed((384, 128), w, b, compute_dtype=compute, out_dtype=compute)
    assert spec16.dtype == compute and torch.equal(pe(spec16), pe(spec32))


def test_patch_embed_out_with_class_token_row_and_strided_batch(mm, O):
    """`out` one row longer than the patches: row 0 is left for the class token (vit_model.py:382-383); images taken from a
    larger buffer (batch stride > H * W); a CPU tensor goes over and comes back."""
    from multi_modal_amd.patch_embed import PatchEmbed
    B, H, W, E = 7, 64, 48, 384                    # 12 patches per image, 84 in all: one full step and a partial one
    x, w, b = dyadic(B, H, W, E, 91)
    ref = O.patch_embed(x, w, b, compute_dtype=None)
    pe = PatchEmbed((H, W), w, b)
    big = torch.zeros(B, H + 5, W, device="cuda")
    big[:, :H] = x[:, 0].cuda()
    tokens = torch.full((B, 13, E), -7.0, device="cuda")
    r = pe(big[:, :H], out=tokens)
    assert r.data_ptr() == tokens.data_ptr()
    assert torch.equal(tokens[:, 1:].cpu(), ref) and bool((tokens[:, 0] == -7.0).all())
    y_cpu = pe(x)
    assert not y_cpu.is_cuda and torch.equal(y_cpu, ref)
    assert pe(x[:0].cuda()).shape == (0, 12, E)
    with pytest.raises(ValueError):
        pe(torch.zeros(2, H, W + 16, device="cuda"))


def test_patch_embed_refusals(mm):
    from multi_modal_amd import MmfeError
    from multi_modal_amd.patch_embed import PatchEmbed
    w = torch.zeros(768, 1, 16, 16)
    with pytest.raises(MmfeError):
        PatchEmbed((64, 64), torch.zeros(192, 1, 16, 16))             # embedding size without a kernel instance
    with pytest.raises(MmfeError):
        PatchEmbed((8, 64), w)                                        # smaller than one patch
    with pytest.raises(ValueError):
        PatchEmbed((64, 64), torch.zeros(768, 3, 16, 16))             # the audio projection has one input channel
    with pytest.raises(ValueError):
        PatchEmbed((64, 64), w, compute_dtype=torch.float32)


def test_spectrogram_to_tokens_pipeline(mm, O):
    """Waveforms -> normalised log-mel (B, 384, 128) -> tokens (B, 192, 768), both steps on the device: against the oracle's
    patch embedding of the oracle's spectrogram, within the front-end's own tolerance carried through the projection."""
    from multi_modal_amd.patch_embed import PatchEmbed
    from conftest import synth
    clips = torch.from_numpy(np.stack([synth("noise0.1", 160000, 300 + i) for i in range(3)]))
    g = torch.Generator().manual_seed(9)
    w = torch.randn(768, 1, 16, 16, generator=g) * 0.02
    b = torch.randn(768, generator=g) * 0.1
    spec = mm.preprocessing_audio_batch(clips.cuda())
    tokens = PatchEmbed((384, 128), w, b)(spec.unsqueeze(1))
    ref_spec = torch.stack([O.preprocessing_audio(c[None]) for c in clips])
    ref = O.patch_embed(ref_spec, w, b, compute_dtype=torch.float16, out_dtype=torch.float64)
    same_spec = O.patch_embed(spec.cpu(), w, b, compute_dtype=torch.float16, out_dtype=torch.float64)
    scale = O.patch_embed(ref_spec.abs(), w.abs(), None, compute_dtype=torch.float16, out_dtype=torch.float64)
    assert tokens.shape == (3, 192, 768)
    assert ((tokens.cpu().double() - same_spec).abs() <= 2e-5 * scale + 1e-6).all()       # the projection itself
    assert ((tokens.cpu().double() - ref).abs() <= 2e-3 * scale).all()                     # end to end: half-precision rounding of a 5e-4 input difference


def test_waveform_to_tokens_in_one_hip_graph_and_from_threads(mm, O):
    """Neither launch allocates or synchronises: the two-kernel chain waveform -> float16 spectrogram -> float16 tokens is recorded into ONE
    HIP graph after a warm-up call and replayed on new data; and one PatchEmbed object serves four host threads on four streams."""
    import threading
    from multi_modal_amd.patch_embed import PatchEmbed
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(768, 1, 16, 16, generator=torch.Generator().manual_seed(4)) * 0.03
    pe = PatchEmbed((384, 128), w, None, compute_dtype=torch.float16, out_dtype=torch.float16)
    x = 0.1 * torch.randn(6, 160000, device="cuda", generator=g)
    spec = torch.empty(6, 384, 128, device="cuda", dtype=torch.float16)
    tokens = torch.empty(6, 193, 768, device="cuda", dtype=torch.float16)

    def chain():
        mm.preprocessing_audio_batch(x, out=spec, dtype=torch.float16)
        pe(spec, out=tokens)

    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            chain()
    torch.cuda.current_stream().wait_stream(side)
    x.copy_(torch.roll(x, 11, dims=-1) * 0.7)
    want = pe(mm.preprocessing_audio_batch(x, dtype=torch.float16)).clone()
    tokens.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(tokens[:, 1:], want) and bool((tokens[:, 0] == 0).all())

    imgs = [torch.randn(40, 384, 128, device="cuda", generator=g) for _ in range(4)]
    serial = [pe(im) for im in imgs]
    torch.cuda.synchronize()
    got, errs = [None] * 4, []

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    y = pe(imgs[t])
                s.synchronize()
            got[t] = y
        except Exception as e:
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    assert all(torch.equal(got[t], serial[t]) for t in range(4))


def test_epic_spectrogram_to_tokens_pipeline(mm, O):
    """EPIC: 1.278 s slices -> log spectrogram / 255, (B, 256, 256) -> the `audio.view(-1, 1, 256, 256)` of
    classification/vggsound/models/mbt.py:74 -> 256 tokens, both steps on the device, against the oracle's chain."""
    from multi_modal_amd.patch_embed import PatchEmbed
    from conftest import synth
    x = np.stack([synth("noise0.1", 30672, 500 + i) for i in range(3)])
    g = torch.Generator().manual_seed(19)
    w = torch.randn(768, 1, 16, 16, generator=g) * 0.05
    b = torch.randn(768, generator=g) * 0.1
    spec = mm.log_specgram_batch(torch.from_numpy(x).cuda(), div255=True)
    assert spec.shape == (3, 256, 256)
    tokens = PatchEmbed((256, 256), w, b)(spec.view(-1, 1, 256, 256))
    ref_spec = torch.stack([O.spec_to_tensor(O.log_specgram(xi))[0] for xi in x])
    ref = O.patch_embed(ref_spec, w, b, compute_dtype=torch.float16, out_dtype=torch.float64)
    same = O.patch_embed(spec.cpu(), w, b, compute_dtype=torch.float16, out_dtype=torch.float64)
    scale = O.patch_embed(ref_spec.abs(), w.abs(), None, compute_dtype=torch.float16, out_dtype=torch.float64)
    assert tokens.shape == (3, 256, 768)
    assert ((tokens.cpu().double() - same).abs() <= 2e-5 * scale + 1e-6).all()
    assert ((tokens.cpu().double() - ref).abs() <= 2e-3 * scale + 1e-5).all()


def test_patch_embed_layer_trains_like_the_reference_module(mm, O, golden_dir):
    """PatchEmbedLayer is the reference's PatchEmbed module with the audio projection: its state_dict keys are `proj.weight` /
    `proj.bias` (what a checkpoint of the reference holds for this layer), forward equals the autocast convolution, the gradients
    of weight, bias and image equal those of torch's convolution on the rounded operands, and after an optimizer step the next
    forward uses the new weights (re-packed on the device) for float32 and for 16-bit images."""
    from multi_modal_amd.patch_embed import PatchEmbedLayer
    torch.manual_seed(0)
    layer = PatchEmbedLayer((384, 128), embed_dim=768, compute_dtype=torch.float16).cuda()
    assert sorted(layer.state_dict().keys()) == ["proj.bias", "proj.weight"] and layer.num_patches == 192
    ref = torch.nn.Conv2d(1, 768, 16, 16)
    layer.load_state_dict({"proj.weight": ref.weight.detach().clone(), "proj.bias": ref.bias.detach().clone()})
    x = (torch.randn(3, 1, 384, 128) * 1.2).cuda().requires_grad_(True)
    opt = torch.optim.SGD(layer.parameters(), lr=0.05)
    for step in range(3):
        opt.zero_grad()
        x.grad = None
        y = layer(x)
        loss = (y * torch.linspace(-1, 1, 768, device="cuda")).sum() / 100.0 + (y ** 2).mean()
        loss.backward()
        # the same step on the CPU: convolution of the operands as rounded by autocast, in float64
        w64 = layer.proj.weight.detach().cpu().half().double().requires_grad_(True)
        b64 = layer.proj.bias.detach().cpu().half().double().requires_grad_(True)
        x64 = x.detach().cpu().half().double().requires_grad_(True)
        y64 = torch.nn.functional.conv2d(x64, w64, b64, stride=16).flatten(2).transpose(1, 2)
        loss64 = (y64 * torch.linspace(-1, 1, 768, dtype=torch.float64)).sum() / 100.0 + (y64 ** 2).mean()
        loss64.backward()
        assert (y.detach().cpu().double() - y64.detach()).abs().max() <= 2e-4
        for got, want in ((layer.proj.weight.grad, w64.grad), (layer.proj.bias.grad, b64.grad), (x.grad, x64.grad)):
            err = (got.cpu().double() - want).abs().max()
            assert err <= 2e-4 * want.abs().max() + 1e-7, (step, err, want.abs().max())
        opt.step()                                   # the parameters move: the next forward must see them
    with torch.no_grad():
        y32 = layer(x.detach())
        y16 = layer(x.detach().half())               # first call with a 16-bit image, after the weights have moved
        want = O.patch_embed(x.detach().cpu(), layer.proj.weight.detach().cpu(), layer.proj.bias.detach().cpu(), compute_dtype=torch.float16,
                             out_dtype=torch.float64)
    assert (y32.cpu().double() - want).abs().max() <= 2e-4
    assert torch.equal(y16, y32)
