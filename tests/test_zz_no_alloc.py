"""mmfe_run_noalloc (the entry point that allocates nothing: the per-row extremes of clamp_below_max plans live in a
static device array inside the library) against mmfe_run, through the C ABI, bit for bit: two calls in a row, the second
with a smaller batch and shorter rows, then a plan without a clamp, then the row limit.  The reference has no
counterpart of this entry point; mmfe_run is the checker here (it is checked against the oracle in test_gpu_parity.py).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mm():
    import multi_modal_amd as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


def _both(plan, x):
    rows, n = x.shape
    frames, feat = plan.out_shape(n)
    shape = (rows, feat, frames) if plan.spec.layout == 1 else (rows, frames, feat)
    a = torch.full(shape, float("nan"), device=x.device)
    b = torch.full(shape, float("nan"), device=x.device)
    stream = torch.cuda.current_stream().cuda_stream
    plan.run_ptr(x.data_ptr(), x.stride(0), rows, n, a.data_ptr(), stream)
    plan.run_noalloc_ptr(x.data_ptr(), x.stride(0), rows, n, b.data_ptr(), stream)
    torch.cuda.synchronize()
    return a, b


def test_noalloc_matches_run_on_two_calls_in_a_row(mm):
    from multi_modal_amd import _native as N
    from multi_modal_amd import logmel
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(77)
    plan = logmel._plan(0, 400, 160, 80, 16000, False)                 # n_fft 400 radix kernel, clamp_below_max = 8
    assert plan.spec.clamp_below_max > 0 and plan.kernel == "rfft400"
    # first call: 96 utterances of 3 s, amplitudes spread over 4 decades so that the clamp bites in some rows only
    x1 = torch.randn(96, 48000, device=dev, generator=g) * torch.logspace(-4, 0, 96, device=dev)[:, None]
    x1[5, 1000:] = 0.0                                                  # mostly silent row: clamp raises the floor cells
    a1, b1 = _both(plan, x1)
    assert torch.isfinite(a1).all() and torch.equal(a1, b1)
    # second call, smaller: fewer rows, shorter rows; stale extremes of the first call must not leak in
    x2 = 0.01 * torch.randn(7, 16000, device=dev, generator=g)
    a2, b2 = _both(plan, x2)
    assert torch.isfinite(a2).all() and torch.equal(a2, b2)
    # and again the first input: the array now holds the second call's extremes
    a3, b3 = _both(plan, x1)
    assert torch.equal(a3, a1) and torch.equal(b3, a1)

    # the any-length kernel with a clamp (float64 window path)
    plan64 = logmel._plan(0, 400, 160, 80, 16000, True)
    assert plan64.kernel == "dftgen"
    a4, b4 = _both(plan64, x2)
    assert torch.isfinite(a4).all() and torch.equal(a4, b4)

    # a plan without a clamp uses no scratch: same result as well
    vplan = mm.kaldi._fbank_plan(0, sample_frequency=16000, frame_length=25.0, frame_shift=25, num_mel_bins=128,
                                 window_type="hanning", blackman_coeff=0.42, preemphasis_coefficient=0.97, remove_dc_offset=True,
                                 round_to_power_of_two=True, low_freq=20.0, high_freq=0.0, use_log_fbank=True)
    x5 = 0.1 * torch.randn(9, 16000, device=dev, generator=g)
    a5, b5 = _both(vplan, x5)
    assert torch.equal(a5, b5)

    # the limit: more rows than the static array holds are refused with a clear message, nothing is launched
    assert N.NOALLOC_MAX_ROWS == 65536
    big = torch.zeros(N.NOALLOC_MAX_ROWS + 1, 800, device=dev)
    frames, feat = plan.out_shape(800)
    out = torch.empty(big.shape[0], feat, frames, device=dev)
    with pytest.raises(N.MmfeError, match="65536"):
        plan.run_noalloc_ptr(big.data_ptr(), big.stride(0), big.shape[0], 800, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    # exactly the limit is served
    okb = big[: N.NOALLOC_MAX_ROWS]
    a6, b6 = _both(plan, okb)
    assert torch.equal(a6, b6)


def test_noalloc_two_streams_ordered_by_events(mm):
    """Every mmfe_run_noalloc call on a GPU uses that GPU's ONE static array, whatever plan or stream it comes from: calls with
    clamp_below_max plans must not overlap on the device.  Two plans on two streams are fine when the second stream waits for the
    first (the documented way to use them): each result equals mmfe_run's on its own stream."""
    from multi_modal_amd import logmel
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(78)
    plan_a = logmel._plan(0, 400, 160, 80, 16000, False, stream=1)      # two plans (the Python layer keeps one per stream)
    plan_b = logmel._plan(0, 400, 160, 80, 16000, False, stream=2)
    assert plan_a is not plan_b
    xa = torch.randn(64, 32000, device=dev, generator=g) * torch.logspace(-3, 0, 64, device=dev)[:, None]
    xb = 0.05 * torch.randn(33, 24000, device=dev, generator=g)
    ref_a, _ = _both(plan_a, xa)
    ref_b, _ = _both(plan_b, xb)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    out_a, out_b = torch.empty_like(ref_a), torch.empty_like(ref_b)
    torch.cuda.synchronize()
    for _ in range(3):
        plan_a.run_noalloc_ptr(xa.data_ptr(), xa.stride(0), xa.shape[0], xa.shape[1], out_a.data_ptr(), sa.cuda_stream)
        sb.wait_stream(sa)                                              # the second call starts after the first has finished
        plan_b.run_noalloc_ptr(xb.data_ptr(), xb.stride(0), xb.shape[0], xb.shape[1], out_b.data_ptr(), sb.cuda_stream)
        sa.wait_stream(sb)
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)
