"""Double tensors on the fp64 matrix cores (csrc/correlation_mfma_f64.hip; VERDICT r5 missing #4): the reference dispatches double
first-class (correlation_cuda_kernel.cu:386-415, :522-554).  FlowNetC's configuration, forward and both gradients.

Forward: the reference sums the channel products of a DOUBLE tensor in a float accumulator (`float acc0`, correlation_cuda_kernel.cu:112-124)
-- its double outputs carry fp32 rounding.  The oracle and the one-thread-per-output kernel restate that; the matrix-core kernel sums in
fp64 (a deliberate superset): it must agree with an independent fp64 formulation to 1e-13 of the output scale and with the
reference's float-accumulated result to fp32 rounding.  Backward: the reference accumulates in the tensor's type (:214-229): fp64 throughout."""
import numpy as np
import pytest
import torch

from conftest import max_abs

pytestmark = pytest.mark.gpu

P = (20, 1, 20, 1, 2)
CASES = [(1, 64, 8, 16), (2, 64, 10, 40), (1, 64, 6, 72), (2, 128, 20, 64), (1, 64, 2, 8), (1, 64, 44, 36), (2, 64, 48, 64)]


def _data(case, seed_off=0):
    B, C, H, W = case
    rng = np.random.default_rng(B * 131 + C * 7 + H * 3 + W + seed_off)
    a = rng.standard_normal((B, C, H, W))
    b = rng.standard_normal((B, C, H, W))
    a[0, 1] *= 1e3; b[0, 2] *= 1e-3                       # channels of very different magnitude: nothing is block-scaled in fp64
    return a, b, rng.standard_normal((B, 441, H, W))


@pytest.mark.parametrize("case", CASES)
def test_correlation_f64_forward_vs_oracle(dev, oracle, case):
    import fn2_capi
    a, b, _ = _data(case)
    ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.full((case[0], 441, case[2], case[3]), float("nan"), dtype=torch.float64, device=dev)
    fn2_capi.correlation_forward(ad, bd, *P, out=out)                                       # AUTO: the fp64 matrix-core kernel
    assert torch.isfinite(out).all(), "output elements left unwritten"
    direct = fn2_capi.correlation_forward(ad, bd, *P, algo=fn2_capi.FN2_CORR_DIRECT)
    scale = float(direct.abs().max())
    # an independent fp64 formulation: 441 shifted channel means
    H, W = case[2], case[3]
    p2 = torch.nn.functional.pad(bd, (20, 20, 20, 20))
    truth = torch.cat([(ad * p2[:, :, 20 + 2 * tj:20 + 2 * tj + H, 20 + 2 * ti:20 + 2 * ti + W]).mean(1, keepdim=True)
                       for tj in range(-10, 11) for ti in range(-10, 11)], 1)
    assert float((out - truth).abs().max()) <= 1e-13 * scale, float((out - truth).abs().max()) / scale
    # the reference's semantics (float accumulator): the one-thread-per-output kernel and the oracle, to fp32 rounding
    assert float((out - direct).abs().max()) <= 2e-6 * scale
    assert float((direct - truth).abs().max()) > 1e-12 * scale, "the reference-style kernel is expected to carry fp32 rounding"
    ref = oracle.corr_fwd(a, b, *P)
    assert ref.dtype == np.float64 and max_abs(out.cpu().numpy(), ref) <= 2e-6 * scale
    # the fused epilogue (LeakyReLU + channel slice of a concat buffer) in double
    buf = torch.full((case[0], 3 + 441, case[2], case[3]), float("nan"), dtype=torch.float64, device=dev)
    fn2_capi.correlation_forward_fused(ad, bd, buf, 3, 0.1, *P)
    want = torch.where(out > 0, out, out * float(np.float32(0.1)))
    assert torch.equal(buf[:, 3:], want) and torch.isnan(buf[:, :3]).all()


@pytest.mark.parametrize("case", CASES)
def test_correlation_f64_backward_vs_oracle(dev, oracle, case):
    import fn2_capi
    a, b, go = _data(case, 1)
    ad, bd, gd = (torch.from_numpy(x).to(dev) for x in (a, b, go))
    g1 = torch.full_like(ad, float("nan")); g2 = torch.full_like(bd, float("nan"))
    fn2_capi.correlation_backward(ad, bd, gd, *P, out=(g1, g2))
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all(), "gradient elements left unwritten"
    d1, d2 = fn2_capi.correlation_backward(ad, bd, gd, *P, algo=fn2_capi.FN2_CORR_DIRECT)
    r1, r2 = oracle.corr_bwd(a, b, go, *P)
    for got, direct, ref in ((g1, d1, r1), (g2, d2, r2)):
        scale = float(direct.abs().max())
        assert float((got - direct).abs().max()) <= 1e-13 * scale
        assert max_abs(got.cpu().numpy(), ref) <= 1e-13 * scale


def test_correlation_f64_gradcheck_through_the_module(dev):
    """What double is for: torch.autograd.gradcheck of the Correlation module (central differences in fp64) on the matrix-core path."""
    from networks.correlation_package.correlation import Correlation
    g = torch.Generator().manual_seed(5)
    a = torch.randn(1, 64, 4, 8, generator=g, dtype=torch.float64).to(dev).requires_grad_()
    b = torch.randn(1, 64, 4, 8, generator=g, dtype=torch.float64).to(dev).requires_grad_()
    sel = torch.randn(1, 441, 4, 8, generator=g, dtype=torch.float64).to(dev)
    corr = Correlation(20, 1, 20, 1, 2, 1)
    assert torch.autograd.gradcheck(lambda x, y: (corr(x, y) * sel).sum(), (a, b), eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=0.0)


def test_correlation_f64_full_size_timing_sanity(dev):
    """8 x 256 x 48 x 64 (BASELINE configs[1]) in double: AUTO = DIRECT to rounding on a sample of planes, and far faster."""
    import fn2_capi
    g = torch.Generator().manual_seed(6)
    a = torch.randn(8, 256, 48, 64, generator=g, dtype=torch.float64).to(dev)
    b = torch.randn(8, 256, 48, 64, generator=g, dtype=torch.float64).to(dev)
    go = torch.randn(8, 441, 48, 64, generator=g, dtype=torch.float64).to(dev)
    out = fn2_capi.correlation_forward(a, b, *P)
    ref = fn2_capi.correlation_forward(a, b, *P, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max())          # (the reference's float accumulator, see above)
    g1, g2 = fn2_capi.correlation_backward(a, b, go, *P)
    d1, d2 = fn2_capi.correlation_backward(a, b, go, *P, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((g1 - d1).abs().max()) <= 1e-13 * float(d1.abs().max()) and float((g2 - d2).abs().max()) <= 1e-13 * float(d2.abs().max())

    def us(fn, n=5):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n
    t_f, t_fd = us(lambda: fn2_capi.correlation_forward(a, b, *P, out=out)), us(lambda: fn2_capi.correlation_forward(a, b, *P, algo=fn2_capi.FN2_CORR_DIRECT, out=ref), 2)
    t_b, t_bd = us(lambda: fn2_capi.correlation_backward(a, b, go, *P, out=(g1, g2))), us(lambda: fn2_capi.correlation_backward(a, b, go, *P, algo=fn2_capi.FN2_CORR_DIRECT, out=(d1, d2)), 1)
    print(f"double 8x256x48x64: forward {t_f:.0f} us (general kernel {t_fd:.0f}), backward {t_b:.0f} us (general kernel {t_bd:.0f})")
    assert t_f < 0.5 * t_fd and t_b < 0.2 * t_bd
