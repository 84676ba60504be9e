"""Known-answer and property tests of the oracle, derived from the reference kernels' semantics
(SURVEY.md 8c): they pin channel order, zero padding, border clamping and the closed forms, and
check the explicit backward passes against autograd of an independent PyTorch formulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import max_abs

FNC = dict(pad=20, k=1, md=20, s1=1, s2=2)  # FlowNetC.py:28


def torch_corr(in1, in2, pad, md, s2):
    """Independent formulation for k=1, s1=1, pad==md: shifted products of the padded in2."""
    dr = md // s2
    H, W = in1.shape[-2:]
    p2 = F.pad(in2, (pad, pad, pad, pad))
    outs = []
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            y0, x0 = pad + tj * s2, pad + ti * s2
            outs.append((in1 * p2[:, :, y0:y0 + H, x0:x0 + W]).mean(1, keepdim=True))
    return torch.cat(outs, 1)


def test_corr_ones_counts_padding(oracle):
    x = np.ones((1, 4, 6, 8), np.float32)
    out = oracle.corr_fwd(x, x, **FNC)
    assert out.shape == (1, 441, 6, 8)
    for tj in range(21):
        for ti in range(21):
            exp = np.zeros((6, 8), np.float32)
            dy, dx = 2 * (tj - 10), 2 * (ti - 10)
            for y in range(6):
                for xx in range(8):
                    if 0 <= y + dy < 6 and 0 <= xx + dx < 8:
                        exp[y, xx] = 1.0
            assert np.array_equal(out[0, tj * 21 + ti], exp), (tj, ti)


def test_corr_impulse_channel_order(oracle):
    a = np.zeros((1, 8, 10, 12), np.float32)
    b = np.zeros_like(a)
    a[0, 3, 4, 5] = 2.0
    b[0, 3, 8, 1] = 3.0   # displacement dy=+4 (tj=12), dx=-4 (ti=8)
    out = oracle.corr_fwd(a, b, **FNC)
    nz = np.argwhere(out != 0)
    assert nz.tolist() == [[0, 12 * 21 + 8, 4, 5]]
    assert out[0, 12 * 21 + 8, 4, 5] == np.float32(6.0 / 8)


def test_corr_channel_permutation_invariant(oracle):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1, 32, 6, 6)).astype(np.float64)
    b = rng.standard_normal((1, 32, 6, 6)).astype(np.float64)
    perm = rng.permutation(32)
    o1 = oracle.corr_fwd(a, b, 4, 1, 4, 1, 2)
    o2 = oracle.corr_fwd(a[:, perm], b[:, perm], 4, 1, 4, 1, 2)
    assert max_abs(o1, o2) < 1e-6   # fp32 accumulation inside (reference :112), order differs


@pytest.mark.parametrize("shape,pad,md,s2", [((2, 16, 8, 10), 4, 4, 2), ((1, 5, 7, 9), 3, 3, 1), ((1, 8, 6, 6), 20, 20, 2)])
def test_corr_matches_torch_and_autograd(oracle, shape, pad, md, s2):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(shape, dtype=torch.float64, generator=g, requires_grad=True)
    b = torch.randn(shape, dtype=torch.float64, generator=g, requires_grad=True)
    ref = torch_corr(a, b, pad, md, s2)
    out = oracle.corr_fwd(a.detach().numpy(), b.detach().numpy(), pad, 1, md, 1, s2)
    assert max_abs(out, ref.detach().numpy()) < 2e-6      # oracle accumulates in fp32 like the reference
    go = torch.randn(ref.shape, dtype=torch.float64, generator=g)
    ga, gb = torch.autograd.grad(ref, (a, b), go)
    g1, g2 = oracle.corr_bwd(a.detach().numpy(), b.detach().numpy(), go.numpy(), pad, 1, md, 1, s2)
    assert max_abs(g1, ga.numpy()) < 1e-12 and max_abs(g2, gb.numpy()) < 1e-12


def test_corr_shapes(oracle):
    from oracle.oracle import corr_shapes
    assert corr_shapes(48, 64, 20, 1, 20, 1, 2)[2:] == (441, 48, 64)
    assert corr_shapes(48, 64, 3, 3, 20, 1, 2)[2:] == (441, 12, 28)   # Function defaults (correlation.py:9)
    assert corr_shapes(16, 16, 4, 1, 4, 2, 2)[2:] == (25, 8, 8)


def test_corr_fp32_error_budget(oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((1, 256, 6, 8))
    b = rng.standard_normal((1, 256, 6, 8))
    o64 = oracle.corr_fwd(a, b, **FNC)   # products in fp64, fp32 accumulation (reference semantics for double)
    o32 = oracle.corr_fwd(a.astype(np.float32), b.astype(np.float32), **FNC)
    exact = torch_corr(torch.from_numpy(a), torch.from_numpy(b), 20, 20, 2).numpy()
    assert max_abs(o32, exact) < 5e-6 and max_abs(o64, exact) < 5e-6   # << the 1e-4 parity budget


# ------------------------------------------------------------------ resample2d
def test_resample_zero_flow_identity(oracle):
    rng = np.random.default_rng(0)
    img = rng.standard_normal((2, 3, 8, 10)).astype(np.float32)
    flow = np.zeros((2, 2, 8, 10), np.float32)
    assert np.array_equal(oracle.resample_fwd(img, flow), img)
    assert np.array_equal(oracle.resample_fwd(img, flow, 1, False), img)


def test_resample_integer_shift_and_border(oracle):
    img = np.arange(2 * 6 * 8, dtype=np.float32).reshape(1, 2, 6, 8)
    flow = np.zeros((1, 2, 6, 8), np.float32)
    flow[:, 0] = 2.0    # dx (channel 0 = horizontal, resample2d_kernel.cu:38)
    flow[:, 1] = -1.0   # dy
    out = oracle.resample_fwd(img, flow)
    for y in range(6):
        for x in range(8):
            assert out[0, 1, y, x] == img[0, 1, max(y - 1, 0), min(x + 2, 7)]
    flow[:] = 1e6       # far out of range -> bottom-right pixel
    assert np.all(oracle.resample_fwd(img, flow)[0, 0] == img[0, 0, 5, 7])
    flow[:] = -1e6
    assert np.all(oracle.resample_fwd(img, flow)[0, 0] == img[0, 0, 0, 0])


def test_resample_matches_grid_sample_and_autograd(oracle):
    g = torch.Generator().manual_seed(2)
    B, C, H, W = 2, 3, 9, 11
    img = torch.randn(B, C, H, W, generator=g, requires_grad=True, dtype=torch.float64)
    flow = (torch.randn(B, 2, H, W, generator=g, dtype=torch.float64) * 2.5)
    flow = (flow + 0.3).requires_grad_(True)   # keep away from integer coordinates (kinks)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    gx = (xs + flow[:, 0]) / (W - 1) * 2 - 1
    gy = (ys + flow[:, 1]) / (H - 1) * 2 - 1
    ref = F.grid_sample(img, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="border", align_corners=True)
    out = oracle.resample_fwd(img.detach().numpy(), flow.detach().numpy())
    assert max_abs(out, ref.detach().numpy()) < 5e-6
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gi, gf = torch.autograd.grad(ref, (img, flow), go)
    gimg, gflow = oracle.resample_bwd(img.detach().numpy(), flow.detach().numpy(), go.numpy())
    assert max_abs(gimg, gi.numpy()) < 2e-5
    # grid_sample's border mode zeroes the flow gradient outside the image; the reference keeps the
    # clamped-corner difference there.  Compare where the sample point is strictly inside.
    xf = (xs + flow[:, 0]).detach().numpy()
    yf = (ys + flow[:, 1]).detach().numpy()
    inside = ((xf > 0) & (xf < W - 1) & (yf > 0) & (yf < H - 1))[:, None].repeat(2, 1)
    assert np.max(np.abs((gflow - gf.numpy())[inside])) < 2e-5


def test_resample_nearest(oracle):
    img = np.arange(5 * 7, dtype=np.float32).reshape(1, 1, 5, 7)
    flow = np.zeros((1, 2, 5, 7), np.float32)
    flow[0, 0] = 0.6
    flow[0, 1] = 0.4
    out = oracle.resample_fwd(img, flow, 1, False)
    for y in range(5):
        for x in range(7):
            assert out[0, 0, y, x] == img[0, 0, y, min(x + 1, 6)]


# ------------------------------------------------------------------ channelnorm
def test_chnorm_kat(oracle):
    x = np.zeros((1, 2, 2, 2), np.float32)
    x[0, :, 0, 0] = [3, 4]
    out = oracle.chnorm_fwd(x)
    assert out[0, 0, 0, 0] == 5.0 and out[0, 0, 1, 1] == 0.0
    gin = oracle.chnorm_bwd(x, out, np.ones_like(out))
    assert np.all(np.isfinite(gin)) and gin[0, 0, 1, 1] == 0.0        # zero pixel: 0, not NaN
    assert abs(gin[0, 0, 0, 0] - 0.6) < 1e-6 and abs(gin[0, 1, 0, 0] - 0.8) < 1e-6


def test_chnorm_autograd(oracle):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 5, 6, dtype=torch.float64, generator=g, requires_grad=True)
    ref = x.pow(2).sum(1, keepdim=True).sqrt()
    out = oracle.chnorm_fwd(x.detach().numpy())
    assert max_abs(out, ref.detach().numpy()) < 1e-6
    go = torch.randn(ref.shape, dtype=torch.float64, generator=g)
    (gx,) = torch.autograd.grad(ref, x, go)
    assert max_abs(oracle.chnorm_bwd(x.detach().numpy(), out, go.numpy()), gx.numpy()) < 1e-6
