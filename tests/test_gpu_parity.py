"""GPU parity: the gfx950 HIP kernels, reached through the reference-named pybind modules and the
C ABI, against the CPU oracle on identical seeded inputs and against the committed golden vectors
(reference kernels under the SIMT shim).

Tolerance: BASELINE.json's north star states <= 1e-4 fp32 max-abs.  TOL below is that bound; most
paths are far tighter (channelnorm and the resample2d gathers restate the reference's operation
order and are checked at 1e-6; the atomically-scattered resample grad_img and the MFMA correlation,
whose summation order differs from the reference's 32-lane tree, get the full budget)."""
import os

import numpy as np
import pytest
import torch

from conftest import full_size_inputs, golden_files, graded_corr_inputs, max_abs

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU (run through gpurun)"
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_native_library_is_loaded(dev):
    import fn2_capi
    import correlation_cuda  # noqa: F401
    fn2_capi.lib()
    maps = open("/proc/self/maps").read()
    assert "libflownet2_hip.so" in maps
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, f"more than one HIP runtime mapped: {hips}"


# ------------------------------------------------------------------ channelnorm
@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 2, 7, 9), (3, 5, 6, 10), (8, 3, 384, 512), (8, 2, 384, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
def test_channelnorm(dev, oracle, shape, dtype):
    import channelnorm_cuda
    if dtype != torch.float32 and shape[-1] == 512:
        pytest.skip("full size checked in fp32")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    x[0, :, 0, 0] = 0
    x = x.to(dtype)
    xd = x.to(dev)
    out = torch.zeros((shape[0], 1) + shape[2:], dtype=dtype, device=dev)
    assert channelnorm_cuda.forward(xd, out, 2) == 1
    go = torch.randn(out.shape, generator=g, dtype=torch.float32).to(dtype)
    gin = torch.zeros_like(xd)
    assert channelnorm_cuda.backward(xd, out, go.to(dev), gin, 2) == 1
    if dtype == torch.float16:
        # oracle has no half type: compare against the fp32 oracle on the half-rounded inputs
        xo = x.float().numpy()
        ref = oracle.chnorm_fwd(xo)
        assert max_abs(out.float().cpu().numpy(), ref) <= 2e-3 * max(1.0, float(np.abs(ref).max()))
        refg = oracle.chnorm_bwd(xo, out.float().cpu().numpy(), go.float().numpy())
        assert max_abs(gin.float().cpu().numpy(), refg) <= 4e-3 * max(1.0, float(np.abs(refg).max()))
        return
    xo = x.numpy()
    ref = oracle.chnorm_fwd(xo)
    assert max_abs(out.cpu().numpy(), ref) <= 1e-6
    refg = oracle.chnorm_bwd(xo, ref, go.numpy())
    assert max_abs(gin.cpu().numpy(), refg) <= 1e-6
    assert torch.isfinite(gin).all()


def test_channelnorm_strided_grad_output(dev, oracle):
    """gradOutput arrives as a channel slice of a cat gradient: strides are honoured."""
    import channelnorm_cuda
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 3, 16, 20, generator=g)
    big = torch.randn(4, 9, 16, 20, generator=g)
    xd = x.to(dev)
    out = torch.zeros(4, 1, 16, 20, device=dev)
    channelnorm_cuda.forward(xd, out, 2)
    go_view = big.to(dev)[:, 4:5]
    assert not go_view.is_contiguous()
    gin = torch.zeros_like(xd)
    channelnorm_cuda.backward(xd, out, go_view, gin, 2)
    ref = oracle.chnorm_bwd(x.numpy(), out.cpu().numpy(), big[:, 4:5].contiguous().numpy())
    assert max_abs(gin.cpu().numpy(), ref) <= 1e-6
    # odd sizes take the scalar kernel, with fully general strides
    x2 = torch.randn(2, 2, 5, 7, generator=g)
    o2 = torch.zeros(2, 1, 5, 7, device=dev)
    channelnorm_cuda.forward(x2.to(dev), o2, 2)
    gbig = torch.randn(2, 1, 7, 5, generator=g)
    gv = gbig.to(dev).permute(0, 1, 3, 2)
    g2 = torch.zeros(2, 2, 5, 7, device=dev)
    channelnorm_cuda.backward(x2.to(dev), o2, gv, g2, 2)
    ref2 = oracle.chnorm_bwd(x2.numpy(), o2.cpu().numpy(), gbig.permute(0, 1, 3, 2).contiguous().numpy())
    assert max_abs(g2.cpu().numpy(), ref2) <= 1e-6


# ------------------------------------------------------------------ resample2d
def _flow(g, shape, scale):
    f = torch.randn(shape, generator=g) * scale
    flat = f.view(-1)
    idx = torch.randint(0, flat.numel(), (max(1, flat.numel() // 100),), generator=g)
    flat[idx] *= 20.0   # 1 % far out of range: forces border clamps
    return f


@pytest.mark.parametrize("shape", [(2, 3, 16, 24, 16, 24), (1, 2, 9, 11, 9, 11), (1, 3, 10, 14, 8, 12), (8, 3, 384, 512, 384, 512)])
@pytest.mark.parametrize("bilinear", [True, False])
def test_resample2d(dev, oracle, shape, bilinear):
    import resample2d_cuda
    B, C, Hi, Wi, H, W = shape
    g = torch.Generator().manual_seed(9)
    img = torch.rand(B, C, Hi, Wi, generator=g) - 0.5
    flow = _flow(g, (B, 2, H, W), 4.0)
    gout = torch.randn(B, C, H, W, generator=g)
    out = torch.zeros(B, C, H, W, device=dev)
    assert resample2d_cuda.forward(img.to(dev), flow.to(dev), out, 1, bilinear) == 1
    ref = oracle.resample_fwd(img.numpy(), flow.numpy(), 1, bilinear)
    assert max_abs(out.cpu().numpy(), ref) <= 1e-6
    gimg = torch.zeros(B, C, Hi, Wi, device=dev)
    gflow = torch.zeros(B, 2, H, W, device=dev)
    assert resample2d_cuda.backward(img.to(dev), flow.to(dev), gout.to(dev), gimg, gflow, 1, bilinear) == 1
    rimg, rflow = oracle.resample_bwd(img.numpy(), flow.numpy(), gout.numpy(), 1, bilinear)
    assert max_abs(gflow.cpu().numpy(), rflow) <= 1e-6
    assert max_abs(gimg.cpu().numpy(), rimg) <= TOL     # fp32 atomics: order differs from the oracle's


@pytest.mark.parametrize("path", golden_files("resample"), ids=os.path.basename)
def test_resample2d_golden(dev, path):
    """The HIP kernels against what the REFERENCE's own device code produced (tests/golden/resample_*.npz: resample2d_kernel.cu
    under the CPU SIMT shim).  The two small fixtures take the untiled kernels; `resample_tiled_*` (make_golden_resample_tiled.py)
    the tiled three-channel ones -- ragged tiles, a translation of (25, -18) px under the flow, noise, far outliers."""
    import resample2d_cuda
    g = np.load(path)
    img, flow, gout = to_dev(g["img"], dev), to_dev(g["flow"], dev), to_dev(g["gout"], dev)
    B, C, H, W = g["gout"].shape
    for bil in (1, 0):
        out = torch.full((B, C, H, W), float("nan"), device=dev)
        assert resample2d_cuda.forward(img, flow, out, 1, bool(bil)) == 1
        assert max_abs(out.cpu().numpy(), g[f"out_bil{bil}"]) <= 1e-6
    gimg, gflow = torch.zeros_like(img), torch.full_like(flow, float("nan"))
    assert resample2d_cuda.backward(img, flow, gout, gimg, gflow, 1, True) == 1
    assert max_abs(gflow.cpu().numpy(), g["gflow"]) <= 1e-6
    assert max_abs(gimg.cpu().numpy(), g["gimg"]) <= 5e-6 * max(1.0, float(np.abs(g["gimg"]).max()))


@pytest.mark.parametrize("path", golden_files("fullsize"), ids=os.path.basename)
def test_full_size_resample_chnorm_golden(dev, path):
    """Resample2d and ChannelNorm at the BASELINE shape (8 x 3 x 384 x 512, the SURVEY's white-noise flow with 1 % outliers) against
    the REFERENCE's own device code (tests/golden/make_golden_full_size.py): the kept rows of every tensor and the float64 sums of every
    (batch, channel) plane -- forward (bilinear and nearest), both gradients of the warp, the norm and its gradient."""
    import channelnorm_cuda
    import resample2d_cuda
    g = np.load(path)
    img, flow, gout, gnorm = (to_dev(a, dev) for a in full_size_inputs(g))
    rows = g["rows"]

    def close(name, t, tol, sum_tol):
        a = t.cpu().numpy()
        scale = max(1.0, float(np.abs(g[name + "_rows"]).max()))
        assert max_abs(a[:, :, rows], g[name + "_rows"]) <= tol * scale, name
        err = np.abs(a.astype(np.float64).sum(axis=(2, 3)) - g[name + "_sum"])
        assert np.all(err <= sum_tol * np.maximum(g[name + "_abs"], 1.0)), (name, float(err.max()))
    for bil, name in ((True, "warp"), (False, "warp_nearest")):
        out = torch.full_like(img, float("nan"))
        assert resample2d_cuda.forward(img, flow, out, 1, bil) == 1
        close(name, out, 1e-6, 1e-6)
    gi, gf = torch.zeros_like(img), torch.full_like(flow, float("nan"))
    assert resample2d_cuda.backward(img, flow, gout, gi, gf, 1, True) == 1
    close("gflow", gf, 1e-6, 1e-6)
    close("gimg", gi, 5e-6, 1e-6)            # fp32 atomics: an order of their own
    n = torch.full((img.shape[0], 1, img.shape[2], img.shape[3]), float("nan"), device=dev)
    assert channelnorm_cuda.forward(img, n, 2) == 1
    close("norm", n, 1e-6, 1e-6)
    gin = torch.full_like(img, float("nan"))
    assert channelnorm_cuda.backward(img, n, gnorm, gin, 2) == 1
    close("gnorm_in", gin, 1e-5, 1e-6)
    assert float(gin[0, :, 5, 7].abs().max()) == 0.0     # the exactly-zero pixel: 0, not nan (channelnorm_kernel.cu:93)


@pytest.mark.parametrize("shape", [(1, 16, 32), (2, 100, 200), (1, 33, 68), (3, 64, 64), (2, 97, 260), (1, 48, 96)])
@pytest.mark.parametrize("spread", [0.5, 4.0, 40.0, (4.0, 25.0, -18.0), (0.5, -41.0, 7.0), (2.0, 6.0, 300.0)])
def test_resample2d_three_channel_kernels(dev, oracle, shape, spread):
    """C = 3 on tileable maps (W % 4 == 0, H >= 16, W >= 32) takes the kernels that hold all three channel windows in LDS at once
    (forward: `resample_fwd_tiled_all`; backward: `resample_bwd_c3x`, two channels per 64-bit compare-and-swap, phase order
    alternating between workgroups).  Ragged tiles in both directions, the smallest tileable map, flows inside the +-16 px window
    (0.5, 4 px), mostly outside it (40 px: global atomics / global gathers), 1 % outliers, and translations of tens of pixels under
    the noise (the backward windows follow the tile's mean flow, `tile_window_offset`); read through the strides of a
    channel slice as models.py:133 passes it; the backward accumulates into a non-zero grad_input1 (resample2d.py:31)."""
    import resample2d_cuda
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.rand(B, 6, H, W, generator=g) - 0.5
    if isinstance(spread, tuple):   # a translation under the noise: the backward windows follow the tile's mean flow
        flow = _flow(g, (B, 2, H, W), spread[0])
        flow[:, 0] += spread[1]; flow[:, 1] += spread[2]
    else:
        flow = _flow(g, (B, 2, H, W), spread)
    gout = torch.randn(B, 3, H, W, generator=g)
    view = x.to(dev)[:, 3:]
    img = np.ascontiguousarray(x[:, 3:].numpy())
    for bilinear in (True, False):
        out = torch.full((B, 3, H, W), float("nan"), device=dev)
        assert resample2d_cuda.forward(view, flow.to(dev), out, 1, bilinear) == 1
        assert max_abs(out.cpu().numpy(), oracle.resample_fwd(img, flow.numpy(), 1, bilinear)) <= 1e-6
    gimg = torch.ones(B, 3, H, W, device=dev)
    gflow = torch.full((B, 2, H, W), float("nan"), device=dev)
    assert resample2d_cuda.backward(view, flow.to(dev), gout.to(dev), gimg, gflow, 1, True) == 1
    rimg, rflow = oracle.resample_bwd(img, flow.numpy(), gout.numpy(), 1, True)
    assert max_abs(gflow.cpu().numpy(), rflow) <= 1e-6
    # fp32 sums in an order of their own; a border cell of the 300-px translation collects thousands of terms
    assert max_abs(gimg.cpu().numpy() - 1.0, rimg) <= 5e-6 * max(1.0, float(np.abs(rimg).max()))


@pytest.mark.parametrize("C", [1, 2, 4])
def test_resample2d_backward_window_follows_the_flow(dev, oracle, C):
    """The per-channel backward kernel (C != 3) with translated flows: same results as the oracle wherever the window lands,
    including windows that leave the image on every side."""
    import resample2d_cuda
    B, H, W = 2, 70, 132
    for i, (sx, sy) in enumerate(((30.0, 0.0), (0.0, -22.0), (-57.0, 49.0), (200.0, -200.0))):
        g = torch.Generator().manual_seed(40 + i)
        img = torch.rand(B, C, H, W, generator=g) - 0.5
        flow = _flow(g, (B, 2, H, W), 2.0)
        flow[:, 0] += sx; flow[:, 1] += sy
        gout = torch.randn(B, C, H, W, generator=g)
        gimg, gflow = torch.zeros(B, C, H, W, device=dev), torch.zeros(B, 2, H, W, device=dev)
        assert resample2d_cuda.backward(img.to(dev), flow.to(dev), gout.to(dev), gimg, gflow, 1, True) == 1
        rimg, rflow = oracle.resample_bwd(img.numpy(), flow.numpy(), gout.numpy(), 1, True)
        assert max_abs(gflow.cpu().numpy(), rflow) <= 1e-6
        # a corner cell of the last two fields collects thousands of terms: the oracle's own sequential fp32 sum is that far from fp64
        assert max_abs(gimg.cpu().numpy(), rimg) <= 2e-5 * max(1.0, float(np.abs(rimg).max()))


def test_resample2d_strided_image_and_accumulate(dev, oracle):
    """The kernel reads input1 through its strides (models.py:133 passes x[:, 3:, :, :]) and the
    backward ACCUMULATES into grad_input1 (caller zero-fills, resample2d.py:31)."""
    import resample2d_cuda
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 6, 12, 16, generator=g)
    flow = _flow(g, (2, 2, 12, 16), 2.0)
    xd = x.to(dev)
    view = xd[:, 3:]
    assert not view.is_contiguous()
    out = torch.zeros(2, 3, 12, 16, device=dev)
    resample2d_cuda.forward(view, flow.to(dev), out, 1, True)
    ref = oracle.resample_fwd(x[:, 3:].contiguous().numpy(), flow.numpy())
    assert max_abs(out.cpu().numpy(), ref) <= 1e-6
    gout = torch.randn(2, 3, 12, 16, generator=g)
    gimg = torch.ones(2, 3, 12, 16, device=dev)
    gflow = torch.zeros(2, 2, 12, 16, device=dev)
    resample2d_cuda.backward(view, flow.to(dev), gout.to(dev), gimg, gflow, 1, True)
    rimg, rflow = oracle.resample_bwd(x[:, 3:].contiguous().numpy(), flow.numpy(), gout.numpy())
    assert max_abs(gimg.cpu().numpy() - 1.0, rimg) <= TOL
    assert max_abs(gflow.cpu().numpy(), rflow) <= 1e-6


@pytest.mark.parametrize("shape", [(2, 3, 40, 64), (1, 3, 384, 512), (2, 2, 17, 33), (1, 1, 16, 36), (3, 3, 50, 132)])
@pytest.mark.parametrize("bilinear", [True, False])
def test_warp_diff_norm_cat(dev, oracle, shape, bilinear):
    """the fused warp -> difference -> channel norm -> concat pass (models.py:133-138) against the
    composition of the oracle's resample and channelnorm restatements, and bitwise against the unfused HIP layers."""
    import fn2_capi
    import resample2d_cuda  # noqa: F401
    from networks.resample2d_package.resample2d import Resample2d, WarpDiffNormCat
    from networks.channelnorm_package.channelnorm import ChannelNorm
    B, C, H, W = shape
    rng = np.random.default_rng(B + C + H + W)
    x = rng.standard_normal((B, 2 * C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((B, 2, H, W)) * 4.0).astype(np.float32)
    flow.reshape(-1)[rng.integers(0, flow.size, flow.size // 100)] *= 20.0   # border clamps, far-out samples
    xd, fd = to_dev(x, dev), to_dev(flow, dev)
    got = fn2_capi.warp_diff_norm_cat(xd, fd, 20.0, bilinear)
    assert got.shape == (B, 3 * C + 3, H, W)
    g = got.cpu().numpy()
    warped = oracle.resample_fwd(np.ascontiguousarray(x[:, C:]), flow, 1, bilinear)
    diff = x[:, :C] - warped
    norm = oracle.chnorm_fwd(diff)
    scaled = flow * (np.float32(1.0) / np.float32(20.0))   # a GPU tensor / scalar in PyTorch = times the fp32 reciprocal
    ref = np.concatenate((x, warped, scaled, norm), axis=1)
    assert np.array_equal(g[:, :2 * C], x)
    assert np.array_equal(g[:, 3 * C:3 * C + 2], scaled)
    assert max_abs(g, ref) <= TOL
    # bit-identical to the unfused HIP layers in the reference model's statement order
    res = Resample2d(1, bilinear)(xd[:, C:].contiguous(), fd)
    unf = torch.cat((xd, res, fd / 20.0, ChannelNorm()(xd[:, :C] - res)), dim=1)
    assert torch.equal(got, unf)
    assert torch.equal(WarpDiffNormCat(20.0, bilinear)(xd, fd), got)


@pytest.mark.parametrize("shape", [(8, 384, 512), (1, 384, 512), (3, 64, 128), (12, 48, 96), (2, 100, 200), (1, 16, 32), (2, 33, 68)])
@pytest.mark.parametrize("kind", ["noise", "translation", "all_far"])
def test_resample2d_backward_c3_batches_and_flows(dev, oracle, shape, kind):
    """The three-channel backward (`resample_bwd_c3x`, round 5: 64-byte aligned windows) over batch sizes 1, 2, 3, 8, 12 (an XCD owns
    one image only at 8), far pixels as scattered atomics (noise), none (translation: the window follows the flow), nothing but far
    pixels (all_far); through the autograd Function on a strided channel slice (models.py:133).  Against the oracle where it
    finishes in seconds, always against a second run (grad_flow bit-repeatable, grad_input1 to the order of the fp32 atomics)."""
    from networks.resample2d_package.resample2d import Resample2dFunction
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + H + W + len(kind))
    x0 = torch.rand(B, 6, H, W, generator=g) - 0.5
    if kind == "noise":
        flow = _flow(g, (B, 2, H, W), 4.0)
    elif kind == "translation":
        flow = torch.randn(B, 2, H, W, generator=g) * 1.5
        flow[:, 0] += 27.0; flow[:, 1] -= 13.0
    else:
        flow = torch.randn(B, 2, H, W, generator=g) * 60.0
    gout = torch.randn(B, 3, H, W, generator=g).to(dev)
    runs = []
    for _ in range(2):
        x, f = x0.to(dev).requires_grad_(True), flow.to(dev).requires_grad_(True)
        Resample2dFunction.apply(x[:, 3:], f, 1, True).backward(gout)
        runs.append((x.grad, f.grad))
    assert torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][0][:, :3].abs().max()) == 0.0
    scale = max(1.0, float(runs[0][0].abs().max()))
    assert float((runs[0][0] - runs[1][0]).abs().max()) <= 5e-6 * scale
    if B * H * W <= 3 * 64 * 128:
        rimg, rflow = oracle.resample_bwd(np.ascontiguousarray(x0.numpy()[:, 3:]), flow.numpy(), gout.cpu().numpy(), 1, True)
        assert max_abs(runs[0][1].cpu().numpy(), rflow) <= 1e-6
        assert max_abs(runs[0][0][:, 3:].cpu().numpy(), rimg) <= 2e-5 * max(1.0, float(np.abs(rimg).max()))


@pytest.mark.parametrize("shape", [(8, 3, 384, 512), (2, 3, 40, 64), (3, 3, 50, 132), (2, 2, 17, 33), (1, 1, 16, 36), (1, 3, 100, 200)])
@pytest.mark.parametrize("bilinear", [True, False])
def test_warp_diff_norm_cat_backward(dev, oracle, shape, bilinear):
    """VERDICT r4 next #4 (row N2, training half): the one-kernel backward of models.py:133-138 against autograd through the
    UNFUSED HIP layers (Resample2d, ChannelNorm, cat, the division) on the same inputs -- grad_flow bit-identical, the pair's
    gradient to the order of the fp32 atomics --, with and without the pair's gradient (without: gather only, no atomics), through
    the ctypes entry point and through the differentiable module; shapes the tiled kernel does not take (C != 3, ragged widths) use
    the one-lane-per-pixel kernel.  Also against the oracle's composition of the reference kernels (every item of the small shapes, one item of the BASELINE shape)."""
    import fn2_capi
    from networks.resample2d_package.resample2d import Resample2d, WarpDiffNormCat
    from networks.channelnorm_package.channelnorm import ChannelNorm
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B + C + H + W)
    x0 = torch.randn(B, 2 * C, H, W, generator=g)
    f0 = _flow(g, (B, 2, H, W), 4.0)
    gcat = torch.randn(B, 3 * C + 3, H, W, generator=g).to(dev)
    # reference: autograd through the unfused HIP layers, the reference model's statements
    x, f = x0.to(dev).requires_grad_(True), f0.to(dev).requires_grad_(True)
    res = Resample2d(1, bilinear)(x[:, C:], f)
    unf = torch.cat((x, res, f / 20.0, ChannelNorm()(x[:, :C] - res)), dim=1)
    unf.backward(gcat)
    # fused module
    x2, f2 = x0.to(dev).requires_grad_(True), f0.to(dev).requires_grad_(True)
    out = WarpDiffNormCat(20.0, bilinear)(x2, f2)
    assert torch.equal(out, unf.detach())
    out.backward(gcat)
    assert torch.equal(f2.grad, f.grad), float((f2.grad - f.grad).abs().max())
    sx = max(1.0, float(x.grad.abs().max()))
    assert torch.equal(x2.grad[:, :C], x.grad[:, :C])
    assert float((x2.grad[:, C:] - x.grad[:, C:]).abs().max()) <= 5e-6 * sx
    # the pair without gradient (FlowNet2's case): gather only
    x3, f3 = x0.to(dev), f0.to(dev).requires_grad_(True)
    WarpDiffNormCat(20.0, bilinear)(x3, f3).backward(gcat)
    assert torch.equal(f3.grad, f.grad)
    # the C ABI entry point through ctypes (outputs start as NaN: everything must be written)
    gp, gf = fn2_capi.warp_diff_norm_cat_backward(x3, f0.to(dev), out.detach(), gcat, 20.0, bilinear, True)
    assert torch.equal(gf, f.grad) and torch.equal(gp[:, :C], x.grad[:, :C])
    assert float((gp[:, C:] - x.grad[:, C:]).abs().max()) <= 5e-6 * sx
    gp, gf = fn2_capi.warp_diff_norm_cat_backward(x3, f0.to(dev), out.detach(), gcat, 20.0, bilinear, False)
    assert gp is None and torch.equal(gf, f.grad)
    # oracle composition: channelnorm_kernel.cu:63-96 then resample2d_kernel.cu:75-198 (both ignore `bilinear`) -- on every batch item of
    # the small shapes, on the first item of the BASELINE shape (round 6: the fused backward at 8 x 3 x 384 x 512 is no longer checked
    # against the unfused HIP layers only; the layers have no cross-item dependence)
    ns = slice(0, B) if B * C * H * W <= 3 * 3 * 50 * 132 else slice(0, 1)
    xn, fn, gn = x0.numpy()[ns], f0.numpy()[ns], gcat.cpu().numpy()[ns]
    warped = oracle.resample_fwd(np.ascontiguousarray(xn[:, C:]), fn, 1, bilinear)
    diff = xn[:, :C] - warped
    nrm = oracle.chnorm_fwd(diff)
    gdiff = oracle.chnorm_bwd(diff, nrm, np.ascontiguousarray(gn[:, 3 * C + 2:3 * C + 3]))
    gw = gn[:, 2 * C:3 * C] - gdiff
    rimg, rflow = oracle.resample_bwd(np.ascontiguousarray(xn[:, C:]), fn, np.ascontiguousarray(gw), 1, True)
    rflow = rflow + gn[:, 3 * C:3 * C + 2] * (np.float32(1.0) / np.float32(20.0))
    assert max_abs(f2.grad[ns].cpu().numpy(), rflow) <= 1e-5 * max(1.0, float(np.abs(rflow).max()))
    assert max_abs(x2.grad[ns, :C].cpu().numpy(), gn[:, :C] + gdiff) <= 1e-5 * max(1.0, float(np.abs(gdiff).max()))
    assert max_abs(x2.grad[ns, C:].cpu().numpy(), gn[:, C:2 * C] + rimg) <= 2e-5 * max(1.0, float(np.abs(rimg).max()))


@pytest.mark.parametrize("shape", [(8, 3, 384, 512), (2, 3, 40, 64), (3, 3, 50, 132), (2, 2, 17, 33), (1, 1, 16, 36), (1, 3, 100, 200)])
@pytest.mark.parametrize("bilinear", [True, False])
def test_warp_diff_norm(dev, oracle, shape, bilinear):
    """Row N2 without the concat (models.py:157-161, :170-174): ||first image - warped second image|| as one kernel, and its flow
    gradient as one gather-only kernel that RECOMPUTES the warp (the forward stores only the norm) -- both bit-identical to the unfused
    HIP layers under autograd (Resample2d, subtraction, ChannelNorm), bilinear and nearest sampling, tiled and untiled shapes; a
    translated flow (windows that follow it); on a small case against the oracle's composition of the reference kernels."""
    import fn2_capi
    from networks.resample2d_package.resample2d import Resample2d, WarpDiffNorm
    from networks.channelnorm_package.channelnorm import ChannelNorm
    B, C, H, W = shape
    g = torch.Generator().manual_seed(7 * B + C + H + W)
    x0 = torch.randn(B, 2 * C, H, W, generator=g)
    for shift in (0.0, 21.0):
        f0 = _flow(g, (B, 2, H, W), 4.0)
        f0[:, 0] += shift; f0[:, 1] -= 0.6 * shift
        gn = torch.randn(B, 1, H, W, generator=g).to(dev)
        x, f = x0.to(dev), f0.to(dev).requires_grad_(True)
        res = Resample2d(1, bilinear)(x[:, C:], f)
        unf = ChannelNorm()(x[:, :C] - res)
        unf.backward(gn)
        f2 = f0.to(dev).requires_grad_(True)
        out = WarpDiffNorm(bilinear)(x, f2)
        assert torch.equal(out, unf.detach())
        out.backward(gn)
        assert torch.equal(f2.grad, f.grad), float((f2.grad - f.grad).abs().max())
        # the C ABI through ctypes (outputs start as NaN)
        n2 = fn2_capi.warp_diff_norm(x, f0.to(dev), bilinear)
        assert torch.equal(n2, unf.detach())
        assert torch.equal(fn2_capi.warp_diff_norm_backward(x, f0.to(dev), n2, gn, bilinear), f.grad)
    # a pair that wants a gradient takes the unfused layers under autograd: same values
    xg, fg = x0.to(dev).requires_grad_(True), f0.to(dev).requires_grad_(True)
    WarpDiffNorm(bilinear)(xg, fg).backward(gn)
    assert torch.equal(fg.grad, f.grad) and xg.grad is not None and float(xg.grad.abs().max()) > 0
    if B * C * H * W <= 3 * 3 * 50 * 132:
        xn, fn, gnn = x0.numpy(), f0.numpy(), gn.cpu().numpy()
        warped = oracle.resample_fwd(np.ascontiguousarray(xn[:, C:]), fn, 1, bilinear)
        diff = xn[:, :C] - warped
        nrm = oracle.chnorm_fwd(diff)
        assert max_abs(out.detach().cpu().numpy(), nrm) <= 1e-6 * max(1.0, float(nrm.max()))
        gdiff = oracle.chnorm_bwd(diff, nrm, gnn)
        _, rflow = oracle.resample_bwd(np.ascontiguousarray(xn[:, C:]), fn, np.ascontiguousarray(-gdiff), 1, True)
        assert max_abs(f2.grad.cpu().numpy(), rflow) <= 1e-5 * max(1.0, float(np.abs(rflow).max()))


def test_resample2d_rejects_non_float(dev):
    import resample2d_cuda
    a = torch.zeros(1, 3, 8, 8, device=dev, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="float32"):
        resample2d_cuda.forward(a, torch.zeros(1, 2, 8, 8, device=dev, dtype=torch.float64), a.clone(), 1, True)


# ------------------------------------------------------------------ correlation
def _corr_both(dev, in1, in2, gout, params):
    import correlation_cuda
    a, b = to_dev(in1, dev), to_dev(in2, dev)
    e = lambda: a.new_empty(0)  # noqa: E731
    out = e()
    assert correlation_cuda.forward(a, b, e(), e(), out, *params, 1) == 1
    res = [out]
    if gout is not None:
        g1, g2 = e(), e()
        assert correlation_cuda.backward(a, b, e(), e(), to_dev(gout, dev), g1, g2, *params, 1) == 1
        res += [g1, g2]
    torch.cuda.synchronize()
    return [r.cpu().numpy() for r in res]


@pytest.mark.parametrize("path", golden_files("corr"), ids=os.path.basename)
def test_correlation_golden(dev, path):
    g = np.load(path)
    params = tuple(int(v) for v in g["params"])
    has_bwd = "g1" in g.files
    res = _corr_both(dev, g["in1"], g["in2"], g["gout"] if has_bwd else None, params)
    assert res[0].shape == g["out"].shape
    assert max_abs(res[0], g["out"]) <= TOL
    if has_bwd:
        assert max_abs(res[1], g["g1"]) <= TOL and max_abs(res[2], g["g2"]) <= TOL


@pytest.mark.parametrize("path", golden_files("corrgraded"), ids=os.path.basename)
def test_correlation_golden_graded_geometry(dev, path):
    """the graded f16x2 kernels on their real 48 x 64 task table -- every (row group, neighbour row block) task,
    ragged displacement ranges at every border -- against the REFERENCE's own device code (tests/golden/make_golden_corr_graded.py:
    correlation_cuda_kernel.cu under the CPU SIMT shim).  The fixture keeps 32 displacement planes / 8 gradient channels in full and
    float64 sums of all 441 planes / 64 channels; the inputs are regenerated from its seed."""
    g = np.load(path)
    in1, in2, gout = graded_corr_inputs(g)
    params = tuple(int(v) for v in g["params"])
    out, g1, g2 = (np.asarray(t) if isinstance(t, np.ndarray) else t.cpu().numpy() for t in _corr_both(dev, in1, in2, gout, params))
    assert max_abs(out[:, g["planes"]], g["out_planes"]) <= 1e-5          # fp32-class sums of 64 products of N(0,1) values (one channel x30)
    o64 = out.astype(np.float64)
    n = out.shape[2] * out.shape[3]
    assert np.max(np.abs(o64.sum(axis=(0, 2, 3)) - g["out_sum"])) <= 1e-5 * n ** 0.5 * 4
    assert np.max(np.abs((o64 * o64).sum(axis=(0, 2, 3)) - g["out_sumsq"]) / np.maximum(g["out_sumsq"], 1e-30)) <= 1e-5
    for got, full, ssum, sabs in ((g1, g["g1_channels"], g["g1_sum"], g["g1_abs"]), (g2, g["g2_channels"], g["g2_sum"], g["g2_abs"])):
        ref_scale = np.abs(full).max(axis=(0, 2, 3), keepdims=True)        # per channel: the two rescaled channels keep their own bar
        assert np.max(np.abs(got[:, g["channels"]] - full) / np.maximum(ref_scale, 1e-30)) <= 2e-6
        assert np.max(np.abs(got.astype(np.float64).sum(axis=(0, 2, 3)) - ssum) / np.maximum(sabs, 1e-30)) <= 1e-6


MFMA_CASES = [  # B, C, H, W, md  (k=1, s1=1, s2=2, pad=md)
    (2, 64, 12, 16, 20), (1, 32, 10, 40, 12), (1, 96, 14, 20, 8), (1, 64, 6, 64, 16), (2, 32, 48, 64, 20),
    (1, 32, 6, 8, 20), (2, 16, 8, 8, 20), (1, 64, 16, 24, 20), (1, 16, 10, 40, 20), (1, 16, 8, 130, 20),
    (1, 32, 12, 16, 4), (2, 16, 10, 12, 6), (1, 16, 6, 6, 2), (1, 48, 14, 18, 10), (1, 16, 20, 72, 14),
    (1, 16, 8, 8, 21),
]


@pytest.mark.parametrize("case", MFMA_CASES)
def test_correlation_mfma_vs_oracle(dev, oracle, case):
    import fn2_capi
    B, C, H, W, md = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W + md)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ad, bd = to_dev(a, dev), to_dev(b, dev)
    out = torch.full((B, (2 * (md // 2) + 1) ** 2, H, W), float("nan"), device=dev)   # every element must be written
    fn2_capi.correlation_forward(ad, bd, md, 1, md, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32, out=out)
    direct = fn2_capi.correlation_forward(ad, bd, md, 1, md, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    ref = oracle.corr_fwd(a, b, md, 1, md, 1, 2)
    o = out.cpu().numpy()
    assert np.isfinite(o).all(), "MFMA kernel left output elements unwritten"
    assert max_abs(o, ref) <= TOL
    assert max_abs(direct.cpu().numpy(), ref) <= TOL
    assert max_abs(o, ref) <= 2e-6, "fp32 MFMA chain should agree with the fp32 oracle to rounding"
    # the auto path (bf16x3 exact-split kernel where it applies) and the explicit bf16x3 selector
    auto = torch.full_like(out, float("nan"))
    fn2_capi.correlation_forward(ad, bd, md, 1, md, 1, 2, algo=fn2_capi.FN2_CORR_AUTO, out=auto)
    assert np.isfinite(auto.cpu().numpy()).all()
    assert max_abs(auto.cpu().numpy(), ref) <= 2e-6
    if W <= 64 and C % 32 == 0 and (md // 2) % 2 == 0 and W % 4 == 0:
        b3 = torch.full_like(out, float("nan"))
        fn2_capi.correlation_forward(ad, bd, md, 1, md, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_BF16X3, out=b3)
        assert np.isfinite(b3.cpu().numpy()).all() and max_abs(b3.cpu().numpy(), ref) <= 2e-6


@pytest.mark.parametrize("case", [(2, 64, 12, 16, 20, torch.float32), (1, 32, 10, 40, 12, torch.float32),
                                  (8, 256, 48, 64, 20, torch.float32), (1, 16, 8, 130, 20, torch.float32),
                                  (1, 8, 9, 7, 4, torch.float16), (2, 4, 6, 6, 2, torch.float64)])
def test_correlation_fused_leakyrelu_cat(dev, oracle, case):
    """LeakyReLU(0.1)(correlation) written into the channel slice of the conv_redir concat buffer
    (FlowNetC.py:86-92) -- C ABI entry, pybind function and Module; the other channels of the buffer stay untouched."""
    import fn2_capi
    import correlation_cuda
    from networks.correlation_package.correlation import CorrelationLeakyReLUCat
    B, C, H, W, md, dtype = case
    npd = {torch.float32: np.float32, torch.float16: np.float16, torch.float64: np.float64}[dtype]
    rng = np.random.default_rng(C + H + W + md)
    a = rng.standard_normal((B, C, H, W)).astype(npd)
    b = rng.standard_normal((B, C, H, W)).astype(npd)
    if B == 8:   # full size: the oracle on two batch items only
        items = (0, 7)
    else:
        items = tuple(range(B))
    D2 = (2 * (md // 2) + 1) ** 2
    Cr = 32
    redir = rng.standard_normal((B, Cr, H, W)).astype(npd)
    ad, bd, rd = to_dev(a, dev), to_dev(b, dev), to_dev(redir, dev)
    sentinel = 7.0
    buf = torch.full((B, Cr + D2 + 3, H, W), sentinel, device=dev, dtype=dtype)
    fn2_capi.correlation_forward_fused(ad, bd, buf, Cr, 0.1, md, 1, md, 1, 2)
    got = buf.cpu().numpy()
    assert (got[:, :Cr] == sentinel).all() and (got[:, Cr + D2:] == sentinel).all(), "wrote outside its channel slice"
    tol = TOL if dtype != torch.float16 else 2e-3
    for n in items:
        ref = oracle.corr_fwd(a[n:n + 1].astype(np.float32 if dtype != torch.float64 else np.float64),
                              b[n:n + 1].astype(np.float32 if dtype != torch.float64 else np.float64), md, 1, md, 1, 2)
        ref = np.where(ref > 0, ref, ref * npd(0.1).astype(ref.dtype))
        assert max_abs(got[n:n + 1, Cr:Cr + D2].astype(np.float64), ref.astype(np.float64)) <= tol
    # pybind entry and Module: cat((redir, leaky(corr)), 1)
    buf2 = torch.empty((B, Cr + D2, H, W), device=dev, dtype=dtype)
    buf2[:, :Cr] = rd
    correlation_cuda.forward_fused(ad, bd, buf2, Cr, 0.1, md, 1, md, 1, 2)
    assert torch.equal(buf2[:, Cr:], buf[:, Cr:Cr + D2])
    mod = CorrelationLeakyReLUCat(md, 1, md, 1, 2, negative_slope=0.1)
    cat = mod(ad, bd, rd)
    assert torch.equal(cat, buf2)
    # against the unfused sequence of the reference model on the same kernels
    from networks.correlation_package.correlation import Correlation
    unfused = torch.cat((rd, torch.nn.functional.leaky_relu(Correlation(md, 1, md, 1, 2, 1)(ad, bd), 0.1)), 1)
    assert torch.equal(cat, unfused) or float((cat - unfused).abs().max()) <= (1e-7 if dtype != torch.float16 else 1e-3)


BWD_CASES = [  # B, C, H, W, md  (k=1, s1=1, s2=2, pad=md); C % 32 == 0
    (1, 32, 6, 8, 20), (2, 64, 8, 8, 20), (1, 64, 16, 24, 20), (1, 32, 10, 40, 20), (1, 32, 8, 130, 20),
    (1, 32, 12, 16, 4), (2, 64, 10, 12, 6), (1, 32, 6, 6, 2), (1, 96, 14, 18, 10), (1, 32, 20, 72, 14), (1, 128, 8, 8, 21),
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_correlation_mfma_backward_vs_oracle(dev, oracle, case):
    import fn2_capi
    B, C, H, W, md = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W + md + 1)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    D = 2 * (md // 2) + 1
    go = rng.standard_normal((B, D * D, H, W)).astype(np.float32)
    ad, bd, gd = to_dev(a, dev), to_dev(b, dev), to_dev(go, dev)
    r1, r2 = oracle.corr_bwd(a, b, go, md, 1, md, 1, 2)
    for algo in (fn2_capi.FN2_CORR_MFMA_F32, 101, fn2_capi.FN2_CORR_DIRECT):   # 101: 32-channel groups
        g1, g2 = fn2_capi.correlation_backward(ad, bd, gd, md, 1, md, 1, 2, algo=algo)
        e1, e2 = max_abs(g1.cpu().numpy(), r1), max_abs(g2.cpu().numpy(), r2)
        assert e1 <= TOL and e2 <= TOL, (algo, e1, e2)
        scale = max(1.0, float(np.abs(r1).max()))
        assert e1 <= 5e-6 * scale and e2 <= 5e-6 * scale, (algo, e1, e2)   # fp32 sums in a different order only


BWD_BF16_CASES = [  # B, C, H, W, md: the bf16x3 backward kernel's domain (radius 10, C % 64 == 0, W % 4 == 0)
    (1, 64, 6, 8, 20), (2, 64, 8, 8, 20), (1, 64, 16, 24, 20), (1, 64, 10, 40, 20), (1, 64, 8, 132, 20),
    (1, 128, 8, 8, 21), (1, 64, 14, 36, 20), (1, 64, 46, 64, 20), (3, 64, 22, 68, 21),
]


@pytest.mark.parametrize("case", BWD_BF16_CASES)
def test_correlation_bf16x3_backward_vs_oracle(dev, oracle, case):
    """The exact-split bf16 backward kernel (32-px tiles = shipped, 64-px tiles = algo 105) and the automatic path,
    against the oracle; every gradient element must be written."""
    import fn2_capi
    B, C, H, W, md = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W + md + 2)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    go = rng.standard_normal((B, 441, H, W)).astype(np.float32)
    ad, bd, gd = to_dev(a, dev), to_dev(b, dev), to_dev(go, dev)
    r1, r2 = oracle.corr_bwd(a, b, go, md, 1, md, 1, 2)
    scale = max(1.0, float(np.abs(r1).max()))
    for algo in (fn2_capi.FN2_CORR_MFMA_BF16X3, 105, fn2_capi.FN2_CORR_AUTO):
        g1 = torch.full((B, C, H, W), float("nan"), device=dev)
        g2 = torch.full((B, C, H, W), float("nan"), device=dev)
        fn2_capi.correlation_backward(ad, bd, gd, md, 1, md, 1, 2, algo=algo, out=(g1, g2))
        n1, n2 = g1.cpu().numpy(), g2.cpu().numpy()
        assert np.isfinite(n1).all() and np.isfinite(n2).all(), "unwritten gradient elements"
        e1, e2 = max_abs(n1, r1), max_abs(n2, r2)
        assert e1 <= TOL and e2 <= TOL, (algo, e1, e2)
        assert e1 <= 5e-6 * scale and e2 <= 5e-6 * scale, (algo, e1, e2)


BWD_F16X2_CASES = [  # B, C, H, W: the f16x2 backward kernel's domain (md = 20, C % 64 == 0, W % 8 == 0 <= 64)
    (1, 64, 6, 8), (2, 64, 8, 8), (1, 64, 16, 24), (1, 64, 10, 40), (1, 128, 22, 56), (1, 64, 46, 64), (3, 64, 2, 16), (1, 192, 12, 32),
]


@pytest.mark.parametrize("case", BWD_F16X2_CASES)
def test_correlation_f16x2_backward_vs_oracle(dev, oracle, case):
    """The f16x2 backward kernel (what FN2_CORR_AUTO picks for FlowNetC's cost volume) against the oracle; every gradient
    element written; ragged lattices and partial widths."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W + 3)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    go = rng.standard_normal((B, 441, H, W)).astype(np.float32)
    ad, bd, gd = to_dev(a, dev), to_dev(b, dev), to_dev(go, dev)
    r1, r2 = oracle.corr_bwd(a, b, go, 20, 1, 20, 1, 2)
    scale = max(1.0, float(np.abs(r1).max()))
    res = {}
    for algo in (fn2_capi.FN2_CORR_MFMA_F16X2, fn2_capi.FN2_CORR_AUTO):
        g1 = torch.full((B, C, H, W), float("nan"), device=dev)
        g2 = torch.full((B, C, H, W), float("nan"), device=dev)
        fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=algo, out=(g1, g2))
        n1, n2 = g1.cpu().numpy(), g2.cpu().numpy()
        assert np.isfinite(n1).all() and np.isfinite(n2).all(), "unwritten gradient elements"
        e1, e2 = max_abs(n1, r1), max_abs(n2, r2)
        assert e1 <= TOL and e2 <= TOL, (algo, e1, e2)
        assert e1 <= 5e-6 * scale and e2 <= 5e-6 * scale, (algo, e1, e2)
        res[algo] = (g1, g2)
    assert torch.equal(res[fn2_capi.FN2_CORR_AUTO][0], res[fn2_capi.FN2_CORR_MFMA_F16X2][0]), "AUTO should select f16x2 here"
    assert torch.equal(res[fn2_capi.FN2_CORR_AUTO][1], res[fn2_capi.FN2_CORR_MFMA_F16X2][1])


BWD_F16X2_WIDE_CASES = [  # W > 64: centre windows of 64 pixels, two neighbour passes (csrc/correlation_f16x2_bwd_wide.hip)
    (1, 64, 6, 72), (2, 64, 8, 96), (1, 64, 10, 104), (1, 128, 12, 128), (1, 64, 4, 200), (1, 64, 28, 136), (2, 64, 2, 80),
]


@pytest.mark.parametrize("case", BWD_F16X2_WIDE_CASES)
def test_correlation_f16x2_backward_wide_vs_oracle(dev, oracle, case):
    """Maps wider than 64 pixels on the windowed f16x2 backward kernel: every gradient element written, fp32-rounding close to
    the oracle, selected by FN2_CORR_AUTO; tiny gradOutput (training magnitudes) keeps its relative accuracy; operands that do
    not fit an f16 are recomputed in fp32."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W + 5)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    go = rng.standard_normal((B, 441, H, W)).astype(np.float32)
    ad, bd, gd = to_dev(a, dev), to_dev(b, dev), to_dev(go, dev)
    r1, r2 = oracle.corr_bwd(a, b, go, 20, 1, 20, 1, 2)
    scale = max(1.0, float(np.abs(r1).max()))
    res = {}
    for algo in (fn2_capi.FN2_CORR_MFMA_F16X2, fn2_capi.FN2_CORR_AUTO):
        g1 = torch.full((B, C, H, W), float("nan"), device=dev)
        g2 = torch.full((B, C, H, W), float("nan"), device=dev)
        fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=algo, out=(g1, g2))
        n1, n2 = g1.cpu().numpy(), g2.cpu().numpy()
        assert np.isfinite(n1).all() and np.isfinite(n2).all(), "unwritten gradient elements"
        e1, e2 = max_abs(n1, r1), max_abs(n2, r2)
        assert e1 <= 5e-6 * scale and e2 <= 5e-6 * scale, (algo, e1, e2)
        res[algo] = (g1, g2)
    assert torch.equal(res[fn2_capi.FN2_CORR_AUTO][0], res[fn2_capi.FN2_CORR_MFMA_F16X2][0]), "AUTO should select f16x2 here"
    assert torch.equal(res[fn2_capi.FN2_CORR_AUTO][1], res[fn2_capi.FN2_CORR_MFMA_F16X2][1])
    g1, g2 = res[fn2_capi.FN2_CORR_MFMA_F16X2]
    sg = 2.0 ** -23
    t1, t2 = fn2_capi.correlation_backward(ad, bd, gd * sg, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert float((t1 / sg - g1).abs().max()) <= 1e-5 * scale and float((t2 / sg - g2).abs().max()) <= 1e-5 * scale
    a2, b2, g3 = ad.clone(), bd.clone(), gd.clone()
    a2[0, 3, H // 2, W - 5] = 1.0e6; b2[0, 7, 1, 70] = -3.0e7; g3[0, 200, H - 1, 66] = 3.0e6; g3[B - 1, 220, 0, 3] = float("inf")
    q1, q2 = fn2_capi.correlation_backward(a2, b2, g3, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    f1, f2 = fn2_capi.correlation_backward(a2, b2, g3, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    for got, ref in ((q1, f1), (q2, f2)):
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), "non-finite gradient elements must coincide"
        assert int((~fin).sum()) > 0
        assert float((got[fin].double() - ref[fin].double()).abs().max()) <= 2e-6 * float(ref[fin].abs().max())


def test_correlation_f16x2_backward_out_of_range_operands(dev):
    """gradOutput / input values that do not fit an f16, infinities and NaNs: the affected gradient elements are
    recomputed in plain fp32 -- same finite/non-finite pattern as the general kernel, same values where finite."""
    import fn2_capi
    g = torch.Generator().manual_seed(6)
    a = torch.randn(2, 64, 16, 24, generator=g)
    b = torch.randn(2, 64, 16, 24, generator=g)
    go = torch.randn(2, 441, 16, 24, generator=g)
    a[0, 3, 5, 7] = 1.0e5; b[0, 7, 9, 9] = -7.0e4; go[0, 200, 4, 4] = 3.0e6; go[0, 17, 12, 20] = -65520.0
    a[1, 5, 4, 4] = float("inf"); b[1, 9, 11, 3] = float("nan"); go[1, 300, 8, 8] = float("inf")
    ad, bd, gd = a.to(dev), b.to(dev), go.to(dev)
    g1, g2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    r1, r2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    for got, ref in ((g1, r1), (g2, r2)):
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(got), fin), "non-finite gradient elements must coincide"
        assert int(fin.sum()) > 0 and int((~fin).sum()) > 0
        err = (got[fin].double() - ref[fin].double()).abs()
        assert float(err.max()) <= 2e-6 * float(ref[fin].abs().max()), float(err.max())


def test_correlation_bf16x3_backward_rejects_outside_domain(dev):
    import fn2_capi
    x = torch.randn(1, 32, 8, 8, device=dev)          # C % 64 != 0
    go = torch.randn(1, 441, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        fn2_capi.correlation_backward(x, x, go, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_BF16X3)
    x = torch.randn(1, 64, 8, 8, device=dev)          # radius 6
    go = torch.randn(1, 169, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        fn2_capi.correlation_backward(x, x, go, 12, 1, 12, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_BF16X3)
    g1, g2 = fn2_capi.correlation_backward(x, x, go, 12, 1, 12, 1, 2)   # automatic: falls back to the fp32 MFMA kernel
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all()


def test_correlation_backward_accuracy_vs_fp64(dev):
    """Backward at the BASELINE size: against an fp64 reference (autograd through an fp64 correlation) the bf16x3 kernel
    is as accurate as the fp32 MFMA kernel; plus linearity in gradOutput and batch independence."""
    import fn2_capi
    import torch.nn.functional as F
    B, C, H, W = 2, 256, 48, 64
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(B, C, H, W, generator=g).to(dev)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    a64, b64 = x1.double().requires_grad_(True), x2.double().requires_grad_(True)
    p2 = F.pad(b64, (20, 20, 20, 20))
    loss = 0.0
    for tj in range(21):
        for ti in range(21):
            o = (a64 * p2[:, :, 2 * tj:2 * tj + H, 2 * ti:2 * ti + W]).mean(1)
            loss = loss + (o * go[:, tj * 21 + ti].double()).sum()
    loss.backward()
    r1, r2 = a64.grad, b64.grad
    errs = {}
    for name, algo in (("f32", fn2_capi.FN2_CORR_MFMA_F32), ("bf16x3", fn2_capi.FN2_CORR_MFMA_BF16X3),
                       ("f16x2", fn2_capi.FN2_CORR_MFMA_F16X2)):
        g1, g2 = fn2_capi.correlation_backward(x1, x2, go, 20, 1, 20, 1, 2, algo=algo)
        errs[name] = max(float((g1.double() - r1).abs().max()), float((g2.double() - r2).abs().max()))
    tol = 3e-6 * float(r1.abs().max())
    assert errs["f32"] <= tol and errs["bf16x3"] <= tol and errs["f16x2"] <= tol, (errs, tol)
    assert errs["bf16x3"] <= 3.0 * errs["f32"] and errs["f16x2"] <= 3.0 * errs["f32"], errs
    # linearity in gradOutput, batch independence (automatic path)
    go2 = torch.randn(B, 441, H, W, generator=g).to(dev)
    s1, s2 = fn2_capi.correlation_backward(x1, x2, go + go2, 20, 1, 20, 1, 2)
    p1, p2_ = fn2_capi.correlation_backward(x1, x2, go, 20, 1, 20, 1, 2)
    q1, q2 = fn2_capi.correlation_backward(x1, x2, go2, 20, 1, 20, 1, 2)
    assert float((s1 - p1 - q1).abs().max()) <= 5e-6 and float((s2 - p2_ - q2).abs().max()) <= 5e-6
    t1, t2 = fn2_capi.correlation_backward(x1[1:2].contiguous(), x2[1:2].contiguous(), go[1:2].contiguous(), 20, 1, 20, 1, 2)
    assert torch.equal(t1[0], p1[1]) and torch.equal(t2[0], p2_[1])


@pytest.mark.parametrize("dist", ["normal", "leaky"])
def test_correlation_full_size_vs_oracle(dev, oracle, dist):
    """BASELINE.json configs[1]: fwd+bwd on 8x256x48x64 fp32, <= 1e-4 max-abs (oracle on 2 of the
    8 batch items to keep CPU time bounded; items are independent)."""
    g = torch.Generator().manual_seed(0)
    shape = (8, 256, 48, 64)
    a = torch.randn(shape, generator=g)
    b = torch.randn(shape, generator=g)
    if dist == "leaky":   # mimic conv3 activations (LeakyReLU 0.1)
        a, b = torch.nn.functional.leaky_relu(a, 0.1), torch.nn.functional.leaky_relu(b, 0.1)
    go = torch.randn((8, 441, 48, 64), generator=torch.Generator().manual_seed(1))
    params = (20, 1, 20, 1, 2)
    out, g1, g2 = _corr_both(dev, a.numpy(), b.numpy(), go.numpy(), params)
    for n in (0, 7):
        sl = slice(n, n + 1)
        ref = oracle.corr_fwd(a[sl].numpy(), b[sl].numpy(), *params)
        assert max_abs(out[sl], ref) <= TOL
        r1, r2 = oracle.corr_bwd(a[sl].numpy(), b[sl].numpy(), go[sl].numpy(), *params)
        assert max_abs(g1[sl], r1) <= TOL and max_abs(g2[sl], r2) <= TOL


def test_correlation_bf16x3_accuracy_vs_fp64(dev):
    """The exact-split bf16 kernel is fp32-class: against an fp64 reference its error is no worse than the fp32
    MFMA kernel's, for unit-scale, large, small and wide-dynamic-range inputs (scale-invariant)."""
    import fn2_capi
    import torch.nn.functional as F
    B, C, H, W = 2, 256, 48, 64
    g = torch.Generator().manual_seed(21)

    def ref64(x1, x2):
        p2 = F.pad(x2.double(), (20, 20, 20, 20))
        outs = [(x1.double() * p2[:, :, 20 + 2 * tj:20 + 2 * tj + H, 20 + 2 * ti:20 + 2 * ti + W]).mean(1, keepdim=True)
                for tj in range(-10, 11) for ti in range(-10, 11)]
        return torch.cat(outs, 1)

    for scale, spread in [(1.0, 0.0), (100.0, 0.0), (1e-3, 0.0), (1.0, 3.0)]:
        x1 = (scale * torch.randn(B, C, H, W, generator=g) * torch.exp(spread * torch.randn(B, C, H, W, generator=g))).to(dev)
        x2 = (scale * torch.randn(B, C, H, W, generator=g) * torch.exp(spread * torch.randn(B, C, H, W, generator=g))).to(dev)
        r = ref64(x1, x2)
        e32 = float((fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32).double() - r).abs().max())
        e3 = float((fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_BF16X3).double() - r).abs().max())
        tol = 3e-6 * float(r.abs().max())
        assert e3 <= tol and e32 <= tol, (scale, spread, e3, e32, tol)
        assert e3 <= 3.0 * e32 + 1e-30, (scale, spread, e3, e32)


F16X2_CASES = [   # (B, C, H, W): md = 20 (FlowNetC), W % 8 == 0 <= 64, C % 64 == 0
    (1, 64, 16, 24), (2, 64, 12, 16), (1, 128, 48, 64), (1, 64, 6, 8), (3, 64, 10, 40), (1, 192, 22, 56), (9, 64, 20, 32),
    (2, 64, 2, 8), (1, 64, 92, 64),
]


@pytest.mark.parametrize("case", F16X2_CASES)
def test_correlation_f16x2_vs_oracle(dev, oracle, case):
    """The f16x2 forward kernel (what FN2_CORR_AUTO picks for FlowNetC's cost volume) against the oracle: every output
    element written, <= 1e-4 (the north star's bound), in fact fp32-rounding close; ragged lattices (H/2 not a multiple
    of 4), partial widths and batch counts that do not divide the 8 streams."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ad, bd = to_dev(a, dev), to_dev(b, dev)
    out = torch.full((B, 441, H, W), float("nan"), device=dev)
    fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2, out=out)
    o = out.cpu().numpy()
    assert np.isfinite(o).all(), "f16x2 kernel left output elements unwritten"
    items = range(B) if B * C * H * W <= 2 ** 21 else (0, B - 1)
    for n in items:
        ref = oracle.corr_fwd(a[n:n + 1], b[n:n + 1], 20, 1, 20, 1, 2)
        assert max_abs(o[n:n + 1], ref) <= TOL
        assert max_abs(o[n:n + 1], ref) <= 2e-6, "two-term f16 split should agree with the fp32 oracle to rounding"
    auto = torch.full_like(out, float("nan"))
    fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_AUTO, out=auto)
    assert torch.equal(auto, out), "FN2_CORR_AUTO should select the f16x2 kernel for this configuration"
    direct = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((direct - out).abs().max()) <= 2e-6


F16X2_WIDE_CASES = [   # (B, C, H, W), W > 64: column windows of 32 pixels (csrc/correlation_f16x2_wide.hip)
    (1, 64, 8, 72), (2, 64, 10, 96), (1, 128, 20, 128), (1, 64, 6, 200), (3, 64, 14, 104), (1, 256, 56, 128), (1, 64, 2, 136),
]


@pytest.mark.parametrize("case", F16X2_WIDE_CASES)
def test_correlation_f16x2_wide_vs_oracle(dev, oracle, case):
    """Maps wider than 64 pixels (Sintel-size inputs: conv3 is 56 x 128) on the windowed f16x2 forward kernel: every output
    written, fp32-rounding close to the oracle, selected by FN2_CORR_AUTO, same fused epilogue, same block scaling (tiny
    operands) and the same fp32 recomputation of outputs whose operands do not fit an f16."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C * 7 + H + W)
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ad, bd = to_dev(a, dev), to_dev(b, dev)
    out = torch.full((B, 441, H, W), float("nan"), device=dev)
    fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2, out=out)
    o = out.cpu().numpy()
    assert np.isfinite(o).all(), "wide f16x2 kernel left output elements unwritten"
    direct = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((direct - out).abs().max()) <= 2e-6
    for n in ((0, B - 1) if B > 1 else (0,)):
        ref = oracle.corr_fwd(a[n:n + 1], b[n:n + 1], 20, 1, 20, 1, 2)
        assert max_abs(o[n:n + 1], ref) <= 2e-6, "two-term f16 split should agree with the fp32 oracle to rounding"
    auto = torch.full_like(out, float("nan"))
    fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_AUTO, out=auto)
    assert torch.equal(auto, out), "FN2_CORR_AUTO should select the f16x2 kernel for this configuration"
    # fused LeakyReLU + channel slice
    buf = torch.full((B, 8 + 441 + 3, H, W), 7.0, device=dev)
    fn2_capi.correlation_forward_fused(ad, bd, buf, 8, 0.1, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_AUTO)
    assert (buf[:, :8] == 7.0).all() and (buf[:, 8 + 441:] == 7.0).all()
    assert torch.equal(buf[:, 8:8 + 441], torch.nn.functional.leaky_relu(out, 0.1))
    # tiny operands: the block scale keeps fp32-class relative accuracy
    sa, sb = 2.0 ** -21, 2.0 ** -19
    tiny = fn2_capi.correlation_forward(ad * sa, bd * sb, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert float((tiny / (sa * sb) - out).abs().max()) <= 4e-6
    # operands beyond the f16 range, inf and nan
    a2, b2 = ad.clone(), bd.clone()
    a2[0, 3, H // 2, W - 5] = 1.0e6; b2[0, 7, 1, 70] = -3.0e7; b2[B - 1, 9, H - 1, 3] = float("inf"); b2[B - 1, 1, 0, W // 2] = float("nan")
    big = fn2_capi.correlation_forward(a2, b2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    ref = fn2_capi.correlation_forward(a2, b2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    assert torch.equal(torch.isfinite(big), torch.isfinite(ref)), "non-finite outputs must coincide"
    fin = torch.isfinite(ref)
    assert int((~fin).sum()) > 0
    assert float(((big - ref)[fin].abs() / ref[fin].abs().clamp_min(1.0)).max()) <= 1e-5


def test_correlation_f16x2_full_size_and_fused(dev, oracle):
    """BASELINE configs[1] shape: against the direct kernel everywhere and the oracle on two batch items; the fused
    LeakyReLU / channel-slice epilogue (fn2_correlation_forward_fused) is bit-identical to the unfused result."""
    import fn2_capi
    g = torch.Generator().manual_seed(11)
    a = torch.randn(8, 256, 48, 64, generator=g)
    b = torch.randn(8, 256, 48, 64, generator=g)
    ad, bd = a.to(dev), b.to(dev)
    out = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    direct = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((out - direct).abs().max()) <= 2e-6
    for n in (0, 5):
        ref = oracle.corr_fwd(a[n:n + 1].numpy(), b[n:n + 1].numpy(), 20, 1, 20, 1, 2)
        assert max_abs(out[n:n + 1].cpu().numpy(), ref) <= 2e-6
    buf = torch.full((8, 32 + 441 + 3, 48, 64), 7.0, device=dev)
    fn2_capi.correlation_forward_fused(ad, bd, buf, 32, 0.1, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert (buf[:, :32] == 7.0).all() and (buf[:, 32 + 441:] == 7.0).all()
    assert torch.equal(buf[:, 32:32 + 441], torch.nn.functional.leaky_relu(out, 0.1))


def test_correlation_f16x2_out_of_range_operands(dev):
    """Operands that do not fit an f16 (|x| >= 65520), infinities and NaNs: the affected outputs are recomputed in plain
    fp32, so the result matches the fp32 kernels wherever those are finite and is non-finite exactly where they are."""
    import fn2_capi
    g = torch.Generator().manual_seed(5)
    a = torch.randn(2, 64, 16, 24, generator=g)
    b = torch.randn(2, 64, 16, 24, generator=g)
    a[0, 3, 5, 7] = 1.0e5; a[0, 10, 2, 20] = -3.0e6; b[0, 7, 9, 9] = 7.0e4; b[0, 63, 15, 23] = -1.0e30
    a[1, 0, 0, 0] = 65519.0; b[1, 1, 8, 12] = 65520.0; b[1, 5, 4, 4] = float("inf"); b[1, 9, 11, 3] = float("nan")
    ad, bd = a.to(dev), b.to(dev)
    out = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    ref = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT).double()
    p2 = torch.nn.functional.pad(bd.double(), (20, 20, 20, 20))
    scale = torch.cat([(ad.double().abs() * p2[:, :, 20 + 2 * tj:36 + 2 * tj, 20 + 2 * ti:44 + 2 * ti].abs()).mean(1, keepdim=True)
                       for tj in range(-10, 11) for ti in range(-10, 11)], 1)   # sum |a b| / C: the natural error scale
    fin = torch.isfinite(ref) & torch.isfinite(scale)
    assert torch.equal(torch.isfinite(out), torch.isfinite(ref.float())), "non-finite outputs must coincide"
    err = ((out.double() - ref).abs() / scale.clamp_min(1e-30))[fin]
    assert float(err.max()) <= 1e-5, float(err.max())
    assert int((~fin).sum()) > 0 and int(fin.sum()) > 0


def test_correlation_f16x2_accuracy_vs_fp64(dev):
    """Against an fp64 reference the f16x2 kernel is as accurate as the fp32 MFMA kernel for unit-scale, large and tiny
    inputs and for wide dynamic range (the operands are block-scaled per task, csrc/f16x2_split.h; the magnitude sweeps
    below go from 2^-27 to 2^13)."""
    import fn2_capi
    import torch.nn.functional as F
    B, C, H, W = 2, 256, 48, 64
    g = torch.Generator().manual_seed(22)

    def ref64(x1, x2):
        p2 = F.pad(x2.double(), (20, 20, 20, 20))
        outs = [(x1.double() * p2[:, :, 20 + 2 * tj:20 + 2 * tj + H, 20 + 2 * ti:20 + 2 * ti + W]).mean(1, keepdim=True)
                for tj in range(-10, 11) for ti in range(-10, 11)]
        return torch.cat(outs, 1)

    for scale, spread, factor in [(1.0, 0.0, 1.5), (100.0, 0.0, 1.5), (1.0, 3.0, 1.5), (1e-3, 0.0, 1.5), (1e-7, 0.0, 1.5)]:
        x1 = (scale * torch.randn(B, C, H, W, generator=g) * torch.exp(spread * torch.randn(B, C, H, W, generator=g))).to(dev)
        x2 = (scale * torch.randn(B, C, H, W, generator=g) * torch.exp(spread * torch.randn(B, C, H, W, generator=g))).to(dev)
        r = ref64(x1, x2)
        e32 = float((fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32).double() - r).abs().max())
        e16 = float((fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2).double() - r).abs().max())
        assert e16 <= factor * e32 + 1e-30, (scale, spread, e16, e32)
        assert e16 <= 1e-4 * max(1.0, float(r.abs().max())), (scale, spread, e16)


def test_correlation_algo_selector_is_validated(dev):
    """Unknown algo values are rejected (the profiling instantiations are only reachable through fn2_debug_*)."""
    import ctypes
    import fn2_capi
    x = torch.randn(1, 64, 8, 8, device=dev)
    out = torch.empty(1, 441, 8, 8, device=dev)
    for algo in (5, 99, 100, 2100, 5001, -1, 0x40000000 | 5001):
        rc = fn2_capi.lib().fn2_correlation_forward_ex(
            ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0, 1, 64, 8, 8,
            20, 1, 20, 1, 2, ctypes.c_int(algo), None)
        assert rc == -1, (algo, rc)   # FN2_EINVAL
    with pytest.raises(RuntimeError):   # outside the f16x2 domain (C % 64 != 0)
        y = torch.randn(1, 32, 8, 8, device=dev)
        fn2_capi.correlation_backward(y, y, torch.randn(1, 441, 8, 8, device=dev), 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)


def test_correlation_full_size_properties(dev):
    """Size-independent properties at the BASELINE size: the all-ones pattern counts padding,
    the output is linear in in2, and batch items are independent."""
    import fn2_capi
    ones = torch.ones(8, 256, 48, 64, device=dev)
    out = fn2_capi.correlation_forward(ones, ones, 20, 1, 20, 1, 2)
    ys = torch.arange(48, device=dev).view(48, 1)
    xs = torch.arange(64, device=dev).view(1, 64)
    for tj, ti in [(0, 0), (10, 10), (20, 3), (7, 19)]:
        dy, dx = 2 * (tj - 10), 2 * (ti - 10)
        exp = ((ys + dy >= 0) & (ys + dy < 48) & (xs + dx >= 0) & (xs + dx < 64)).float()
        assert torch.equal(out[3, tj * 21 + ti], exp), (tj, ti)
    g = torch.Generator().manual_seed(3)
    a = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    b1 = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    b2 = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    lhs = fn2_capi.correlation_forward(a, b1 + b2, 20, 1, 20, 1, 2)
    rhs = fn2_capi.correlation_forward(a, b1, 20, 1, 20, 1, 2) + fn2_capi.correlation_forward(a, b2, 20, 1, 20, 1, 2)
    assert float((lhs - rhs).abs().max()) <= 2e-6
    solo = fn2_capi.correlation_forward(a[5:6].contiguous(), b1[5:6].contiguous(), 20, 1, 20, 1, 2)
    full = fn2_capi.correlation_forward(a, b1, 20, 1, 20, 1, 2)
    assert torch.equal(solo[0], full[5])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
def test_correlation_other_dtypes(dev, oracle, dtype):
    import correlation_cuda
    rng = np.random.default_rng(11)
    a = rng.standard_normal((2, 24, 10, 12)).astype(np.float32)
    b = rng.standard_normal((2, 24, 10, 12)).astype(np.float32)
    ad, bd = torch.from_numpy(a).to(dev, dtype), torch.from_numpy(b).to(dev, dtype)
    out = ad.new_empty(0)
    correlation_cuda.forward(ad, bd, ad.new_empty(0), ad.new_empty(0), out, 4, 1, 4, 1, 2, 1)
    assert out.dtype == dtype and tuple(out.shape) == (2, 25, 10, 12)
    a_r, b_r = ad.double().cpu().numpy(), bd.double().cpu().numpy()
    ref = oracle.corr_fwd(a_r, b_r, 4, 1, 4, 1, 2)
    tol = TOL if dtype == torch.float64 else 3e-3
    assert max_abs(out.double().cpu().numpy(), ref) <= tol
    go = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32)).to(dev, dtype)
    g1, g2 = ad.new_empty(0), ad.new_empty(0)
    correlation_cuda.backward(ad, bd, ad.new_empty(0), ad.new_empty(0), go, g1, g2, 4, 1, 4, 1, 2, 1)
    r1, r2 = oracle.corr_bwd(a_r, b_r, go.double().cpu().numpy(), 4, 1, 4, 1, 2)
    assert max_abs(g1.double().cpu().numpy(), r1) <= tol * 4 and max_abs(g2.double().cpu().numpy(), r2) <= tol * 4


def test_correlation_kernel3_and_stride1(dev, oracle):
    """Parameter-general path: kernel_size 3 where the reference is defined (md - dr*s2 >= 1) and
    stride1 = 2 forward."""
    import fn2_capi
    rng = np.random.default_rng(12)
    a = rng.standard_normal((1, 8, 12, 12)).astype(np.float32)
    b = rng.standard_normal((1, 8, 12, 12)).astype(np.float32)
    for params in [(5, 3, 5, 1, 2), (4, 1, 4, 2, 2), (6, 3, 5, 2, 3)]:
        out = fn2_capi.correlation_forward(to_dev(a, dev), to_dev(b, dev), *params)
        assert max_abs(out.cpu().numpy(), oracle.corr_fwd(a, b, *params)) <= TOL
    out = fn2_capi.correlation_forward(to_dev(a, dev), to_dev(b, dev), 5, 3, 5, 1, 2)
    go = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    g1, g2 = fn2_capi.correlation_backward(to_dev(a, dev), to_dev(b, dev), to_dev(go, dev), 5, 3, 5, 1, 2)
    r1, r2 = oracle.corr_bwd(a, b, go, 5, 3, 5, 1, 2)
    assert max_abs(g1.cpu().numpy(), r1) <= TOL and max_abs(g2.cpu().numpy(), r2) <= TOL


# ------------------------------------------------------------------ autograd wrappers (reference API)
def test_wrappers_autograd(dev, oracle):
    from networks.channelnorm_package.channelnorm import ChannelNorm
    from networks.correlation_package.correlation import Correlation
    from networks.resample2d_package.resample2d import Resample2d
    g = torch.Generator().manual_seed(13)
    a = torch.randn(2, 32, 12, 16, generator=g)
    b = torch.randn(2, 32, 12, 16, generator=g)
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
    out = corr(ad, bd)
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(dev))
    r1, r2 = oracle.corr_bwd(a.numpy(), b.numpy(), go.numpy(), 20, 1, 20, 1, 2)
    assert max_abs(out.detach().cpu().numpy(), oracle.corr_fwd(a.numpy(), b.numpy(), 20, 1, 20, 1, 2)) <= TOL
    assert max_abs(ad.grad.cpu().numpy(), r1) <= TOL and max_abs(bd.grad.cpu().numpy(), r2) <= TOL

    # the FlowNet2 warp block (models.py:133-138): resample -> diff -> channelnorm -> cat
    x = torch.randn(2, 6, 16, 24, generator=g)
    flow = torch.randn(2, 2, 16, 24, generator=g) * 3
    xd, fd = x.to(dev).requires_grad_(True), flow.to(dev).requires_grad_(True)
    warped = Resample2d()(xd[:, 3:, :, :], fd)
    diff = xd[:, :3, :, :] - warped
    norm = ChannelNorm()(diff.contiguous())
    cat = torch.cat((xd, warped, fd / 20.0, norm), dim=1)
    w = torch.randn(cat.shape, generator=g).to(dev)
    (cat * w).sum().backward()
    # CPU restatement of the same graph with the oracle's explicit backward passes
    wn = w.cpu().numpy()
    warped_o = oracle.resample_fwd(x[:, 3:].contiguous().numpy(), flow.numpy())
    diff_o = x[:, :3].numpy() - warped_o
    norm_o = oracle.chnorm_fwd(diff_o)
    assert max_abs(norm.detach().cpu().numpy(), norm_o) <= TOL
    g_norm = wn[:, 11:12]
    g_diff = oracle.chnorm_bwd(diff_o, norm_o, np.ascontiguousarray(g_norm))
    g_warp = wn[:, 6:9] - g_diff
    gimg_o, gflow_o = oracle.resample_bwd(x[:, 3:].contiguous().numpy(), flow.numpy(), np.ascontiguousarray(g_warp))
    gx = wn[:, :6].copy()
    gx[:, :3] += g_diff
    gx[:, 3:] += gimg_o
    gf = wn[:, 9:11] / 20.0 + gflow_o
    assert max_abs(xd.grad.cpu().numpy(), gx) <= TOL
    assert max_abs(fd.grad.cpu().numpy(), gf) <= TOL


def _fuzz_cases(seed, n):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        md = int(rng.choice([2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 21]))
        C = int(rng.choice([1, 3, 16, 32, 48, 64, 128]))
        H = int(rng.integers(2, 27))
        W = int(rng.integers(2, 75))
        if rng.random() < 0.6:       # bias towards the tiled kernels' domain
            H += H & 1
            W += (-W) % 4
            C = max(16, C - C % 16)
        cases.append((int(rng.integers(1, 4)), C, H, W, md))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(2024, 36))
def test_correlation_fuzz_vs_oracle(dev, oracle, case):
    """Seeded random shapes through the automatic dispatch (tiled MFMA kernels where they apply, the general kernel
    otherwise), forward, fused forward and backward against the oracle."""
    import fn2_capi
    B, C, H, W, md = case
    rng = np.random.default_rng(hash(case) % (1 << 31))
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    D2 = (2 * (md // 2) + 1) ** 2
    go = rng.standard_normal((B, D2, H, W)).astype(np.float32)
    ad, bd, gd = to_dev(a, dev), to_dev(b, dev), to_dev(go, dev)
    ref = oracle.corr_fwd(a, b, md, 1, md, 1, 2)
    out = torch.full((B, D2, H, W), float("nan"), device=dev)
    fn2_capi.correlation_forward(ad, bd, md, 1, md, 1, 2, out=out)
    assert max_abs(out.cpu().numpy(), ref) <= 2e-6
    buf = torch.full((B, D2 + 2, H, W), float("nan"), device=dev)
    fn2_capi.correlation_forward_fused(ad, bd, buf, 1, 0.1, md, 1, md, 1, 2)
    lref = np.where(ref > 0, ref, ref * np.float32(0.1))
    assert max_abs(buf[:, 1:1 + D2].cpu().numpy(), lref) <= 2e-6
    assert torch.isnan(buf[:, 0]).all() and torch.isnan(buf[:, 1 + D2]).all()
    r1, r2 = oracle.corr_bwd(a, b, go, md, 1, md, 1, 2)
    g1 = torch.full((B, C, H, W), float("nan"), device=dev)
    g2 = torch.full((B, C, H, W), float("nan"), device=dev)
    fn2_capi.correlation_backward(ad, bd, gd, md, 1, md, 1, 2, out=(g1, g2))
    scale = max(1.0, float(np.abs(r1).max()), float(np.abs(r2).max()))
    assert max_abs(g1.cpu().numpy(), r1) <= 5e-6 * scale and max_abs(g2.cpu().numpy(), r2) <= 5e-6 * scale


def _fuzz_img_cases(seed, n):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(1, 70)), int(rng.integers(1, 150)),
             bool(rng.integers(0, 2))) for _ in range(n)]


@pytest.mark.parametrize("case", _fuzz_img_cases(7, 24))
def test_resample_channelnorm_fuzz_vs_oracle(dev, oracle, case):
    """Seeded random image shapes (tiled and untiled kernels, W % 4 != 0, single rows/columns): Resample2d forward and
    backward, ChannelNorm forward and backward, and the fused warp-diff-norm-cat row against the oracle."""
    import fn2_capi
    from networks.resample2d_package.resample2d import Resample2dFunction
    from networks.channelnorm_package.channelnorm import ChannelNormFunction
    B, C, H, W, bilinear = case
    rng = np.random.default_rng(hash(case) % (1 << 31))
    img = rng.standard_normal((B, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((B, 2, H, W)) * 3.0).astype(np.float32)
    flow.reshape(-1)[rng.integers(0, flow.size, max(1, flow.size // 50))] *= 30.0
    gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
    imd = to_dev(img, dev).requires_grad_(True)
    fld = to_dev(flow, dev).requires_grad_(True)
    out = Resample2dFunction.apply(imd, fld, 1, bilinear)
    ref = oracle.resample_fwd(img, flow, 1, bilinear)
    assert max_abs(out.detach().cpu().numpy(), ref) <= TOL
    out.backward(to_dev(gout, dev))
    rgi, rgf = oracle.resample_bwd(img, flow, gout, 1, bilinear)
    s = max(1.0, float(np.abs(rgi).max()), float(np.abs(rgf).max()))
    assert max_abs(imd.grad.cpu().numpy(), rgi) <= 1e-5 * s
    assert max_abs(fld.grad.cpu().numpy(), rgf) <= 1e-5 * s
    x = to_dev(img, dev).requires_grad_(True)
    nrm = ChannelNormFunction.apply(x, 2)
    assert max_abs(nrm.detach().cpu().numpy(), oracle.chnorm_fwd(img)) <= 1e-6 * max(1.0, float(np.abs(img).max()) * C)
    pair = rng.standard_normal((B, 2 * C, H, W)).astype(np.float32)
    got = fn2_capi.warp_diff_norm_cat(to_dev(pair, dev), to_dev(flow, dev), 20.0, bilinear).cpu().numpy()
    warped = oracle.resample_fwd(np.ascontiguousarray(pair[:, C:]), flow, 1, bilinear)
    refc = np.concatenate((pair, warped, flow * (np.float32(1.0) / np.float32(20.0)), oracle.chnorm_fwd(pair[:, :C] - warped)), axis=1)
    assert max_abs(got, refc) <= TOL


@pytest.mark.parametrize("shape", [(2, 64, 128), (8, 384, 512), (1, 100, 200), (3, 192, 64)])
def test_multiscale_l1_epe(dev, shape):
    """the fused MultiScale-L1 loss + EPE against the numpy restatement of losses.py:52-86, against the
    reference's formulation in plain PyTorch on the GPU, and its gradient against autograd of that formulation."""
    import fn2_capi
    from oracle.oracle import multiscale_l1_epe_sums
    from losses_fused import MultiScaleL1
    B, H, W = shape
    rng = np.random.default_rng(B + H + W)
    target = (rng.standard_normal((B, 2, H, W)) * 5.0).astype(np.float32)
    outs = [rng.standard_normal((B, 2, H // (4 << i), W // (4 << i))).astype(np.float32) * 0.3 for i in range(5)]
    weights = [0.32 / 2 ** i for i in range(5)]
    td, od = to_dev(target, dev), [to_dev(o, dev) for o in outs]
    sums, _ = fn2_capi.multiscale_l1_epe(od, td, weights)
    rl1, repe = multiscale_l1_epe_sums(outs, target)
    got = sums.cpu().numpy().astype(np.float64)
    for i in range(5):
        if outs[i].size == 0:
            assert got[i] == 0 and got[5 + i] == 0
            continue
        assert abs(got[i] - rl1[i]) <= 2e-5 * max(1.0, rl1[i]), (i, got[i], rl1[i])
        assert abs(got[5 + i] - repe[i]) <= 2e-5 * max(1.0, repe[i]), (i, got[5 + i], repe[i])
    # the reference's statements (losses.py:74-78) on the same device tensors, and autograd through them
    if all(o.size for o in outs):
        ot = [o.clone().requires_grad_(True) for o in od]
        t = 0.05 * td
        loss_ref = sum(w * torch.abs(o - torch.nn.functional.avg_pool2d(t, 4 << i, 4 << i)).mean() for i, (w, o) in enumerate(zip(weights, ot)))
        epe_ref = sum(w * torch.norm(torch.nn.functional.avg_pool2d(t, 4 << i, 4 << i) - o, p=2, dim=1).mean() for i, (w, o) in enumerate(zip(weights, ot)))
        loss_ref.backward()
        of = [o.clone().requires_grad_(True) for o in od]
        loss, epe = MultiScaleL1()(tuple(of), td)
        assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-5 * max(1.0, abs(float(loss_ref.detach())))
        assert abs(float(epe.detach()) - float(epe_ref.detach())) <= 1e-5 * max(1.0, abs(float(epe_ref.detach())))
        (2.0 * loss).backward()
        for a, b in zip(of, ot):
            # sign() flips where |out - t_i| is at rounding level: allow a handful of such elements
            diff = (a.grad - 2.0 * b.grad).abs()
            assert int((diff > 1e-9).sum()) <= max(2, a.numel() // 5000), int((diff > 1e-9).sum())


@pytest.mark.parametrize("path", golden_files("multiscale"), ids=lambda p: os.path.basename(p))
@pytest.mark.parametrize("norm", ["L1", "L2"])
def test_multiscale_golden(dev, path, norm):
    """the fused loss against numbers the REFERENCE's losses.py produced (tests/golden/make_golden_losses.py:
    MultiScale(args, norm)(outputs, target) and autograd through it) -- loss, EPE and every prediction's gradient."""
    from losses_fused import MultiScale
    d = np.load(path)
    outs = [to_dev(d[f"out{i}"], dev).requires_grad_(True) for i in range(5)]
    loss, epe = MultiScale(None, norm=norm)(tuple(outs), to_dev(d["target"], dev))
    rl, re = float(d[f"loss_{norm}"]), float(d[f"epe_{norm}"])
    assert abs(float(loss.detach()) - rl) <= 1e-5 * max(1.0, abs(rl)), (float(loss.detach()), rl)
    assert abs(float(epe.detach()) - re) <= 1e-5 * max(1.0, abs(re)), (float(epe.detach()), re)
    loss.backward()
    for i, o in enumerate(outs):
        ref = d[f"grad_{norm}_{i}"]
        diff = np.abs(o.grad.cpu().numpy().astype(np.float64) - ref)
        scale = max(float(np.abs(ref).max()), float(d["weights"][i]) / ref.size)   # (the 1 x 1 level of the zero_diff fixture is all zeros)
        if norm == "L1":
            # sign() is decided at rounding level where |out - t_i| is: a handful of flips are allowed; the elements the fixture
            # made EXACTLY equal to torch's pooled target (reference gradient 0) may come out +-w/N with this kernel's summation order
            assert int(((diff > 1e-6 * scale) & (ref != 0)).sum()) <= max(2, ref.size // 5000), int((diff > 1e-6 * scale).sum())
            assert float(diff.max()) <= 1.000001 * scale
        else:
            # (out - t) / ||out - t||: relative error grows where the norm is at rounding level; pixels the fixture made EXACTLY equal to
            # torch's pooled target (reference gradient 0: torch.norm's backward at 0) come out as a unit vector x w/N wherever this
            # kernel's summation order leaves a difference of one ulp
            zero_px = np.broadcast_to(np.abs(ref).sum(axis=1, keepdims=True) == 0, ref.shape)
            assert int(((diff > 1e-4 * scale) & ~zero_px).sum()) <= max(2, ref.size // 5000), float(diff.max() / scale)
            assert float(diff.max()) <= 1.00001 * float(d["weights"][i]) / (ref.size / 2)        # a unit vector x w / (N / 2)


def test_multiscale_l2_vs_autograd(dev):
    """norm = 'L2' (losses.py:64-67) at the training shape against autograd of the reference's statements on the device."""
    from losses_fused import MultiScale
    g = torch.Generator().manual_seed(31)
    B, H, W = 8, 384, 512
    target = (torch.randn(B, 2, H, W, generator=g) * 5).to(dev)
    outs = [(torch.randn(B, 2, H // (4 << i), W // (4 << i), generator=g) * 0.3).to(dev) for i in range(5)]
    weights = [0.32 / 2 ** i for i in range(5)]
    ot = [o.clone().requires_grad_(True) for o in outs]
    t = 0.05 * target
    ref = sum(w * torch.norm(o - torch.nn.functional.avg_pool2d(t, 4 << i, 4 << i), p=2, dim=1).mean() for i, (w, o) in enumerate(zip(weights, ot)))
    ref.backward()
    of = [o.clone().requires_grad_(True) for o in outs]
    loss, epe = MultiScale(None, norm="L2")(tuple(of), target)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    assert float(loss.detach()) == float(epe.detach())
    (0.5 * loss).backward()
    for a, b in zip(of, ot):
        s = float(b.grad.abs().max())
        assert float((a.grad - 0.5 * b.grad).abs().max()) <= 2e-5 * s


def test_multiscale_l1_double_backward_and_layouts(dev):
    """the fused loss must not rescale its cached gradients in place (second backward over the same graph,
    gradient scalers), must take non-contiguous predictions (channels_last), and accepts a single tensor (eval branch of
    the reference, losses.py:80-83)."""
    from losses_fused import MultiScaleL1
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 128, 192
    target = (torch.randn(B, 2, H, W, generator=g) * 5).to(dev)
    outs = [(torch.randn(B, 2, H // (4 << i), W // (4 << i), generator=g) * 0.3).to(dev) for i in range(5)]
    crit = MultiScaleL1()
    of = [o.clone().requires_grad_(True) for o in outs]
    loss, _ = crit(tuple(of), target)
    (3.0 * loss).backward(retain_graph=True)
    first = [o.grad.clone() for o in of]
    for o in of:
        o.grad = None
    (3.0 * loss).backward()
    for a, b in zip(of, first):
        assert torch.equal(a.grad, b), "second backward over the same graph changed the gradients"
    # non-contiguous predictions: same values, gradients arrive
    nc = [o.clone().to(memory_format=torch.channels_last).requires_grad_(True) for o in outs]
    loss2, _ = crit(tuple(nc), target)
    loss2.backward()
    assert abs(float(loss2.detach()) - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach())))
    for a, b in zip(nc, first):
        assert a.grad is not None and float((3.0 * a.grad - b).abs().max()) <= 1e-7
    # single full-resolution tensor
    full = (torch.randn(B, 2, H, W, generator=g)).to(dev)
    l1, epe = crit(full, target)
    assert abs(float(l1) - float((full - target).abs().mean())) <= 1e-6
    assert abs(float(epe) - float(torch.norm(target - full, p=2, dim=1).mean())) <= 1e-5


def test_resample2d_flag_and_kernel_size_handling(dev):
    """any non-zero `bilinear` means bilinear at the public C entry points (no profiling bits there), and
    kernel_size != 1 is refused when the module is built, with a clear message."""
    import ctypes
    import fn2_capi
    from networks.resample2d_package.resample2d import Resample2d
    g = torch.Generator().manual_seed(2)
    img = torch.rand(2, 3, 32, 64, generator=g).to(dev)
    flow = (torch.randn(2, 2, 32, 64, generator=g) * 3).to(dev)
    outs = []
    for flag in (1, 2, 0x100, 0x1001):
        out = torch.empty_like(img)
        rc = fn2_capi.lib().fn2_resample2d_forward(ctypes.c_void_p(img.data_ptr()), None, ctypes.c_void_p(flow.data_ptr()),
                                                   ctypes.c_void_p(out.data_ptr()), 2, 3, 32, 64, 32, 64, 1, flag, None)
        assert rc == 0
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    with pytest.raises(ValueError, match="kernel_size must be >= 1"):
        Resample2d(kernel_size=0)


@pytest.mark.gpu
@pytest.mark.parametrize("ks", [2, 3, 4])
@pytest.mark.parametrize("shape", [(2, 3, 32, 64), (1, 2, 9, 13)])
def test_resample2d_kernel_size_window(dev, oracle, ks, shape):
    """kernel_size > 1 (window sums, resample2d_kernel.cu:54-61 / :116-123 / :171-191) through the module,
    forward and both gradients, against the oracle -- including samples whose shifted corners are clamped at the border."""
    from networks.resample2d_package.resample2d import Resample2d
    B, C, H, W = shape
    g = torch.Generator().manual_seed(70 + ks)
    img = (torch.rand(B, C, H, W, generator=g) - 0.5)
    flow = torch.randn(B, 2, H, W, generator=g) * 3
    gout = torch.randn(B, C, H, W, generator=g)
    for bilinear in (True, False):
        imd, fld = img.to(dev).requires_grad_(True), flow.to(dev).requires_grad_(True)
        out = Resample2d(kernel_size=ks, bilinear=bilinear)(imd, fld)
        ref = oracle.resample_fwd(img.numpy(), flow.numpy(), ks, bilinear)
        assert max_abs(out.detach().cpu().numpy(), ref) <= 1e-5
        out.backward(gout.to(dev))
        r1, r2 = oracle.resample_bwd(img.numpy(), flow.numpy(), gout.numpy(), ks, bilinear)
        # a window sums ks^2 x 4 terms per pixel (grad_img: in atomic order): tolerance relative to the result's magnitude
        for got, want in ((imd.grad, r1), (fld.grad, r2)):
            assert max_abs(got.cpu().numpy(), want) <= 1e-5 * (1.0 + float(np.abs(want).max()))


def test_correlation_f16x2_repeatable_at_full_size(dev):
    """Regression (round 2): the row-store epilogue once let an inline-asm register copy land in a data register of the 16-byte
    store issued just before it -- a few dozen of the 10.8 M outputs, different ones every launch, came out as 1/C.  Twenty
    launches of forward and backward at BASELINE size must agree bit for bit with each other and with the general kernel
    within tolerance."""
    import fn2_capi
    g = torch.Generator().manual_seed(11)
    a = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    b = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    go = torch.randn(8, 441, 48, 64, generator=g).to(dev)
    ref = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=1)
    r1, r2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, algo=1)
    first = None
    for _ in range(20):
        out = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=4)
        g1, g2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, algo=4)
        assert float((out - ref).abs().max()) <= 1e-5
        assert float((g1 - r1).abs().max()) <= 1e-5 and float((g2 - r2).abs().max()) <= 1e-5
        if first is None:
            first = (out.clone(), g1.clone(), g2.clone())
        else:
            assert torch.equal(out, first[0]) and torch.equal(g1, first[1]) and torch.equal(g2, first[2])


# ------------------------------------------------------------------ magnitude sweeps
def _corr_fwd_fp64(a, b):
    """fp64 cost volume of FlowNetC's configuration on the device (correlation_cuda_kernel.cu:73-147, closed form SURVEY 8a a4)."""
    import torch.nn.functional as F
    H, W = a.shape[2:]
    p2 = F.pad(b.double(), (20, 20, 20, 20))
    a64 = a.double()
    return torch.cat([(a64 * p2[:, :, 20 + 2 * tj:20 + 2 * tj + H, 20 + 2 * ti:20 + 2 * ti + W]).mean(1, keepdim=True)
                      for tj in range(-10, 11) for ti in range(-10, 11)], 1)


def _corr_bwd_fp64(a, b, go):
    """fp64 input gradients (closed forms SURVEY 8a a7, a8; correlation_cuda_kernel.cu:150-334), no autograd graph."""
    import torch.nn.functional as F
    B, C, H, W = a.shape
    a64, go64 = a.double(), go.double()
    p2 = F.pad(b.double(), (20, 20, 20, 20))
    g1 = torch.zeros_like(a64)
    g2p = torch.zeros_like(p2)
    d = 0
    for tj in range(-10, 11):
        for ti in range(-10, 11):
            g = go64[:, d:d + 1]
            ys, xs = slice(20 + 2 * tj, 20 + 2 * tj + H), slice(20 + 2 * ti, 20 + 2 * ti + W)
            g1 += g * p2[:, :, ys, xs]
            g2p[:, :, ys, xs] += g * a64
            d += 1
    return g1 / C, g2p[:, :, 20:20 + H, 20:20 + W] / C


_SWEEP = [-27, -20, -13, -7, 0, 7, 13]   # powers of two ~ 1e-8, 1e-6, 1e-4, 1e-2, 1, 1e2, 1e4: the fp64 reference scales exactly


def _rel(got, ref):
    return float((got.double() - ref).abs().max()) / float(ref.abs().max())


def test_correlation_forward_magnitude_sweep(dev):
    """The f16x2 forward (and AUTO, which selects it) at input magnitudes 1e-8 .. 1e4, each input scaled independently:
    error relative to the largest output, against fp64, within 3x of the fp32 MFMA kernel's at EVERY magnitude -- the
    reference forms its products and sums in fp32 whatever the scale (correlation_cuda_kernel.cu:112,124)."""
    import fn2_capi
    B, C, H, W = 1, 256, 48, 64
    g = torch.Generator().manual_seed(31)
    x1 = torch.randn(B, C, H, W, generator=g).to(dev)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    r = _corr_fwd_fp64(x1, x2)
    worst = 0.0
    for ea in _SWEEP:
        for eb in _SWEEP:
            a, b, ref = torch.ldexp(x1, torch.tensor(ea)), torch.ldexp(x2, torch.tensor(eb)), torch.ldexp(r, torch.tensor(ea + eb))
            e32 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32), ref)
            e16 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2), ref)
            eau = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2), ref)
            assert e16 <= 3.0 * e32 and eau <= 3.0 * e32, (ea, eb, e16, eau, e32)
            assert e16 <= 1e-6, (ea, eb, e16)
            worst = max(worst, e16 / e32)
    print("forward sweep: worst f16x2 / f32 relative-error ratio", worst)
    # scales that are not powers of two (the scaled inputs round): own fp64 reference
    for sa, sb in ((1e-6, 1e-6), (3e-8, 7e3), (1e-3, 1e-5)):
        a, b = (x1 * sa).contiguous(), (x2 * sb).contiguous()
        ref = _corr_fwd_fp64(a, b)
        e32 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32), ref)
        e16 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2), ref)
        assert e16 <= 3.0 * e32, (sa, sb, e16, e32)


def test_correlation_backward_magnitude_sweep(dev):
    """The f16x2 backward (and AUTO) with the inputs and gradOutput scaled independently from 1e-8 to 1e4: both gradients
    within 3x of the fp32 MFMA kernel's relative error against fp64.  gradOutput reaches this layer at 1e-6 .. 1e-8 in
    training (mean-reduced MultiScale L1), which the unscaled split of round 2 could not represent."""
    import fn2_capi
    B, C, H, W = 1, 256, 48, 64
    g = torch.Generator().manual_seed(32)
    x1 = torch.randn(B, C, H, W, generator=g).to(dev)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    r1, r2 = _corr_bwd_fp64(x1, x2, go)
    worst = 0.0
    for ei in _SWEEP:
        for eg in _SWEEP:
            a, b, gg = torch.ldexp(x1, torch.tensor(ei)), torch.ldexp(x2, torch.tensor(ei)), torch.ldexp(go, torch.tensor(eg))
            f1, f2 = torch.ldexp(r1, torch.tensor(ei + eg)), torch.ldexp(r2, torch.tensor(ei + eg))
            res = {}
            for name, algo in (("f32", fn2_capi.FN2_CORR_MFMA_F32), ("f16x2", fn2_capi.FN2_CORR_MFMA_F16X2), ("auto", fn2_capi.FN2_CORR_AUTO)):
                g1, g2 = fn2_capi.correlation_backward(a, b, gg, 20, 1, 20, 1, 2, algo=algo)
                res[name] = max(_rel(g1, f1), _rel(g2, f2))
            assert res["f16x2"] <= 3.0 * res["f32"] and res["auto"] <= 3.0 * res["f32"], (ei, eg, res)
            assert res["f16x2"] <= 2e-6, (ei, eg, res)
            worst = max(worst, res["f16x2"] / res["f32"])
    print("backward sweep: worst f16x2 / f32 relative-error ratio", worst)
    # different magnitudes for the two inputs, scales that are not powers of two
    for s1, s2, sg in ((1e-6, 1e-6, 1e-6), (3e-8, 7e3, 1e-7), (1e2, 1e-4, 3e-8)):
        a, b, gg = (x1 * s1).contiguous(), (x2 * s2).contiguous(), (go * sg).contiguous()
        f1, f2 = _corr_bwd_fp64(a, b, gg)
        q1, q2 = fn2_capi.correlation_backward(a, b, gg, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32)
        g1, g2 = fn2_capi.correlation_backward(a, b, gg, 20, 1, 20, 1, 2)
        assert _rel(g1, f1) <= 3.0 * _rel(q1, f1) and _rel(g2, f2) <= 3.0 * _rel(q2, f2), (s1, s2, sg)


def test_correlation_backward_on_real_training_gradient(dev):
    """gradOutput as it really reaches Correlation.backward: captured inside Trainer.train_step (FlowNet2C, bs 8 @ 384x512,
    MultiScale L1) together with the conv3 features.  Element magnitudes are ~1e-7: checked against fp64
    relative to the largest gradient element and per batch item, next to the fp32 MFMA kernel."""
    import fn2_capi
    from harness.train import Trainer, synthetic_batch
    tr = Trainer(dev)
    tr.model.fused_training = False          # the separate Correlation module, whose gradOutput is what this test captures
    cap = {}
    inner = tr.model.corr.forward

    def spy(c3a, c3b):
        out = inner(c3a, c3b)
        cap["a"], cap["b"] = c3a.detach().clone(), c3b.detach().clone()
        out.register_hook(lambda gr: cap.__setitem__("go", gr.detach().clone()))
        return out

    tr.model.corr.forward = spy
    inputs, target = synthetic_batch(8, 384, 512, dev)
    tr.train_step(inputs, target)
    a, b, go = cap["a"].contiguous(), cap["b"].contiguous(), cap["go"].contiguous()
    assert tuple(a.shape) == (8, 256, 48, 64) and tuple(go.shape) == (8, 441, 48, 64)
    gmax = float(go.abs().max())
    print("training gradOutput: max |g| %.3e, rms %.3e; conv3 features max %.3e" % (gmax, float(go.pow(2).mean().sqrt()), float(a.abs().max())))
    assert 0.0 < gmax < 1e-3, "expected the tiny mean-reduced loss gradient here"
    f1, f2 = _corr_bwd_fp64(a, b, go)
    q1, q2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32)
    g1, g2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2)            # what autograd calls
    for got, q, ref in ((g1, q1, f1), (g2, q2, f2)):
        assert _rel(got, ref) <= 3.0 * _rel(q, ref), (_rel(got, ref), _rel(q, ref))
        for n in range(8):
            assert _rel(got[n], ref[n]) <= 3.0 * _rel(q[n], ref[n]) + 1e-9, n
        # element-wise: error against the natural scale of each element's sum, mean_c-free: |g| summed over the window
        assert _rel(got, ref) <= 2e-6
    # and the forward on the same features
    r = _corr_fwd_fp64(a, b)
    e32 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32), r)
    e16 = _rel(fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2), r)
    assert e16 <= 3.0 * e32, (e16, e32)


def test_correlation_f16x2_scale_sample_misses(dev):
    """The operand scales come from a task's first 64 channels / first gradOutput image.  Operands far above that sample
    (here: later channels 3e4 x larger, later displacement rows 1e5 x larger) overflow the scaled f16 and take the fp32
    recompute path; a sample of exact zeros means no scaling.  Results stay correct in both cases."""
    import fn2_capi
    g = torch.Generator().manual_seed(33)
    B, C, H, W = 1, 128, 8, 16
    a = torch.randn(B, C, H, W, generator=g)
    b = torch.randn(B, C, H, W, generator=g)
    go = torch.randn(B, 441, H, W, generator=g)
    a[:, 64:] *= 3e4
    b[:, 100:] *= 5e3
    go[:, :100] *= 1e5
    ad, bd, gd = a.to(dev), b.to(dev), go.to(dev)
    ref = _corr_fwd_fp64(ad, bd)
    out = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert _rel(out, ref) <= 1e-6, _rel(out, ref)
    f1, f2 = _corr_bwd_fp64(ad, bd, gd)
    g1, g2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert _rel(g1, f1) <= 2e-6 and _rel(g2, f2) <= 2e-6, (_rel(g1, f1), _rel(g2, f2))
    # all-zero sample, tiny data elsewhere
    a2, b2, go2 = a * 0, b * 0, go * 0
    a2[:, 64:] = 1e-3 * torch.randn(B, 64, H, W, generator=g)
    b2[:, 64:] = 1e-3 * torch.randn(B, 64, H, W, generator=g)
    go2[:, 300:] = 1e-3 * torch.randn(B, 141, H, W, generator=g)
    ad, bd, gd = a2.to(dev), b2.to(dev), go2.to(dev)
    ref = _corr_fwd_fp64(ad, bd)
    out = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert float((out.double() - ref).abs().max()) <= 1e-9
    f1, f2 = _corr_bwd_fp64(ad, bd, gd)
    g1, g2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert float((g1.double() - f1).abs().max()) <= 1e-9 and float((g2.double() - f2).abs().max()) <= 1e-9


def test_correlation_auto_falls_back_when_f16x2_declines(dev):
    """Shapes inside corr_f16x2_applicable's domain that its launcher declines (task table: H > 512; 16-bit task index:
    B x tasks-per-item >= 65536) must still work through FN2_CORR_AUTO / fn2_correlation_forward: they go on to
    the fp32 MFMA / general kernels; the explicit selector reports FN2_EUNSUPPORTED."""
    import fn2_capi
    g = torch.Generator().manual_seed(41)
    for shape in ((1, 64, 1024, 64), (10940, 64, 2, 8), (1, 64, 1024, 72), (3700, 64, 2, 72)):   # the last two: column windows
        a = torch.randn(shape, generator=g).to(dev)
        b = torch.randn(shape, generator=g).to(dev)
        with pytest.raises(RuntimeError):
            fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
        out = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2)
        ref = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
        assert float((out - ref).abs().max()) <= 2e-6
        del out, ref
        if shape[0] == 1:   # backward: the f16x2 launcher has no such limits, AUTO stays on it
            go = torch.randn(shape[0], 441, shape[2], shape[3], generator=g).to(dev)
            g1, g2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2)
            r1, r2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
            assert float((g1 - r1).abs().max()) <= 1e-5 and float((g2 - r2).abs().max()) <= 1e-5
    # half tensors: the same launcher limits, the same fall-through (here to the general kernel)
    for shape in ((1, 128, 1024, 16), (1, 128, 1024, 72)):
        a = torch.randn(shape, generator=g).half().to(dev)
        b = torch.randn(shape, generator=g).half().to(dev)
        with pytest.raises(RuntimeError):
            fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
        out = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2)
        ref = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
        assert torch.equal(out, ref)


# ------------------------------------------------------------------ half tensors on the matrix cores (correlation_f16_fwd.hip)
@pytest.mark.parametrize("case", [(1, 128, 6, 8), (2, 128, 16, 24), (1, 256, 22, 56), (1, 128, 46, 64), (3, 128, 2, 16), (2, 256, 48, 64),
                                  (1, 128, 8, 72), (2, 128, 10, 96), (1, 256, 56, 128), (1, 128, 4, 200), (3, 128, 6, 104)])   # W > 64: column windows
def test_correlation_half_forward_matrix_kernel(dev, oracle, case):
    """Half inputs (the reference dispatches its kernels for at::Half, correlation_cuda_kernel.cu:386-415): the single-product f16
    MFMA kernel against the oracle on the half-rounded inputs.  Products are exact and accumulated in fp32 here (the reference
    rounds every product to half, :112): the only error left is the rounding of the result to half, so the bound is one half
    ulp of the largest output -- tighter than the 3e-3 the general half kernel is held to.  Every output element written."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C + H + W)
    a = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).half()
    b = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).half()
    ad, bd = a.to(dev), b.to(dev)
    ref = oracle.corr_fwd(a.float().numpy(), b.float().numpy(), 20, 1, 20, 1, 2)
    out = torch.full((B, 441, H, W), float("nan"), dtype=torch.float16, device=dev)
    fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, out=out)                   # AUTO
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all(), "unwritten output elements"
    scale = float(np.abs(ref).max())
    assert max_abs(got, ref) <= 2.0 ** -11 * scale + 1e-6, (max_abs(got, ref), scale)
    direct = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    assert float((out.float() - direct.float()).abs().max()) <= 3e-3 * max(1.0, scale)
    sel = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert torch.equal(sel, out)                                                     # the explicit selector reaches the same kernel
    # fused LeakyReLU + concat slice in half; the bytes around the slice stay untouched
    buf = torch.full((B, 8 + 441 + 3, H, W), 7.0, dtype=torch.float16, device=dev)
    fn2_capi.correlation_forward_fused(ad, bd, buf, 8, 0.1, 20, 1, 20, 1, 2)
    assert (buf[:, :8] == 7.0).all() and (buf[:, 8 + 441:] == 7.0).all()
    want = torch.nn.functional.leaky_relu(torch.from_numpy(ref).to(dev), 0.1)
    assert float((buf[:, 8:8 + 441].float() - want).abs().max()) <= 2.0 ** -11 * scale + 1e-6


@pytest.mark.parametrize("case", [(1, 64, 6, 8), (2, 64, 16, 24), (1, 128, 22, 56), (1, 64, 46, 64), (3, 64, 2, 16), (2, 256, 48, 64), (1, 192, 12, 40)])
def test_correlation_half_backward_matrix_kernel(dev, oracle, case):
    """Half tensors through the backward (the reference dispatches it for at::Half, correlation_cuda_kernel.cu:460-554, and sums in
    half there; here the products are exact and the sums fp32, like the oracle on the half-rounded inputs): the single-product
    f16 MFMA kernel against the oracle -- the only error left is the rounding of each result to half; every element written; AUTO
    and the explicit selector reach it; non-finite inputs give the general kernel's finite / non-finite pattern."""
    import fn2_capi
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C + H + W + 9)
    a = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).half()
    b = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).half()
    go = torch.from_numpy(rng.standard_normal((B, 441, H, W)).astype(np.float32)).half()
    ad, bd, gd = a.to(dev), b.to(dev), go.to(dev)
    r1, r2 = oracle.corr_bwd(a.float().numpy(), b.float().numpy(), go.float().numpy(), 20, 1, 20, 1, 2)
    g1 = torch.full((B, C, H, W), float("nan"), dtype=torch.float16, device=dev)
    g2 = torch.full((B, C, H, W), float("nan"), dtype=torch.float16, device=dev)
    fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, out=(g1, g2))               # AUTO
    for got, ref in ((g1, r1), (g2, r2)):
        n = got.float().cpu().numpy()
        assert np.isfinite(n).all(), "unwritten gradient elements"
        assert (np.abs(n - ref) <= 2.0 ** -11 * np.abs(ref) + 2e-6 * np.abs(ref).max() + 1e-7).all(), float(np.abs(n - ref).max())
    s1, s2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2)
    assert torch.equal(s1, g1) and torch.equal(s2, g2)                                     # the explicit selector: same kernel
    d1, d2 = fn2_capi.correlation_backward(ad, bd, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    scale = max(1.0, float(np.abs(r1).max()))
    assert float((d1.float() - g1.float()).abs().max()) <= 3e-3 * scale and float((d2.float() - g2.float()).abs().max()) <= 3e-3 * scale
    # tiny gradOutput (f16 subnormals included) and non-finite values
    t1, t2 = fn2_capi.correlation_backward(ad, bd, gd * 2.0 ** -12, 20, 1, 20, 1, 2)
    q1, q2 = oracle.corr_bwd(a.float().numpy(), b.float().numpy(), (go * 2.0 ** -12).float().numpy(), 20, 1, 20, 1, 2)
    for got, ref in ((t1, q1), (t2, q2)):
        assert (np.abs(got.float().cpu().numpy() - ref) <= 2.0 ** -11 * np.abs(ref) + 6e-8 + 2e-6 * np.abs(ref).max()).all()
    if H >= 6 and W >= 16:
        a2, b2, g3 = ad.clone(), bd.clone(), gd.clone()
        a2[0, 3, H // 2, W - 5] = float("inf"); b2[0, 7, 1, 3] = float("nan"); g3[B - 1, 220, 0, 3] = float("inf"); g3[0, 17, H - 1, 9] = 65504.0
        n1, n2 = fn2_capi.correlation_backward(a2, b2, g3, 20, 1, 20, 1, 2)
        f1, f2 = fn2_capi.correlation_backward(a2.float(), b2.float(), g3.float(), 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
        for got, ref in ((n1, f1), (n2, f2)):
            small = torch.isfinite(ref) & (ref.abs() < 6.0e4)       # representable in half
            assert torch.equal(torch.isnan(got), torch.isnan(ref)), "nan pattern"
            assert bool(torch.isfinite(got)[small].all()) and int((~torch.isfinite(ref)).sum()) > 0
            assert float((got.float() - ref)[small].abs().max()) <= 2.0 ** -10 * float(ref[small].abs().max())


def test_correlation_half_special_values_and_wrapper(dev):
    """inf / nan / f16-subnormal inputs are matrix operands like any other (no out-of-range path to get wrong), and the
    Correlation module takes half tensors end to end (forward and backward on the half matrix-core kernels)."""
    import fn2_capi
    from networks.correlation_package.correlation import Correlation
    g = torch.Generator().manual_seed(51)
    a = torch.randn(1, 128, 8, 16, generator=g).half()
    b = torch.randn(1, 128, 8, 16, generator=g).half()
    # (non-finite values go into input2 only: an inf in input1 that meets the zero padding of input2 is inf * 0 = nan in the
    #  reference's padded buffers, 0 in the general kernel, which tests the bounds instead -- DESIGN.md 1, deviations)
    b[0, 3, 2, 2] = float("inf"); b[0, 5, 4, 4] = float("nan"); a[0, 7, 6, 6] = 6e-8; b[0, 9, 1, 1] = 65504.0
    ad, bd = a.to(dev), b.to(dev)
    out = fn2_capi.correlation_forward(ad, bd, 20, 1, 20, 1, 2)
    ref = fn2_capi.correlation_forward(ad.float(), bd.float(), 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
    fin = torch.isfinite(ref) & (ref.abs() < 6.0e4)
    assert torch.equal(torch.isfinite(out)[fin], torch.ones_like(fin)[fin]) and int((~torch.isfinite(ref)).sum()) > 0
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert float((out.float() - ref)[fin].abs().max()) <= 2.0 ** -10 * float(ref[fin].abs().max())
    x1 = torch.randn(2, 128, 12, 16, generator=g).half().to(dev).requires_grad_(True)
    x2 = torch.randn(2, 128, 12, 16, generator=g).half().to(dev).requires_grad_(True)
    y = Correlation(20, 1, 20, 1, 2, 1)(x1, x2)
    assert y.dtype == torch.float16 and tuple(y.shape) == (2, 441, 12, 16)
    y.backward(torch.randn(y.shape, generator=g).half().to(dev))
    assert x1.grad.dtype == torch.float16 and torch.isfinite(x1.grad).all() and torch.isfinite(x2.grad).all()


def _per_channel_rel(got, ref):
    """max |err| over (batch, pixels) of each channel / max |ref| of that channel -> (C,) tensor"""
    err = (got.double() - ref).abs().amax(dim=(0, 2, 3))
    return err / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-300)


@pytest.mark.parametrize("decades,shape", [(2, (1, 256, 48, 64)), (3, (1, 256, 48, 64)), (3, (1, 128, 24, 136))])
def test_correlation_backward_per_channel_error(dev, decades, shape):
    """Channel magnitudes spread log-uniformly over 10^-decades .. 10^+decades, independently per channel and input (FlowNetC is
    built with batchNorm=False: nothing ties the scales of conv3's 256 channels together).  The reference forms fp32 products at
    any scale (correlation_cuda_kernel.cu:214-229), so the error of gradInput[:, c] relative to the largest element OF THAT
    CHANNEL must not depend on how large the other channels are: each channel within 3x of the fp32 MFMA kernel's error for it."""
    import fn2_capi
    B, C, H, W = shape                       # the last one: a map wider than 64 px (column-window kernel)
    g = torch.Generator().manual_seed(41 + decades)
    s1 = torch.pow(10.0, (torch.rand(C, generator=g) * 2 - 1) * decades).view(1, C, 1, 1)
    s2 = torch.pow(10.0, (torch.rand(C, generator=g) * 2 - 1) * decades).view(1, C, 1, 1)
    x1 = (torch.randn(B, C, H, W, generator=g) * s1).to(dev)
    x2 = (torch.randn(B, C, H, W, generator=g) * s2).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    f1, f2 = _corr_bwd_fp64(x1, x2, go)
    q1, q2 = fn2_capi.correlation_backward(x1, x2, go, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32)
    for name, algo in (("f16x2", fn2_capi.FN2_CORR_MFMA_F16X2), ("auto", fn2_capi.FN2_CORR_AUTO)):
        g1, g2 = fn2_capi.correlation_backward(x1, x2, go, 20, 1, 20, 1, 2, algo=algo)
        for which, got, q, ref in (("gradInput1", g1, q1, f1), ("gradInput2", g2, q2, f2)):
            e, e32 = _per_channel_rel(got, ref), _per_channel_rel(q, ref)
            ratio = float((e / e32.clamp_min(1e-12)).max())
            print("per-channel error, +-%d decades, %s %s: worst %.2e (fp32 MFMA %.2e), worst ratio %.2f" %
                  (decades, name, which, float(e.max()), float(e32.max()), ratio))
            assert float(e.max()) <= 3.0 * float(e32.max()) and ratio <= 8.0, (decades, name, which, float(e.max()), float(e32.max()), ratio)


def test_correlation_wide_parity_at_batch_8(dev, oracle):
    """The column-window kernels at the size scripts/wide_micro.py times them (8 x 256 x 56 x 128, Sintel conv3), against an fp64
    formulation: fp32 forward + backward and the half forward (timed at B = 8, parity-checked only at B <= 3)."""
    import fn2_capi
    B, C, H, W = 8, 256, 56, 128
    g = torch.Generator().manual_seed(77)
    x1 = torch.randn(B, C, H, W, generator=g).to(dev)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    ref = _corr_fwd_fp64(x1, x2)
    out = fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2)
    assert _rel(out, ref) <= 2e-6, _rel(out, ref)
    q = fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F32)
    assert _rel(out, ref) <= 3.0 * _rel(q, ref)
    f1, f2 = _corr_bwd_fp64(x1, x2, go)
    g1, g2 = fn2_capi.correlation_backward(x1, x2, go, 20, 1, 20, 1, 2)
    assert _rel(g1, f1) <= 2e-6 and _rel(g2, f2) <= 2e-6, (_rel(g1, f1), _rel(g2, f2))
    for n in range(B):   # every batch item on its own (a wrong item stride would hide behind the global maximum)
        assert _rel(out[n], ref[n]) <= 4e-6 and _rel(g1[n], f1[n]) <= 4e-6 and _rel(g2[n], f2[n]) <= 4e-6, n
    h1, h2 = x1.half(), x2.half()
    oh = fn2_capi.correlation_forward(h1, h2, 20, 1, 20, 1, 2)
    rh = _corr_fwd_fp64(h1.float(), h2.float())
    assert oh.dtype == torch.float16 and float((oh.double() - rh).abs().max()) <= 0.5 * 2.0 ** -10 * float(rh.abs().max()) * 1.01


def test_correlation_f16x2_huge_and_tiny_operands(dev):
    """a tile whose sample sits at 2^126 used to get the scale 2^-127 -- encoded as 0.0, every product silently 0.
    in1 ~ 2^126 against in2 ~ 2^-126: products of order 1, as the reference computes them in fp32."""
    import fn2_capi
    B, C, H, W = 1, 64, 16, 24
    g = torch.Generator().manual_seed(5)
    m1 = (torch.rand(B, C, H, W, generator=g) + 0.5) * (torch.randint(0, 2, (B, C, H, W), generator=g) * 2 - 1)   # 0.5 <= |m1| < 1.5
    m2 = torch.randn(B, C, H, W, generator=g)
    x1, x2 = torch.ldexp(m1, torch.tensor(126)).to(dev), torch.ldexp(m2, torch.tensor(-126)).to(dev)
    assert torch.isfinite(x1).all()
    ref = _corr_fwd_fp64(m1.to(dev), torch.ldexp(x2.double(), torch.tensor(126)).float())   # what x2 holds (some of it subnormal), rescaled exactly
    for algo in (fn2_capi.FN2_CORR_MFMA_F16X2, fn2_capi.FN2_CORR_AUTO):
        out = fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=algo)
        assert torch.isfinite(out).all() and float(out.abs().max()) > 1e-3 and _rel(out, ref) <= 2e-6, (float(out.abs().max()), _rel(out, ref))
    # both gradients with the inputs 2^100 apart
    a, b = torch.ldexp(m1, torch.tensor(100)).to(dev), torch.ldexp(m2, torch.tensor(-100)).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    f1, f2 = _corr_bwd_fp64(m1.to(dev), m2.to(dev), go)
    g1, g2 = fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2)
    assert _rel(torch.ldexp(g1.double(), torch.tensor(100)), f1) <= 2e-6 and _rel(torch.ldexp(g2.double(), torch.tensor(-100)), f2) <= 2e-6


def test_correlation_leakyrelu_cat_backward(dev, oracle):
    """N1, training half: the differentiable fused op against the three statements of FlowNetC.py:86-92 under autograd -- same
    concat buffer, BIT-identical gradients for both feature maps and the redirected features (the masked gradient is the same
    fp32 product s * g, the backward kernels are the same) -- and against the oracle on the masked gradient."""
    import fn2_capi
    from networks.correlation_package.correlation import Correlation, CorrelationLeakyReLUCat
    for (B, C, H, W, Cr) in ((2, 64, 16, 24, 8), (1, 256, 48, 64, 32), (2, 64, 56, 72, 32)):
        g = torch.Generator().manual_seed(B * 1000 + W)
        a0 = torch.randn(B, C, H, W, generator=g)
        b0 = torch.randn(B, C, H, W, generator=g)
        r0 = torch.randn(B, Cr, H, W, generator=g)
        gbuf = torch.randn(B, Cr + 441, H, W, generator=g).to(dev)
        leaves = lambda: [t.to(dev).requires_grad_() for t in (a0, b0, r0)]
        a, b, r = leaves()
        fused = CorrelationLeakyReLUCat(20, 1, 20, 1, 2, negative_slope=0.1)(a, b, r)
        fused.backward(gbuf)
        a2, b2, r2 = leaves()
        plain = torch.cat((r2, torch.nn.functional.leaky_relu(Correlation(20, 1, 20, 1, 2, 1)(a2, b2), 0.1)), 1)
        plain.backward(gbuf)
        assert torch.equal(fused, plain)
        assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad) and torch.equal(r.grad, r2.grad), (B, C, H, W)
        # the C ABI entry point directly (ctypes), and the oracle on the masked gradient
        g1, g2 = fn2_capi.correlation_backward_fused(a.detach(), b.detach(), fused.detach(), gbuf, Cr, 0.1, 20, 1, 20, 1, 2)
        assert torch.equal(g1, a.grad) and torch.equal(g2, b.grad)
        if C * H * W <= 64 * 16 * 24 * 4:
            out = fused.detach()[:, Cr:].cpu().numpy()
            gm = gbuf[:, Cr:].cpu().numpy()
            gm = np.where(out > 0, gm, gm * np.float32(0.1)).astype(np.float32)
            o1, o2 = oracle.corr_bwd(a0.numpy(), b0.numpy(), np.ascontiguousarray(gm), 20, 1, 20, 1, 2)
            assert max_abs(g1.cpu().numpy(), o1) <= TOL and max_abs(g2.cpu().numpy(), o2) <= TOL
    # rejected: a slope that does not keep the sign, a workspace that is too small
    import ctypes
    lib = fn2_capi.lib()
    a, b = a.detach(), b.detach()
    ws = torch.empty(16, device=dev)
    args = lambda slope, w, wb: (fn2_capi._p(a), fn2_capi._p(b), fn2_capi._p(fused), ctypes.c_int64(fused.shape[1] * H * W), fn2_capi._p(gbuf),
                                 ctypes.c_int64(fused.shape[1] * H * W), ctypes.c_float(slope), fn2_capi._p(w), ctypes.c_size_t(wb), fn2_capi._p(g1), fn2_capi._p(g2),
                                 0, B, C, H, W, 20, 1, 20, 1, 2, 0, None)
    assert lib.fn2_correlation_backward_fused(*args(0.0, ws, 64)) == -1 and lib.fn2_correlation_backward_fused(*args(-0.1, ws, 64)) == -1
    assert lib.fn2_correlation_backward_fused(*args(0.1, ws, 64)) == -1


def test_correlation_leakyrelu_cat_backward_half(dev):
    """`fn2_correlation_backward_fused` for half tensors (the mask pass has a half instantiation; the backward is the half
    matrix-core kernel): identical to masking by hand and calling `fn2_correlation_backward` on the same half tensors."""
    import fn2_capi
    for (B, C, H, W, Cr) in ((1, 128, 16, 24, 8), (2, 64, 12, 40, 32)):
        g = torch.Generator().manual_seed(C + W)
        a = torch.randn(B, C, H, W, generator=g).half().to(dev)
        b = torch.randn(B, C, H, W, generator=g).half().to(dev)
        buf = torch.randn(B, Cr + 441, H, W, generator=g).half().to(dev)      # stands for the forward's concat buffer: only its sign is used
        gbuf = torch.randn(B, Cr + 441, H, W, generator=g).half().to(dev)
        g1, g2 = fn2_capi.correlation_backward_fused(a, b, buf, gbuf, Cr, 0.1, 20, 1, 20, 1, 2)
        out, gs = buf[:, Cr:].float(), gbuf[:, Cr:].float()
        masked = torch.where(out > 0, gs, gs * 0.1).half().contiguous()       # the same fp32 product, rounded to half once
        r1, r2 = fn2_capi.correlation_backward(a, b, masked, 20, 1, 20, 1, 2)
        assert torch.equal(g1, r1) and torch.equal(g2, r2), (B, C, H, W)


def test_correlation_half_backward_wide_map_fused_equals_unfused(dev):
    """`backward_fused` takes the widened column-window path for half tensors on maps wider than 64 px exactly as
    `backward` does: the fused training path and autograd's three passes give the same half gradients on a Sintel-size map."""
    import correlation_cuda
    B, C, H, W, Cr = 1, 64, 16, 72, 8
    g = torch.Generator().manual_seed(92)
    a = torch.randn(B, C, H, W, generator=g).half().to(dev)
    b = torch.randn(B, C, H, W, generator=g).half().to(dev)
    buf = torch.randn(B, Cr + 441, H, W, generator=g).half().to(dev)
    gbuf = torch.randn(B, Cr + 441, H, W, generator=g).half().to(dev)
    e = a.new_empty
    f1, f2 = e(0), e(0)
    correlation_cuda.backward_fused(a, b, buf, gbuf, Cr, 0.1, f1, f2, 20, 1, 20, 1, 2)
    out, gs = buf[:, Cr:].float(), gbuf[:, Cr:].float()
    masked = torch.where(out > 0, gs, gs * 0.1).half().contiguous()
    u1, u2 = e(0), e(0)
    correlation_cuda.backward(a, b, e(0), e(0), masked, u1, u2, 20, 1, 20, 1, 2, 1)
    assert f1.dtype == torch.float16 and torch.equal(f1, u1) and torch.equal(f2, u2)


def test_correlation_leakyrelu_cat_backward_double_and_slope_check(dev):
    """the mask pass computes in double for double tensors (it used to round every gradient through fp32) and tests the
    stored output in its own type (a value below the float range is still positive); a slope that does not keep the sign is
    refused where the module is built, not in backward()."""
    import fn2_capi
    from networks.correlation_package.correlation import CorrelationLeakyReLUCat
    B, C, H, W, Cr = 1, 32, 8, 16, 4
    g = torch.Generator().manual_seed(5)
    a = torch.randn(B, C, H, W, generator=g, dtype=torch.float64).to(dev)
    b = torch.randn(B, C, H, W, generator=g, dtype=torch.float64).to(dev)
    buf = torch.randn(B, Cr + 441, H, W, generator=g, dtype=torch.float64).to(dev)
    buf[0, Cr, 0, :4] = torch.tensor([1e-300, -1e-300, 1e-60, -1e-60], dtype=torch.float64)
    gbuf = torch.randn(B, Cr + 441, H, W, generator=g, dtype=torch.float64).to(dev)
    g1, g2 = fn2_capi.correlation_backward_fused(a, b, buf, gbuf, Cr, 0.1, 20, 1, 20, 1, 2)
    masked = torch.where(buf[:, Cr:] > 0, gbuf[:, Cr:], gbuf[:, Cr:] * float(np.float32(0.1))).contiguous()
    r1, r2 = fn2_capi.correlation_backward(a, b, masked, 20, 1, 20, 1, 2)
    assert torch.equal(g1, r1) and torch.equal(g2, r2)
    for slope in (0.0, -0.1):
        with pytest.raises(ValueError):
            CorrelationLeakyReLUCat(20, 1, 20, 1, 2, negative_slope=slope)


def test_correlation_half_backward_wide_map_through_module(dev, oracle):
    """Half tensors on a map wider than 64 px: the pybind module widens to fp32 around the column-window kernel instead of taking
    the general one-lane-per-output kernel (9 ms at 8 x 256 x 56 x 128).  Against the oracle on the half-rounded inputs, within
    half an ulp of each gradient's largest element (the result is rounded to half once)."""
    import correlation_cuda
    B, C, H, W = 1, 64, 16, 72
    g = torch.Generator().manual_seed(91)
    a = torch.randn(B, C, H, W, generator=g).half()
    b = torch.randn(B, C, H, W, generator=g).half()
    go = torch.randn(B, 441, H, W, generator=g).half()
    r1, r2 = oracle.corr_bwd(a.float().numpy(), b.float().numpy(), go.float().numpy(), 20, 1, 20, 1, 2)
    ad, bd, god = a.to(dev), b.to(dev), go.to(dev)
    e = ad.new_empty
    g1, g2, s1, s2 = e(0), e(0), e(0), e(0)
    correlation_cuda.backward(ad, bd, s1, s2, god, g1, g2, 20, 1, 20, 1, 2, 1)
    assert g1.dtype == torch.float16 and tuple(g1.shape) == (B, C, H, W)
    for got, ref in ((g1, r1), (g2, r2)):
        assert max_abs(got.float().cpu().numpy(), ref) <= 2.0 ** -10 * float(np.abs(ref).max()) + 1e-6
