"""Restated oracle vs the reference's kernels executed live under the CPU SIMT shim
(oracle/_ref/libfn2_ref.so).  Skipped where the .so has not been built; it is built in the dev
container (where /root/reference exists) and travels to the GPU box with the snapshot."""
import numpy as np
import pytest

from conftest import max_abs

CASES = [  # B, C, H, W, pad, k, md, s1, s2
    (1, 24, 6, 6, 4, 1, 4, 1, 2),
    (2, 7, 5, 9, 3, 1, 3, 1, 1),
    (1, 64, 4, 6, 6, 1, 6, 1, 2),
    (1, 9, 8, 8, 1, 1, 4, 1, 2),   # pad < md
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_corr_live(oracle, ref_oracle, case, dt):
    B, C, H, W, pad, k, md, s1, s2 = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    a = rng.standard_normal((B, C, H, W)).astype(dt)
    b = rng.standard_normal((B, C, H, W)).astype(dt)
    y0, y1 = oracle.corr_fwd(a, b, pad, k, md, s1, s2), ref_oracle.corr_fwd(a, b, pad, k, md, s1, s2)
    assert max_abs(y0, y1) == 0.0
    go = rng.standard_normal(y0.shape).astype(dt)
    g0, g1 = oracle.corr_bwd(a, b, go, pad, k, md, s1, s2), ref_oracle.corr_bwd(a, b, go, pad, k, md, s1, s2)
    assert max_abs(g0[0], g1[0]) == 0.0 and max_abs(g0[1], g1[1]) == 0.0


def test_resample_live(oracle, ref_oracle):
    rng = np.random.default_rng(5)
    img = rng.uniform(-1, 1, (2, 3, 11, 13)).astype(np.float32)
    flow = (rng.standard_normal((2, 2, 11, 13)) * 4).astype(np.float32)
    flow[1, 0, 3, 3] = 1e4
    flow[1, 1, 4, 4] = -1e4
    go = rng.standard_normal((2, 3, 11, 13)).astype(np.float32)
    for bil in (True, False):
        assert max_abs(oracle.resample_fwd(img, flow, 1, bil), ref_oracle.resample_fwd(img, flow, 1, bil)) == 0.0
    a, b = oracle.resample_bwd(img, flow, go), ref_oracle.resample_bwd(img, flow, go)
    assert max_abs(a[0], b[0]) == 0.0 and max_abs(a[1], b[1]) == 0.0


@pytest.mark.parametrize("ks", [2, 3])
def test_resample_kernel_size_live(oracle, ref_oracle, ks):
    """kernel_size > 1 (resample2d_kernel.cu:54-61, :116-123, :171-191): the reference has no bounds test on the shifted
    indices, so it is only defined where corner + offset stays inside the image -- a flow that keeps every sample at
    least kernel_size pixels away from the lower/right border.  There the oracle (which clamps) must agree bit for bit."""
    rng = np.random.default_rng(50 + ks)
    B, C, H, W = 2, 3, 12, 14
    img = rng.uniform(-1, 1, (B, C, H, W)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    tx = rng.uniform(0, W - 2 - ks, (B, H, W)).astype(np.float32)   # target sample position, floor + 1 + (ks-1) <= W - 1
    ty = rng.uniform(0, H - 2 - ks, (B, H, W)).astype(np.float32)
    flow = np.stack([tx - xs, ty - ys], axis=1).astype(np.float32)
    go = rng.standard_normal((B, C, H, W)).astype(np.float32)
    assert max_abs(oracle.resample_fwd(img, flow, ks, True), ref_oracle.resample_fwd(img, flow, ks, True)) == 0.0
    a, b = oracle.resample_bwd(img, flow, go, ks), ref_oracle.resample_bwd(img, flow, go, ks)
    assert max_abs(a[0], b[0]) == 0.0 and max_abs(a[1], b[1]) == 0.0


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_chnorm_live(oracle, ref_oracle, dt):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 4, 5, 6)).astype(dt)
    x[1, :, 2, 2] = 0
    y0, y1 = oracle.chnorm_fwd(x), ref_oracle.chnorm_fwd(x)
    assert max_abs(y0, y1) == 0.0
    go = rng.standard_normal(y0.shape).astype(dt)
    assert max_abs(oracle.chnorm_bwd(x, y0, go), ref_oracle.chnorm_bwd(x, y0, go)) == 0.0
