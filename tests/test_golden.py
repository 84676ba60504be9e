"""The restated oracle (oracle/fn2_oracle.c) against the committed golden vectors, which were
produced by the reference's own kernels under the CPU SIMT shim (tests/golden/make_golden.py).
The restatement keeps the reference's operation order, so agreement is bit-exact."""
import os

import numpy as np
import pytest

from conftest import full_size_inputs, golden_files, graded_corr_inputs, max_abs


@pytest.mark.parametrize("path", golden_files("corr"), ids=os.path.basename)
def test_corr_golden(oracle, path):
    g = np.load(path)
    pad, k, md, s1, s2 = (int(v) for v in g["params"])
    out = oracle.corr_fwd(g["in1"], g["in2"], pad, k, md, s1, s2)
    assert out.dtype == g["out"].dtype
    assert max_abs(out, g["out"]) == 0.0
    if "g1" in g.files:
        g1, g2 = oracle.corr_bwd(g["in1"], g["in2"], g["gout"], pad, k, md, s1, s2)
        assert max_abs(g1, g["g1"]) == 0.0
        assert max_abs(g2, g["g2"]) == 0.0


@pytest.mark.parametrize("path", golden_files("resample"), ids=os.path.basename)
def test_resample_golden(oracle, path):
    g = np.load(path)
    for bil in (1, 0):
        assert max_abs(oracle.resample_fwd(g["img"], g["flow"], 1, bool(bil)), g[f"out_bil{bil}"]) == 0.0
    gimg, gflow = oracle.resample_bwd(g["img"], g["flow"], g["gout"], 1, True)
    assert max_abs(gimg, g["gimg"]) == 0.0   # same (thread-index) accumulation order as the SIMT run
    assert max_abs(gflow, g["gflow"]) == 0.0


@pytest.mark.parametrize("path", golden_files("chnorm"), ids=os.path.basename)
def test_chnorm_golden(oracle, path):
    g = np.load(path)
    out = oracle.chnorm_fwd(g["x"])
    assert max_abs(out, g["out"]) == 0.0
    assert max_abs(oracle.chnorm_bwd(g["x"], g["out"], g["gout"]), g["gin"]) == 0.0


@pytest.mark.parametrize("path", golden_files("corrgraded"), ids=os.path.basename)
def test_corr_graded_geometry_golden(oracle, path):
    """The fixture of the graded geometry (48 x 64 map, FlowNetC's parameters; inputs by seed, the reference's results as sampled
    planes / channels + float64 sums of all of them): the restated oracle reproduces the reference's device code bit for bit."""
    g = np.load(path)
    in1, in2, gout = graded_corr_inputs(g)
    pad, k, md, s1, s2 = (int(v) for v in g["params"])
    out = oracle.corr_fwd(in1, in2, pad, k, md, s1, s2)
    assert max_abs(out[:, g["planes"]], g["out_planes"]) == 0.0
    o64 = out.astype(np.float64)
    assert np.array_equal(o64.sum(axis=(0, 2, 3)), g["out_sum"]) and np.array_equal((o64 * o64).sum(axis=(0, 2, 3)), g["out_sumsq"])
    g1, g2 = oracle.corr_bwd(in1, in2, gout, pad, k, md, s1, s2)
    assert max_abs(g1[:, g["channels"]], g["g1_channels"]) == 0.0 and max_abs(g2[:, g["channels"]], g["g2_channels"]) == 0.0
    assert np.array_equal(g1.astype(np.float64).sum(axis=(0, 2, 3)), g["g1_sum"])
    assert np.array_equal(g2.astype(np.float64).sum(axis=(0, 2, 3)), g["g2_sum"])


@pytest.mark.parametrize("path", golden_files("fullsize"), ids=os.path.basename)
def test_full_size_resample_chnorm_golden(oracle, path):
    """Resample2d and ChannelNorm at the BASELINE shape (8 x 3 x 384 x 512, the SURVEY's flow): the restated oracle reproduces the
    reference's device code bit for bit on the kept rows and on the float64 sums of every plane."""
    g = np.load(path)
    img, flow, gout, gnorm = full_size_inputs(g)
    rows = g["rows"]

    def same(name, a):
        assert max_abs(a[:, :, rows], g[name + "_rows"]) == 0.0, name
        assert np.array_equal(a.astype(np.float64).sum(axis=(2, 3)), g[name + "_sum"]), name
    same("warp", oracle.resample_fwd(img, flow, 1, True))
    same("warp_nearest", oracle.resample_fwd(img, flow, 1, False))
    gi, gf = oracle.resample_bwd(img, flow, gout, 1, True)
    same("gimg", gi)
    same("gflow", gf)
    n = oracle.chnorm_fwd(img)
    same("norm", n)
    same("gnorm_in", oracle.chnorm_bwd(img, n, gnorm))


def test_golden_present():
    assert len(golden_files("corr")) >= 5 and len(golden_files("resample")) >= 4 and len(golden_files("chnorm")) >= 3
