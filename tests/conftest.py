import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flownet2-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref_oracle():
    from oracle.oracle import Oracle, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libfn2_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return Oracle(ref=True)


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU (run through gpurun)"
    return torch.device("cuda:0")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def max_abs(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def graded_corr_inputs(fixture):
    """Inputs of tests/golden/corrgraded_*.npz: not stored, regenerated from the fixture's seed exactly as
    tests/golden/make_golden_corr_graded.py drew them; the stored checksum tells a different numpy generator from a wrong result."""
    B, C, H, W = (int(v) for v in fixture["shape"])
    rng = np.random.default_rng(int(fixture["seed"]))          # (the stored seed is the generator's SEED + C)
    in1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    in2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    gout = rng.standard_normal((B, 441, H, W)).astype(np.float32)
    in1[0, 3] *= 30.0; in2[0, 7] *= 1e-3
    chk = np.array([float(np.sum(a.astype(np.float64) * np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape) % 7.0)) for a in (in1, in2, gout)])
    if not np.array_equal(chk, fixture["input_checksum"]):
        pytest.skip("numpy's default_rng draws differ from the ones the fixture was generated with")
    return in1, in2, gout


def full_size_inputs(fixture):
    """Inputs of tests/golden/fullsize_*.npz, regenerated from the fixture's seed as tests/golden/make_golden_full_size.py drew them."""
    B, C, H, W = (int(v) for v in fixture["shape"])
    rng = np.random.default_rng(int(fixture["seed"]))
    img = rng.uniform(-0.5, 0.5, (B, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((B, 2, H, W)) * 4.0).astype(np.float32)
    flow.reshape(-1)[rng.integers(0, flow.size, flow.size // 100)] *= np.float32(20.0)
    gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
    gnorm = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    img[0, :, 5, 7] = 0.0
    chk = np.array([float(np.sum(a.astype(np.float64) * np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape) % 7.0)) for a in (img, flow, gout, gnorm)])
    if not np.array_equal(chk, fixture["input_checksum"]):
        pytest.skip("numpy's default_rng draws differ from the ones the fixture was generated with")
    return img, flow, gout, gnorm
