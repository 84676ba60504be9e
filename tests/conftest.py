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
