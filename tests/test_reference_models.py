"""BASELINE.json configs[0]: the reference's own, unmodified models.py / networks/*.py import and run on top
of this repo's extension modules.  Needs the reference checkout (/root/reference): dev container only, skipped on
the GPU box.  FlowNet2S uses none of the custom ops (plumbing check, CPU); models that do use them must fail
loudly on CPU because there is no CPU implementation behind the boundary."""
import importlib
import os
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import PKG

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="reference checkout not present")


@pytest.fixture()
def ref_models():
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "models"]:
        del sys.modules[k]
    sys.path.insert(0, REF)   # the reference's networks/ wrappers, importing OUR correlation_cuda etc. by bare name
    sys.path.insert(0, PKG)
    sys.path.remove(PKG)
    sys.path.insert(1, PKG)
    try:
        yield importlib.import_module("models")
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved_mods]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


def test_flownet2s_cpu_plumbing(ref_models):
    import correlation_cuda
    assert correlation_cuda.__file__.startswith(PKG)
    assert ref_models.__file__.startswith(REF)
    args = SimpleNamespace(rgb_max=255.0, fp16=False)
    torch.manual_seed(0)
    net = ref_models.FlowNet2S(args).eval()
    assert sum(p.numel() for p in net.parameters()) == 38_676_506
    x = 255.0 * torch.rand(1, 3, 2, 384, 512)
    with torch.no_grad():
        y = net(x)
    assert tuple(y.shape) == (1, 2, 384, 512) and torch.isfinite(y).all()


def test_flownet2c_constructs_and_fails_loudly_on_cpu(ref_models):
    args = SimpleNamespace(rgb_max=255.0, fp16=False)
    net = ref_models.FlowNet2C(args).eval()
    assert sum(p.numel() for p in net.parameters()) == 39_175_298   # FlowNetC.py:11
    from networks.correlation_package.correlation import Correlation
    assert type(net.corr).__name__ == "Correlation" and net.corr.max_displacement == 20
    assert sys.modules[Correlation.__module__].__file__.startswith(REF)   # the reference's own wrapper ...
    with torch.no_grad(), pytest.raises(RuntimeError, match="GPU"):       # ... on top of our extension: no CPU path
        net(255.0 * torch.rand(1, 3, 2, 64, 64))
