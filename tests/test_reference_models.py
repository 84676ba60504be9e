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


@pytest.mark.gpu
def test_reference_models_run_on_the_hip_modules(ref_models):
    """The literal north_star clause, for a machine that has BOTH the reference checkout and an MI355X (not this pool: the GPU boxes
    have no /root/reference and the reference's Python may not be shipped to them -- there tests/test_reference_model_golden.py holds
    what the reference's models produced; this test is skipped).  The reference's unmodified models.FlowNet2C / FlowNet2, its own
    wrapper modules and THIS repo's correlation_cuda / resample2d_cuda / channelnorm_cuda, against harness.* with one state dict."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib
    dev = torch.device("cuda:0")
    args = SimpleNamespace(rgb_max=255.0, fp16=False)
    saved = {k: v for k, v in sys.modules.items() if k == "networks" or k.startswith("networks.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, PKG)
    try:
        for k in [k for k in sys.modules if k.startswith("harness")]:
            del sys.modules[k]
        f2c = importlib.import_module("harness.flownet2c")
        f2 = importlib.import_module("harness.flownet2")
    finally:
        sys.path.remove(PKG)
        for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    g = torch.Generator().manual_seed(0)
    x = (255.0 * torch.rand(8, 3, 2, 384, 512, generator=g)).to(dev)
    torch.manual_seed(1)
    ref = ref_models.FlowNet2C(args).to(dev).eval()
    ours = f2c.FlowNet2C(fused_inference=False, fused_training=False).to(dev).eval()
    ours.load_state_dict(ref.state_dict())
    with torch.no_grad():
        a, b = ref(x), ours(x)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    ref2 = ref_models.FlowNet2(args).to(dev).eval()
    ours2 = f2.FlowNet2().to(dev).eval()
    ours2.load_state_dict(ref2.state_dict())
    with torch.no_grad():
        a, b = ref2(x), ours2(x)
    assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
