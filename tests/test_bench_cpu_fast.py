"""bench.py's `cpu_baseline_fast` leg: the vectorised PyTorch formulation of the step (441 shifted channel contractions with
explicit backward, grid_sample + autograd, closed-form norm gradient) computes what the oracle computes -- to tolerance: its
summation order is whatever the vectorised kernels choose (reference semantics: correlation_cuda_kernel.cu:73-334,
resample2d_kernel.cu:15-198, channelnorm_kernel.cu:18-96)."""
import numpy as np
import torch

import bench


def _d(t, ref):
    return float(np.abs(t.detach().numpy().astype(np.float64) - ref).max())


def test_torch_formulation_matches_oracle(oracle):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(2, 16, 12, 16, generator=g)
    b = torch.randn(2, 16, 12, 16, generator=g)
    go = torch.randn(2, 441, 12, 16, generator=g)
    p = (20, 1, 20, 1, 2)
    assert _d(bench.torch_corr_fwd(a, b), oracle.corr_fwd(a.numpy(), b.numpy(), *p)) <= 2e-6
    g1, g2 = bench.torch_corr_bwd(a, b, go)
    r1, r2 = oracle.corr_bwd(a.numpy(), b.numpy(), go.numpy(), *p)
    assert _d(g1, r1) <= 5e-6 and _d(g2, r2) <= 5e-6
    img = torch.rand(2, 3, 24, 32, generator=g) - 0.5
    flow = torch.randn(2, 2, 24, 32, generator=g) * 4
    gw = torch.randn(2, 3, 24, 32, generator=g)
    w = bench.torch_resample_fwd(img, flow)
    assert _d(w, oracle.resample_fwd(img.numpy(), flow.numpy())) <= 1e-5
    gi, gf = bench.torch_resample_bwd(img, flow, gw)
    ri, rf = oracle.resample_bwd(img.numpy(), flow.numpy(), gw.numpy())
    assert _d(gi, ri) <= 1e-4 and _d(gf, rf) <= 1e-4
    n = bench.torch_chnorm_fwd(w)
    assert _d(n, oracle.chnorm_fwd(w.numpy())) <= 1e-6
    gn = torch.randn(2, 1, 24, 32, generator=g)
    assert _d(bench.torch_chnorm_bwd(w, n, gn), oracle.chnorm_bwd(w.numpy(), n.numpy(), gn.numpy())) <= 1e-5


def test_cpu_baseline_fast_line():
    r = bench.cpu_baseline_fast(max_seconds=0.0)          # one warm-up + one timed step
    assert r["kind"] == "port" and r["unit"] == "image-pairs/s" and r["value"] > 0 and r["cores"] >= 1
