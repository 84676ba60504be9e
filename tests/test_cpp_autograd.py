"""the autograd nodes of the wrappers live in the pybind modules (C++); `XFunction.apply` goes there directly.  The
Python static methods (`XFunction.forward` / `.backward`, the reference's shape of the API) stay and must give the same bits."""
import numpy as np
import pytest
import torch
from torch.autograd import Function

pytestmark = pytest.mark.gpu


def _python_apply(cls, *args):
    """The node defined by cls.forward / cls.backward in Python (what `apply` was before round 6)."""
    return Function.apply.__func__(cls, *args)


def _grads(fn, tensors, gout):
    leaves = [t.detach().clone().requires_grad_(t.requires_grad) for t in tensors]
    out = fn(*leaves)
    out.backward(gout)
    return out.detach(), [l.grad for l in leaves]


def test_cpp_nodes_equal_python_nodes(dev):
    from networks.channelnorm_package.channelnorm import ChannelNormFunction
    from networks.correlation_package.correlation import CorrelationFunction, CorrelationLeakyReLUCatFunction
    from networks.resample2d_package.resample2d import Resample2dFunction, WarpDiffNormCatFunction, WarpDiffNormFunction
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    a, b = rn(2, 64, 16, 24).requires_grad_(), rn(2, 64, 16, 24).requires_grad_()
    p = (20, 1, 20, 1, 2, 1)
    cases = [
        ("Correlation", CorrelationFunction, (a, b), p, rn(2, 441, 16, 24)),
        ("CorrelationLeakyReLUCat", CorrelationLeakyReLUCatFunction, (a, b, rn(2, 8, 16, 24).requires_grad_()), (20, 1, 20, 1, 2, 0.1), rn(2, 449, 16, 24)),
        ("Resample2d", Resample2dFunction, (rn(2, 3, 32, 64).requires_grad_(), (rn(2, 2, 32, 64) * 3).requires_grad_()), (1, True), rn(2, 3, 32, 64)),
        ("ChannelNorm", ChannelNormFunction, (rn(2, 3, 32, 64).requires_grad_(),), (2,), rn(2, 1, 32, 64)),
        ("WarpDiffNormCat", WarpDiffNormCatFunction, (rn(2, 6, 32, 64).requires_grad_(), (rn(2, 2, 32, 64) * 3).requires_grad_()), (20.0, True), rn(2, 12, 32, 64)),
        ("WarpDiffNorm", WarpDiffNormFunction, (rn(2, 6, 32, 64), (rn(2, 2, 32, 64) * 3).requires_grad_()), (True,), rn(2, 1, 32, 64)),
    ]
    for name, cls, tensors, params, gout in cases:
        o1, g1 = _grads(lambda *t: cls.apply(*t, *params), tensors, gout)
        o2, g2 = _grads(lambda *t: _python_apply(cls, *t, *params), tensors, gout)
        assert torch.equal(o1, o2), name
        for x, y, t in zip(g1, g2, tensors):
            assert (x is None) == (y is None) == (not t.requires_grad), name
            if x is not None:
                if name in ("Resample2d", "WarpDiffNormCat") and t is tensors[0]:    # the image gradient is summed by fp32 atomics: order varies
                    assert float((x - y).abs().max()) <= 5e-6 * float(y.abs().max()), name
                else:
                    assert torch.equal(x, y), name
    # the C++ node has a name of its own in the graph, defaults like the reference's signature, and refuses a second differentiation
    out = CorrelationFunction.apply(a, b, *p)
    assert "CorrelationOp" in out.grad_fn.name()
    import correlation_cuda
    assert torch.equal(correlation_cuda.apply(a, b, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2).detach(), out.detach())
    go = torch.ones_like(out).requires_grad_()
    with pytest.raises(RuntimeError, match="second time"):
        torch.autograd.grad(out, a, go, create_graph=True)
    # only the gradients that are asked for are computed where the kernel can skip one (WarpDiffNormCat: the pair's scatter)
    x, fl = rn(2, 6, 32, 64), (rn(2, 2, 32, 64) * 3).requires_grad_()
    WarpDiffNormCatFunction.apply(x, fl, 20.0, True).sum().backward()
    assert fl.grad is not None and x.grad is None


def test_modules_run_without_the_gil_in_backward(dev):
    """A backward pass through the C++ nodes from a thread that does not hold the GIL-dependent Python Function machinery: the autograd
    engine's device thread runs them; a Python-side lock held across .backward() must not deadlock."""
    import threading
    from networks.correlation_package.correlation import Correlation
    g = torch.Generator().manual_seed(4)
    a = torch.randn(1, 64, 16, 24, generator=g).to(dev).requires_grad_()
    b = torch.randn(1, 64, 16, 24, generator=g).to(dev).requires_grad_()
    done = []

    def run():
        Correlation(20, 1, 20, 1, 2, 1)(a, b).sum().backward()
        done.append(True)
    t = threading.Thread(target=run)
    t.start()
    t.join(60)
    assert done and a.grad is not None and torch.isfinite(a.grad).all()


def test_multiscale_node_one_launch_and_repeatable(dev):
    """N3 through multiscale_loss_cuda: loss and metric come from the kernel's last workgroup (no second launch), the cached
    workspace's ticket counter is left at zero (call after call gives the same bits), the raw-sums entry point of the C ABI (memset
    in front, any scratch) agrees, and the backward scales out of place."""
    import multiscale_loss_cuda
    import fn2_capi
    from losses_fused import MultiScale
    g = torch.Generator().manual_seed(7)
    B, H, W = 8, 384, 512
    target = (torch.randn(B, 2, H, W, generator=g) * 5).to(dev)
    outs = [(torch.randn(B, 2, H // (4 << i), W // (4 << i), generator=g) * 0.3).to(dev) for i in range(5)]
    weights = [0.32 / 2 ** i for i in range(5)]
    first = None
    for rep in range(25):
        of = [o.clone().requires_grad_(True) for o in outs]
        loss, epe = multiscale_loss_cuda.apply(target, of, 4, 0.05, weights, 1)
        (2.5 * loss).backward()
        cur = (loss.detach().clone(), epe.detach().clone(), [o.grad.clone() for o in of])
        if first is None:
            first = cur
        assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]) and all(torch.equal(x, y) for x, y in zip(cur[2], first[2])), rep
    assert not epe.requires_grad and loss.requires_grad
    sums, _ = fn2_capi.multiscale_l1_epe(outs, target, weights)                  # un-primed scratch, memset node in front of the kernel
    sums2 = multiscale_loss_cuda.sums(target, outs, 4, 0.05)
    assert torch.equal(sums, sums2)
    n = [o.numel() for o in outs]
    want_loss = sum(w * float(sums[i]) / n[i] for i, w in enumerate(weights))
    want_epe = sum(w * float(sums[5 + i]) / (n[i] / 2) for i, w in enumerate(weights))
    assert abs(float(first[0]) - want_loss) <= 2e-6 * want_loss and abs(float(first[1]) - want_epe) <= 2e-6 * want_epe
    # unit gradients x 2.5: sign(out - t) * w_i / N_i * 2.5
    for i, gr in enumerate(first[2]):
        mag = gr.abs()
        assert float((mag[mag > 0] - 2.5 * np.float32(weights[i] / n[i])).abs().max()) <= 1e-7 * weights[i] / n[i] * 2.5 + 1e-12
    # the module takes this path for GPU tensors, with and without gradients, on a side stream too (another cached workspace)
    crit = MultiScale(None)
    with torch.no_grad():
        l0, e0 = crit(tuple(outs), target)
    assert torch.equal(l0, first[0]) and torch.equal(e0, first[1])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        l1, e1 = crit(tuple(outs), target)
    s.synchronize()
    assert torch.equal(l1, first[0]) and torch.equal(e1, first[1])
