"""SURVEY.md 8(b) "streams / sync" and "threading": the modules launch on the caller's CURRENT stream, keep no global mutable
state and are called from one Python worker thread per device or from the autograd engine's thread (reference:
`correlation_cuda.cc:76,158`, `resample2d_kernel.cu:221`, `channelnorm_kernel.cu:113`; `main.py:189,200` DataParallel).
Two threads, each on a stream of its own with its own inputs, run all three layers forward and backward several times at
once; every result must be the one the same call gives alone on the default stream (the correlation and ChannelNorm kernels
are deterministic: bit-identical; Resample2d's grad_img is summed with atomics: 1e-5)."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def layers(in1, in2, gcorr, img, flow, gwarp, gnorm):
    import channelnorm_cuda
    import correlation_cuda
    import resample2d_cuda
    e = in1.new_empty
    s1, s2, out, g1, g2 = e(0), e(0), e(0), e(0), e(0)
    p = (20, 1, 20, 1, 2, 1)
    assert correlation_cuda.forward(in1, in2, s1, s2, out, *p) == 1
    assert correlation_cuda.backward(in1, in2, s1, s2, gcorr, g1, g2, *p) == 1
    warped = torch.zeros_like(img)
    assert resample2d_cuda.forward(img, flow, warped, 1, True) == 1
    gimg, gflow = torch.zeros_like(img), torch.zeros_like(flow)
    assert resample2d_cuda.backward(img, flow, gwarp, gimg, gflow, 1, True) == 1
    norm = torch.zeros_like(img[:, :1])
    assert channelnorm_cuda.forward(warped, norm, 2) == 1
    gdiff = torch.zeros_like(img)
    assert channelnorm_cuda.backward(warped, norm, gnorm, gdiff, 2) == 1
    return dict(out=out, g1=g1, g2=g2, warped=warped, gimg=gimg, gflow=gflow, norm=norm, gdiff=gdiff)


def make(seed, dev, B, C, H, W, HI, WI):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    return (r(B, C, H, W), r(B, C, H, W), r(B, 441, H, W), r(B, 3, HI, WI) * 0.5, r(B, 2, HI, WI) * 4.0, r(B, 3, HI, WI),
            r(B, 1, HI, WI))


@pytest.mark.parametrize("shape", [(2, 256, 48, 64, 384, 512), (1, 128, 24, 72, 96, 160)])
def test_two_threads_two_streams(shape):
    dev = torch.device("cuda:0")
    args = [make(11 + i, dev, *shape) for i in range(2)]
    alone = [layers(*a) for a in args]
    torch.cuda.synchronize()
    results, errors = [None, None], []
    start = threading.Barrier(2)

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            st.wait_stream(torch.cuda.default_stream(dev))
            with torch.cuda.stream(st):
                start.wait()
                for _ in range(6):
                    r = layers(*args[i])
                st.synchronize()
            results[i] = r
        except Exception as e:   # surfaced in the main thread
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for i in range(2):
        for k, v in alone[i].items():
            if k == "gimg":
                assert float((results[i][k] - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), (i, k)
            else:
                assert torch.equal(results[i][k], v), (i, k)


def test_side_stream_is_the_one_used():
    """A call made under `torch.cuda.stream(s)` must be ordered on s: the kernel has to see a value that only a preceding
    operation on s produces, and the default stream must not have to be synchronised for the result to be complete."""
    import channelnorm_cuda
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(device=dev)
    x = torch.zeros(4, 3, 384, 512, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(20):
            x.add_(1.0)            # queued on s; a kernel launched on another stream would race with these
        out = torch.zeros(4, 1, 384, 512, device=dev)
        assert channelnorm_cuda.forward(x, out, 2) == 1
        s.synchronize()
    assert torch.equal(out, torch.full_like(out, 20.0 * 3 ** 0.5).float()) or float((out - 20.0 * 3 ** 0.5).abs().max()) < 1e-4


def test_layers_replay_from_a_hip_graph():
    """The six launches of a step can be captured into a hipGraph (no synchronisation, no host-side state, no allocation the
    capture cannot own) and replayed on new input VALUES in the same buffers; each replay equals the eager calls."""
    dev = torch.device("cuda:0")
    shape = (2, 256, 48, 64, 192, 256)
    static = list(make(21, dev, *shape))
    layers(*static)                      # warm-up outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = layers(*static)
    for seed in (22, 23):
        fresh = make(seed, dev, *shape)
        for dst, src in zip(static, fresh):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        eager = layers(*fresh)
        torch.cuda.synchronize()
        for k, v in eager.items():
            if k == "gimg":
                assert float((captured[k] - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), (seed, k)
            else:
                assert torch.equal(captured[k], v), (seed, k)


def test_modules_and_loss_replay_from_a_hip_graph():
    """Round 6: the MODULE path -- the C++ autograd nodes of Correlation / Resample2d / ChannelNorm, the fused rows and the MultiScale loss
    node (whose workspace is created, zero-filled and cached inside the capture) -- forward AND backward inside one hipGraph: warm-up on a
    side stream, capture, then replays on new input values in the same buffers must equal the eager modules."""
    from losses_fused import MultiScale
    from networks.channelnorm_package.channelnorm import ChannelNorm
    from networks.correlation_package.correlation import Correlation, CorrelationLeakyReLUCat
    from networks.resample2d_package.resample2d import Resample2d, WarpDiffNormCat
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    B, C, H, W, HI, WI = 2, 64, 24, 32, 128, 192
    a, b, redir = r(B, C, H, W).requires_grad_(), r(B, C, H, W).requires_grad_(), r(B, 8, H, W).requires_grad_()
    pair, flow = r(B, 6, HI, WI), (r(B, 2, HI, WI) * 3).requires_grad_()
    target = r(B, 2, HI, WI) * 5
    outs = [(r(B, 2, HI // (4 << i), WI // (4 << i)) * 0.3).requires_grad_() for i in range(5)]
    leaves = [a, b, redir, flow] + outs
    corr, fused = Correlation(20, 1, 20, 1, 2, 1), CorrelationLeakyReLUCat(20, 1, 20, 1, 2, 0.1)
    warp, norm, wcat, crit = Resample2d(), ChannelNorm(), WarpDiffNormCat(20.0), MultiScale(None)

    def fwd_bwd():
        for t in leaves:
            t.grad = None
        o = corr(a, b)
        f = fused(a, b, redir)
        n = norm(pair[:, :3] - warp(pair[:, 3:].contiguous(), flow))
        c = wcat(pair, flow)
        loss, epe = crit(tuple(outs), target)
        (o.square().mean() + f.mean() + n.mean() + c.square().mean() + loss).backward()
        return [o.detach(), f.detach(), n.detach(), c.detach(), loss.detach(), epe.detach()] + [t.grad for t in leaves]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    for t in leaves:
        t.grad = None
    with torch.cuda.graph(graph):
        captured = fwd_bwd()
    for seed in (32, 33):
        g2 = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for t in [a, b, redir, pair, target] + outs:
                t.copy_(torch.randn(t.shape, generator=g2).to(dev) * (0.3 if t.shape[1] == 2 and t is not target else 1.0))
            flow.copy_(torch.randn(flow.shape, generator=g2).to(dev) * 3)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in captured]
        eager = fwd_bwd()
        torch.cuda.synchronize()
        for i, (x, y) in enumerate(zip(got, eager)):
            assert x.shape == y.shape and torch.isfinite(x).all(), (seed, i)
            assert float((x - y).abs().max()) <= 1e-5 * max(1e-6, float(y.abs().max())), (seed, i, float((x - y).abs().max()))
