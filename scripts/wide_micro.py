"""Timing of the correlation kernels on maps wider than 64 pixels (Sintel-size conv3 maps): f16x2 windows vs the fp32 matrix-core kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]
for (B, C, H, W) in ((1, 256, 56, 128), (8, 256, 56, 128), (4, 256, 48, 64), (8, 256, 48, 64), (4, 256, 96, 128)):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(B, C, H, W, generator=g).to(dev); b = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 441, H, W, generator=g).to(dev)
    out = torch.empty(B, 441, H, W, device=dev); g1 = torch.empty_like(a); g2 = torch.empty_like(a)
    line = "B %d C %d %dx%d:" % (B, C, H, W)
    for name, algo in (("f16x2", fn2_capi.FN2_CORR_MFMA_F16X2), ("f32-mfma", fn2_capi.FN2_CORR_MFMA_F32)):
        try:
            t = timeit(lambda: fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=algo, out=out))
            line += "  fwd %s %.1f us" % (name, t)
        except Exception as ex:
            line += "  fwd %s n/a" % name
        try:
            t = timeit(lambda: fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, algo=algo, out=(g1, g2)))
            line += "  bwd %s %.1f us" % (name, t)
        except Exception as ex:
            line += "  bwd %s n/a" % name
    ah, bh = a.half(), b.half(); oh = torch.empty(B, 441, H, W, device=dev, dtype=torch.float16)
    line += "  fwd half %.1f us" % timeit(lambda: fn2_capi.correlation_forward(ah, bh, 20, 1, 20, 1, 2, out=oh))
    line += " (general kernel %.1f us)" % timeit(lambda: fn2_capi.correlation_forward(ah, bh, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT, out=oh), n=5)
    if W <= 64:
        gh = go.half(); h1 = torch.empty_like(ah); h2 = torch.empty_like(ah)
        line += "  bwd half %.1f us" % timeit(lambda: fn2_capi.correlation_backward(ah, bh, gh, 20, 1, 20, 1, 2, out=(h1, h2)))
        line += " (general kernel %.1f us)" % timeit(lambda: fn2_capi.correlation_backward(ah, bh, gh, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT, out=(h1, h2)), n=3)
    if W <= 64 and B == 8:   # double tensors: the fp64 matrix-core kernels (round 6) against the one-thread-per-output kernel
        ad, bd_, gd = a.double(), b.double(), go.double()
        od = torch.empty(B, 441, H, W, device=dev, dtype=torch.float64); d1 = torch.empty_like(ad); d2 = torch.empty_like(ad)
        line += "  fwd double %.1f us" % timeit(lambda: fn2_capi.correlation_forward(ad, bd_, 20, 1, 20, 1, 2, out=od), n=10)
        line += " (general kernel %.1f us)" % timeit(lambda: fn2_capi.correlation_forward(ad, bd_, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT, out=od), n=3)
        line += "  bwd double %.1f us" % timeit(lambda: fn2_capi.correlation_backward(ad, bd_, gd, 20, 1, 20, 1, 2, out=(d1, d2)), n=10)
        line += " (general kernel %.1f us)" % timeit(lambda: fn2_capi.correlation_backward(ad, bd_, gd, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT, out=(d1, d2)), n=2)
    print(line)
