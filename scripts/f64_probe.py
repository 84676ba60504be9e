import sys; sys.path.insert(0, "flownet2-pytorch_amd")
import torch, fn2_capi
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for B in (8, 16):
    for C in (64, 128, 256, 512):
        a = torch.randn(B, C, 48, 64, device=dev, dtype=torch.float64); b = torch.randn_like(a)
        go = torch.randn(B, 441, 48, 64, device=dev, dtype=torch.float64)
        out = torch.empty(B, 441, 48, 64, device=dev, dtype=torch.float64); g1 = torch.empty_like(a); g2 = torch.empty_like(a)
        tf = timeit(lambda: fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, out=out))
        tb = timeit(lambda: fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, out=(g1, g2)))
        print("B %2d C %3d: fwd %.1f us  bwd %.1f us" % (B, C, tf, tb))
