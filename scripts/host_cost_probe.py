"""Host cost of one call of each pybind entry point of the step (tiny tensors: the GPU work is negligible; 2000 calls, one sync at the end),
and of the same calls at the bench's shapes with the GPU saturated (enqueue rate)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-pytorch_amd")]
import torch, correlation_cuda, resample2d_cuda, channelnorm_cuda
dev = torch.device("cuda:0")
a = torch.randn(1, 64, 8, 8, device=dev); b = torch.randn(1, 64, 8, 8, device=dev); go = torch.randn(1, 441, 8, 8, device=dev)
e = a.new_empty
s1, s2, out, g1, g2 = e(0), e(0), e(0), e(0), e(0)
img = torch.randn(1, 3, 32, 64, device=dev); flow = torch.randn(1, 2, 32, 64, device=dev); w = torch.empty_like(img)
gi = torch.zeros_like(img); gf = torch.empty_like(flow); nrm = torch.empty(1, 1, 32, 64, device=dev); gn = torch.randn(1, 1, 32, 64, device=dev); gd = torch.empty_like(img)
P = (20, 1, 20, 1, 2, 1)
calls = {
    "correlation_cuda.forward": lambda: correlation_cuda.forward(a, b, s1, s2, out, *P),
    "correlation_cuda.backward": lambda: correlation_cuda.backward(a, b, s1, s2, go, g1, g2, *P),
    "resample2d_cuda.forward": lambda: resample2d_cuda.forward(img, flow, w, 1, True),
    "resample2d_cuda.backward": lambda: resample2d_cuda.backward(img, flow, w, gi, gf, 1, True),
    "channelnorm_cuda.forward": lambda: channelnorm_cuda.forward(w, nrm, 2),
    "channelnorm_cuda.backward": lambda: channelnorm_cuda.backward(w, nrm, gn, gd, 2),
    "Tensor.zero_": lambda: gi.zero_(),
    "empty python lambda": lambda: None,
}
tot = 0.0
for name, fn in calls.items():
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    us = (t1 - t0) / 2000 * 1e6
    if "lambda" not in name: tot += us
    print("%-28s %6.2f us per call (host)" % (name, us))
print("sum of the step's seven calls: %.1f us of host time per step" % tot)
