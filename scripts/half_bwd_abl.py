"""Times the half-precision correlation backward (8 x 256 x 48 x 64) through the library given on the command line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
fn2_capi.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 256, 48, 64
a = torch.randn(B, C, H, W, generator=g).half().to(dev); b = torch.randn(B, C, H, W, generator=g).half().to(dev)
go = torch.randn(B, 441, H, W, generator=g).half().to(dev)
g1 = torch.empty_like(a); g2 = torch.empty_like(a)
fn = lambda: fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, out=(g1, g2))
for _ in range(3): fn()
torch.cuda.synchronize()
ev = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); ev.append((s, e))
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
print("half bwd %.1f us (median of 20)" % ts[10])
