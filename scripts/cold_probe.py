"""Cold-operand probe: the correlation forward / backward timed (HIP events) back-to-back and with a cache-thrashing fill
(1 GiB memset) between launches -- do cold operands explain the slow boxes of the pool (bench.py context)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 256, 48, 64
a, b = torch.randn(B, C, H, W, generator=g).to(dev), torch.randn(B, C, H, W, generator=g).to(dev)
go = torch.randn(B, 441, H, W, generator=g).to(dev)
out = torch.empty(B, 441, H, W, device=dev); g1, g2 = torch.empty_like(a), torch.empty_like(b)
junk = torch.empty(1 << 28, device=dev)
def timed(fn, thrash, n=15):
    ts = []
    for _ in range(n):
        if thrash: junk.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
fwd = lambda: fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, out=out)
bwd = lambda: fn2_capi.correlation_backward(a, b, go, 20, 1, 20, 1, 2, out=(g1, g2))
for _ in range(5): fwd(); bwd()
print("fwd  warm %.1f us   after 1 GiB fill %.1f us" % (timed(fwd, False), timed(fwd, True)))
print("bwd  warm %.1f us   after 1 GiB fill %.1f us" % (timed(bwd, False), timed(bwd, True)))
small = torch.empty(1 << 24, device=dev)   # 64 MiB fill: evicts L2 (32 MiB) but not a 256 MiB infinity cache
def timed2(fn, n=15):
    ts = []
    for _ in range(n):
        small.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
print("fwd  after 64 MiB fill %.1f us    bwd %.1f us" % (timed2(fwd), timed2(bwd)))
