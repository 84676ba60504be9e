"""Per-step and per-task cost of the wide forward kernel: time at C = 128 / 256 / 512 (4 / 8 / 16 steps per task) on 8 x C x 56 x 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]
res = {}
for (H, W) in ((56, 128), (48, 64)):
    for C in (128, 256, 512):
        g = torch.Generator().manual_seed(0)
        a = torch.randn(8, C, H, W, generator=g).to(dev); b = torch.randn(8, C, H, W, generator=g).to(dev)
        out = torch.empty(8, 441, H, W, device=dev)
        res[(H, W, C)] = timeit(lambda: fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_MFMA_F16X2, out=out))
    t1, t2, t3 = res[(H, W, 128)], res[(H, W, 256)], res[(H, W, 512)]
    print("8 x C x %d x %d forward: C=128 %.1f us, C=256 %.1f us, C=512 %.1f us -> per 128 channels %.1f / %.1f us, fixed part %.1f us"
          % (H, W, t1, t2, t3, t2 - t1, (t3 - t2) / 2, t1 - (t2 - t1)))
