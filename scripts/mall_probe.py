"""Effective last-level (Infinity Cache / MALL) reach of THIS box: read bandwidth of repeated passes over a buffer of S MB
(torch.sum, read-only) and of copies (read S/2 + write S/2) as S grows; first line: the hot-path step (fast / slow box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda:0")
hp = bench.HotPath(dev, 1234)
for _ in range(50): hp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): hp.step()
torch.cuda.synchronize()
print("step %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6), flush=True)
def med(fn, n=24):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e-3
rd, cp = [], []
for mb in (16, 32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 384, 512, 1024):
    print("  size", mb, flush=True)
    x = torch.ones(mb << 18, dtype=torch.float32, device=dev)
    t = med(lambda: x.sum())
    rd.append("%d:%.2f" % (mb, (mb << 20) / t / 1e12))
    h = x.numel() // 2
    a, b = x[:h], x[h:]
    t = med(lambda: b.copy_(a))
    cp.append("%d:%.2f" % (mb, (mb << 20) / t / 1e12))
    del x, a, b
print("read  TB/s by MB: " + " ".join(rd))
print("copy  TB/s by MB (read+write bytes): " + " ".join(cp))
