"""Time series of the hot-path step from process start on a fresh box: 100-step batches (wall clock, synchronised)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda:0")
hp = bench.HotPath(dev, 1234)
torch.cuda.synchronize()
t_start = time.perf_counter()
out = []
for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter()
    for _ in range(100):
        hp.step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t_start, (time.perf_counter() - t0) * 10))   # ms per step
print(" ".join("%.2fs:%.4f" % o for o in out))
