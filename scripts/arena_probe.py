"""Does the physical placement of the step's buffers matter?  Hot-path step time with torch's default allocations vs with every
buffer carved from ONE pre-allocated 6 GiB block (arg 'arena'); run each in its own process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "arena":
    big = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
    del big                        # stays in the caching allocator; the step's tensors are split off it
hp = bench.HotPath(dev, 1234)
for _ in range(20): hp.step()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(300): hp.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 300 * 1e6)
seg = torch.cuda.memory_stats(dev)
print("%-8s step %s us   segments %d  reserved %.2f GiB" % (mode, " ".join("%.1f" % t for t in ts), seg.get("segment.all.current", -1), seg.get("reserved_bytes.all.current", 0) / 2**30))
