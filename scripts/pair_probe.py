"""Which predecessor makes a kernel of the step slow?  For every (predecessor, kernel) pair: loop {predecessor; timed kernel}
and report the kernel's median HIP-event duration; first line: the whole step (classifies the box as fast / slow)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda:0")
hp = bench.HotPath(dev, 1234)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB
def res_bwd():
    hp.gimg.zero_(); hp.m_res.backward(hp.img, hp.flow, hp.gwarp, hp.gimg, hp.gflow, 1, True)
ops = {
    "corr_fwd": hp.corr_fwd, "corr_bwd": hp.corr_bwd,
    "res_fwd": lambda: hp.m_res.forward(hp.img, hp.flow, hp.warped, 1, True),
    "cn_fwd": lambda: hp.m_cn.forward(hp.warped, hp.norm, 2),
    "cn_bwd": lambda: hp.m_cn.backward(hp.warped, hp.norm, hp.gnorm, hp.gdiff, 2),
    "res_bwd": res_bwd, "fill1GiB": lambda: big.fill_(1.0), "fill64MB": lambda: big[:1 << 24].fill_(1.0),
}
for _ in range(50): hp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): hp.step()
torch.cuda.synchronize()
print("step %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6), flush=True)
def timed_after(pred, name, n=60):
    ev = []
    for _ in range(n):
        if pred: ops[pred]()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops[name](); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev[5:])
    return ts[len(ts) // 2]
for name in ("corr_fwd", "corr_bwd", "res_fwd", "res_bwd"):
    line = "%-9s after:" % name
    for pred in (None, "corr_fwd", "corr_bwd", "res_fwd", "cn_fwd", "cn_bwd", "res_bwd", "fill64MB", "fill1GiB"):
        if pred == name: continue
        line += "  %s %.1f" % (pred or "itself", timed_after(pred, name))
    print(line, flush=True)
