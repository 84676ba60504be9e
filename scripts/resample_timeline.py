"""Timeline of resample_bwd_c3x's workgroups (debug flag 0x10000: wall-clock stamps of every workgroup's phases, 100 MHz):
when do the flushes start, how long does a flush take in the first and in the second round of workgroups, where is the chip idle?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
import numpy as np
dbg = fn2_capi.debug_lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 3, 384, 512
img = (torch.rand(B, C, H, W, generator=g) - 0.5).to(dev)
flow = torch.randn(B, 2, H, W, generator=g) * 4.0
idx = torch.randint(0, flow.numel(), (flow.numel() // 100,), generator=g)
flow.view(-1)[idx] *= 20.0
flow = flow.to(dev)
gout = torch.randn(B, C, H, W, generator=g).to(dev)
gimg = torch.zeros_like(img); gflow = torch.zeros(B, 2, H, W, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
stamps = torch.zeros(768 * 8, dtype=torch.int64, device=dev)
dbg.fn2_debug_set_buffer(P(stamps))
for flags, name in ((0x10000, "full kernel"), (0x10200, "no flush"), (0x10400, "no scatter")):
    for _ in range(3):
        gimg.zero_()
        dbg.fn2_debug_resample2d_backward(P(img), None, P(flow), P(gout), P(gimg), P(gflow), B, C, H, W, H, W, 1, 1, flags, st)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(768, 8).astype(np.float64)
    t0 = t[:, 0].min()
    us = (t[:, :6] - t0) / 100.0          # 100 MHz -> us
    order = np.argsort(us[:, 0])
    first = us[:, 0] < 2.0
    print(name, ": %d workgroups start within 2 us, the rest from %.1f us on; kernel ends at %.1f us" % (first.sum(), us[~first, 0].min() if (~first).any() else -1, us[:, 5].max()))
    names = ["start", "loads + zeroing + barrier", "scatter done (barrier)", "flush issued", "windows staged (barrier)", "end (queue drained)"]
    for grp, lab in ((first, "first round"), (~first, "second round")):
        if not grp.any(): continue
        print("  ", lab, "(%d workgroups)" % grp.sum())
        for i, nm in enumerate(names):
            v = us[grp, i]
            d = v - us[grp, i - 1] if i else v
            print("      %-30s at %6.1f us (min %6.1f max %6.1f)   phase %5.1f us (min %5.1f max %5.1f)" % (nm, v.mean(), v.min(), v.max(), d.mean(), d.min(), d.max()))
    # how many workgroups are inside their flush (stamps 2 -> 4) over time
    grid = np.arange(0, us[:, 5].max() + 1, 2.0)
    busy = [(int(((us[:, 2] <= x) & (us[:, 4] > x)).sum())) for x in grid]
    print("   workgroups between 'scatter done' and 'windows staged' (flush in progress) every 2 us:", busy)


# ---- forward (resample_fwd_tiled_all, 768 workgroups)
out = torch.zeros_like(img)
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); fn(); e_.record(); ev.append((s_, e_))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]
smooth = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 30, 31, 1, 15).to(dev)
for fl, name in ((flow, "white-noise flow"), (smooth, "smooth flow")):
    ref = None
    for flags, lab, nwg in ((0, "768 tiles", 768),):
        t_us = timeit(lambda: dbg.fn2_debug_resample2d_forward(P(img), None, P(fl), P(out), B, C, H, W, H, W, 1, 1, flags, st))
        stamps.zero_()
        dbg.fn2_debug_resample2d_forward(P(img), None, P(fl), P(out), B, C, H, W, H, W, 1, 1, flags | 0x10000, st)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        t = stamps.cpu().numpy().reshape(768, 8)[:nwg].astype(np.float64)
        us = (t[:, :5] - t[:, 0].min()) / 100.0
        print("forward, %s, %s: %.1f us by event pair; inside the kernel %d workgroups, first tile: windows in LDS at %.1f us, barrier %.1f, stores issued %.1f; "
              "workgroups end at %.1f us on average, the last at %.1f us;  max |d| vs the plain grid %.1e" % (
                  name, lab, t_us, nwg, us[:, 1].mean(), us[:, 2].mean(), us[:, 3].mean(), us[:, 4].mean(), us[:, 4].max(), float((out - ref).abs().max())))
