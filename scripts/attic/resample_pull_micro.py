"""Resample2d backward: the owner-tile ("pull") kernel against the push kernel (fn2_debug_resample2d_backward, selector bits 12-15;
0xB000 = pull, accumulate; 0xD000 = pull, overwrite).  Timings at 8 x 3 x 384 x 512 and correctness on ragged shapes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
lib = fn2_capi.debug_lib()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]


def make(B, C, H, W, seed, kind):
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(B, C, H, W, generator=g) - 0.5).to(dev)
    if kind == "noise":
        flow = torch.randn(B, 2, H, W, generator=g) * 4.0
        idx = torch.randint(0, flow.numel(), (max(flow.numel() // 100, 1),), generator=g)
        flow.view(-1)[idx] *= 20.0
    elif kind == "smooth":
        flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 30, 31, 1, 15)
    else:   # large: most corners are far
        flow = torch.randn(B, 2, H, W, generator=g) * 40.0
    return img, flow.to(dev), torch.randn(B, C, H, W, generator=g).to(dev)


def bwd(img, flow, gout, gimg, gflow, sel):
    B, C, H, W = img.shape
    return lib.fn2_debug_resample2d_backward(P(img), None, P(flow), P(gout), P(gimg), P(gflow), B, C, H, W, H, W, 1, 1, sel, st)


print("correctness against the push kernel (grad_img pre-filled with NaN for the overwriting variant)")
worst = 0.0
for (B, C, H, W) in ((1, 3, 16, 32), (2, 3, 100, 200), (1, 1, 33, 68), (3, 2, 64, 64), (1, 3, 384, 512), (2, 3, 97, 260)):
    for kind in ("noise", "smooth", "large"):
        img, flow, gout = make(B, C, H, W, 7, kind)
        ref_i, ref_f = torch.zeros_like(img), torch.zeros_like(flow)
        assert bwd(img, flow, gout, ref_i, ref_f, 0) == 0
        for sel, lab in ((0xB000, "accumulate"), (0xD000, "overwrite")):
            gi = torch.zeros_like(img) if sel == 0xB000 else torch.full_like(img, float("nan"))
            gf = torch.full_like(flow, float("nan"))
            assert bwd(img, flow, gout, gi, gf, sel) == 0
            torch.cuda.synchronize()
            di, df = float((gi - ref_i).abs().max()), float((gf - ref_f).abs().max())
            scale = float(ref_i.abs().max())
            ok = di <= 2e-5 * max(scale, 1.0) and df == 0.0
            worst = max(worst, di / max(scale, 1.0))
            print("   %-18s %-7s %-10s |d grad_img| %.2e (max %.1f)  |d grad_flow| %.2e  %s" % ((B, C, H, W), kind, lab, di, scale, df, "ok" if ok else "FAIL"))
print("worst relative grad_img difference %.2e" % worst)

B, C, H, W = 8, 3, 384, 512
for kind in ("noise", "smooth"):
    img, flow, gout = make(B, C, H, W, 0, kind)
    gimg, gflow = torch.zeros_like(img), torch.zeros_like(flow)
    print("%s flow, %d x %d x %d x %d" % (kind, B, C, H, W))
    for sel, lab, fill in ((0, "push (default)", True), (0xB000, "pull, accumulate", True), (0xD000, "pull, overwrite", False), (0xE000, "pull, overwrite, no far kernel", False),
                           (0xD000 | 0x200, "pull, no store / far kernel", False), (0xD000 | 0x400, "pull, no scatter", False),
                           (0xD000 | 0x800, "pull, no gather", False), (0xD000 | 0xE00, "pull, none of them", False)):
        t = timeit(lambda: bwd(img, flow, gout, gimg, gflow, sel))
        def run():
            gimg.zero_()
            bwd(img, flow, gout, gimg, gflow, sel)
        tz = timeit(run) if fill else t
        print("   %-30s %.1f us kernel(s)   %.1f us as the wrapper runs it" % (lab, t, tz))
