"""Round 5: Resample2d backward (resample_bwd_c3x) with its ablations, and row N2's fused backward against autograd through the
unfused layers.  (The A/B against the round-4 kernel, the in-kernel zero fill and the far-pixel list: profiles/r05_b_*.log.)  8 x 3 x 384 x 512, the SURVEY's white-noise flow, a smooth flow and a translated one."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
dbg, lib = fn2_capi.debug_lib(), fn2_capi.lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 3, 384, 512
img = (torch.rand(B, C, H, W, generator=g) - 0.5).to(dev)
flow = torch.randn(B, 2, H, W, generator=g) * 4.0
idx = torch.randint(0, flow.numel(), (flow.numel() // 100,), generator=g)
flow.view(-1)[idx] *= 20.0
smooth = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 30, 31, 1, 15).to(dev)
gout = torch.randn(B, C, H, W, generator=g).to(dev)
gimg = torch.zeros_like(img); gflow = torch.zeros(B, 2, H, W, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
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
shift = smooth.clone(); shift[:, 0] += 25.0; shift[:, 1] -= 18.0
for name, fl in (("white-noise flow N(0,4)+1% outliers", flow.to(dev)), ("smooth flow (sigma ~ 5 px)", smooth), ("the smooth flow + a translation of (25, -18) px", shift)):
    print(name)
    ref = None
    for flags, lab in ((0, "round-5 kernel (c3x)"), (0x200, "round-5, no flush"), (0x400, "round-5, no scatter")):
        call = lambda: dbg.fn2_debug_resample2d_backward(P(img), None, P(fl), P(gout), P(gimg), P(gflow), B, C, H, W, H, W, 1, 1, flags, st)
        def run():
            gimg.zero_(); call()
        t, tz = timeit(call), timeit(run)
        run(); torch.cuda.synchronize()
        if ref is None: ref = (gimg.clone(), gflow.clone())
        print("   %-36s %.1f us  (%.1f with torch's zero fill)   max |d grad_img| %.2e  |d grad_flow| %.2e vs the first row" % (
            lab, t, tz, float((gimg - ref[0]).abs().max()), float((gflow - ref[1]).abs().max())))
    # row N2: fused backward against autograd through the unfused layers
    from networks.resample2d_package.resample2d import Resample2d, WarpDiffNormCat
    from networks.channelnorm_package.channelnorm import ChannelNorm
    x = torch.randn(B, 6, H, W, generator=g).to(dev)
    gcat = torch.randn(B, 12, H, W, generator=g).to(dev)
    for need_x in (False, True):
        xl, f = x.clone().requires_grad_(need_x), fl.clone().requires_grad_(True)
        res = Resample2d()(xl[:, 3:], f)
        unf = torch.cat((xl, res, f / 20.0, ChannelNorm()(xl[:, :3] - res)), dim=1)
        tu = timeit(lambda: unf.backward(gcat, retain_graph=True))
        xl2, f2 = x.clone().requires_grad_(need_x), fl.clone().requires_grad_(True)
        fus = WarpDiffNormCat(20.0)(xl2, f2)
        tf = timeit(lambda: fus.backward(gcat, retain_graph=True))
        print("   N2 backward, pair %s a gradient: autograd through the unfused layers %.1f us, fused %.1f us" % ("needs" if need_x else "without", tu, tf))
    out = fus.detach()
    gp, gf = torch.empty_like(x), torch.empty_like(fl)
    k1 = lambda: lib.fn2_warp_diff_norm_cat_backward(P(x), P(fl), P(out), P(gcat), P(gp), P(gf), ctypes.c_float(20.0), B, 3, H, W, 1, st)
    k0 = lambda: lib.fn2_warp_diff_norm_cat_backward(P(x), P(fl), P(out), P(gcat), None, P(gf), ctypes.c_float(20.0), B, 3, H, W, 1, st)
    print("   N2 backward kernel alone (C ABI): with the pair's gradient %.1f us, flow gradient only %.1f us" % (timeit(k1), timeit(k0)))
