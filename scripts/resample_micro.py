"""Micro-benchmark of the resample2d kernels through the C ABI (fn2_debug_resample2d_*: flag bits 8.. select profiling variants)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
lib = fn2_capi.debug_lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 3, 384, 512
img = (torch.rand(B, C, H, W, generator=g) - 0.5).to(dev)
flow = torch.randn(B, 2, H, W, generator=g) * 4.0
idx = torch.randint(0, flow.numel(), (flow.numel() // 100,), generator=g)
flow.view(-1)[idx] *= 20.0
smooth = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, generator=g) * 30, 31, 1, 15).to(dev)
gout = torch.randn(B, C, H, W, generator=g).to(dev)
out = torch.zeros_like(img); gimg = torch.zeros_like(img); gflow = torch.zeros(B, 2, H, W, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    for _ in range(3): fn()
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
    for flags, lab in ((1, "fwd (default: 3 windows resident, 32x64)"), (1 | 0x1000, "fwd per channel 48x64"), (1 | 0x2000, "fwd per channel 32x64"), (1 | 0x3000, "fwd 96x64, 3 windows resident"), (1 | 0x4000, "fwd 48x64, 4 px per thread"), (1 | 0x8000, "fwd 64x64, 4 px per thread"), (1 | 0x100, "fwd untiled")):
        t = timeit(lambda: lib.fn2_debug_resample2d_forward(P(img), None, P(fl), P(out), B, C, H, W, H, W, 1, 1, flags & ~0xff, st))
        torch.cuda.synchronize()
        if flags == 1:
            ref_out = out.clone()
        print("   %-28s %.1f us   max |d| vs the first row %.2e" % (lab, t, float((out - ref_out).abs().max())))
    ref = None
    for flags, lab in ((1, "bwd (default: 3 channels at once)"), (1 | 0xA000, "bwd tiled, round-3 choice"), (1 | 0x1000, "bwd tiled 48x64"), (1 | 0x2000, "bwd tiled 32x64"), (1 | 0x3000, "bwd tiled 64x64"),
                       (1 | 0x5000, "bwd 48x64 +-12 f32 CAS"), (1 | 0x4000, "bwd 48x64 +-12 fp64 cells"), (1 | 0x8000, "bwd 32x64 +-16 fp64 cells"),
                       (1 | 0x200, "3ch, no flush"), (1 | 0x400, "3ch, no scatter"), (1 | 0xE00, "3ch, none of them"), (1 | 0xC000, "bwd 48x64 +-16 fp64, 1 WG/CU"), (1 | 0x9000, "bwd 48x64 +-16 f32, 1 WG/CU"),
                       (1 | 0x6000, "bwd 96x64 +-16 f32, 1 WG/CU"), (1 | 0x7000, "bwd 96x64 +-16 fp64, 1 WG/CU"),
                       (1 | 0x4000 | 0x200, "fp64 48x64+-12, no flush"), (1 | 0x4000 | 0x400, "fp64 48x64+-12, no scatter"),
                       (1 | 0x8200, "fp64 32x64, no flush"), (1 | 0x8400, "fp64 32x64, no scatter"), (1 | 0x8800, "fp64 32x64, no img gather"),
                       (1 | 0x8E00, "fp64 32x64, none of them"), (1 | 0x100, "bwd untiled")):
        def run():
            gimg.zero_()
            lib.fn2_debug_resample2d_backward(P(img), None, P(fl), P(gout), P(gimg), P(gflow), B, C, H, W, H, W, 1, 1, flags & ~0xff, st)
        t = timeit(lambda: lib.fn2_debug_resample2d_backward(P(img), None, P(fl), P(gout), P(gimg), P(gflow), B, C, H, W, H, W, 1, 1, flags & ~0xff, st))
        tz = timeit(run)
        run(); torch.cuda.synchronize()
        if ref is None:
            ref = (gimg.clone(), gflow.clone())
        d = (float((gimg - ref[0]).abs().max()), float((gflow - ref[1]).abs().max()))
        print("   %-28s %.1f us  (%.1f with the zero fill)   max |d grad_img| %.2e  |d grad_flow| %.2e vs the first row" % (lab, t, tz, d[0], d[1]))
