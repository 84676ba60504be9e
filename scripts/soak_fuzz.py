"""Soak test (GPU box): many seeded random shapes through the C ABI against the oracle -- the same checks as
tests/test_gpu_parity.py's fuzz tests, more cases, time-boxed.  Usage: python scripts/soak_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fn2_capi
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc, dev, rng = Oracle(), torch.device("cuda:0"), np.random.default_rng(seed)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
mx = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.asarray(a).size else 0.0
t0, ncorr, nimg, worst = time.time(), 0, 0, {}
def note(k, v): worst[k] = max(worst.get(k, 0.0), v)
while time.time() - t0 < budget:
    if rng.random() < 0.6:
        md = int(rng.choice([2, 4, 8, 10, 12, 16, 18, 20, 20, 20, 21]))
        C = int(rng.choice([1, 5, 16, 32, 64, 64, 128, 192]))
        H, W, B = int(rng.integers(2, 40)), int(rng.integers(2, 140)), int(rng.integers(1, 4))
        if rng.random() < 0.7: H += H & 1; W += (-W) % 4; C = max(16, C - C % 16)
        if rng.random() < 0.4:   # shapes the f16x2 matrix-core kernels take (FlowNetC's cost volume on maps up to 64 wide)
            md, C, H, W = 20, int(rng.choice([64, 128, 192, 256])), 2 * int(rng.integers(1, 31)), 8 * int(rng.integers(1, 9))
            if rng.random() < 0.4: W, B = 8 * int(rng.integers(9, 33)), int(rng.integers(1, 3))   # wider than 64 px: column windows
        # operand magnitudes: the block-scaled f16x2 kernels must be fp32-class at any of them (errors below are RELATIVE to the
        # largest reference value); now and then a heavy-tailed operand (log-normal magnitudes) or a few huge outliers
        sc = lambda: np.float32(rng.choice([1e-7, 1e-3, 1.0, 1.0, 30.0, 1e4]))
        a = rng.standard_normal((B, C, H, W)).astype(np.float32) * sc()
        b = rng.standard_normal((B, C, H, W)).astype(np.float32) * sc()
        D2 = (2 * (md // 2) + 1) ** 2
        go = rng.standard_normal((B, D2, H, W)).astype(np.float32) * sc()
        if rng.random() < 0.15: a *= np.exp(2 * rng.standard_normal(a.shape)).astype(np.float32)
        if rng.random() < 0.15: go.reshape(-1)[rng.integers(0, go.size, 3)] *= np.float32(1e6)
        ref = orc.corr_fwd(a, b, md, 1, md, 1, 2)
        out = torch.full((B, D2, H, W), float("nan"), device=dev)
        fn2_capi.correlation_forward(D(a), D(b), md, 1, md, 1, 2, out=out)
        s = max(1e-30, float(np.abs(ref).max()))
        e = mx(out.cpu().numpy(), ref) / s; note("corr_fwd", e); assert e <= 3e-6, ("fwd", B, C, H, W, md, e)
        r1, r2 = orc.corr_bwd(a, b, go, md, 1, md, 1, 2)
        g1 = torch.full((B, C, H, W), float("nan"), device=dev); g2 = torch.full_like(g1, float("nan"))
        fn2_capi.correlation_backward(D(a), D(b), D(go), md, 1, md, 1, 2, out=(g1, g2))
        e = max(mx(g1.cpu().numpy(), r1) / max(1e-30, float(np.abs(r1).max())), mx(g2.cpu().numpy(), r2) / max(1e-30, float(np.abs(r2).max())))
        note("corr_bwd", e); assert e <= 6e-6, ("bwd", B, C, H, W, md, e)
        ncorr += 1
        if md == 20 and C % 64 == 0 and H % 2 == 0 and W % 8 == 0 and rng.random() < 0.5:
            # half tensors on the single-product kernels (forward C % 128 == 0; backward W <= 64; otherwise the general kernel):
            # against the oracle on the half-rounded inputs, half an ulp of the result plus fp32 summation noise
            ah, bh, gh = (torch.from_numpy(x / np.float32(max(1e-30, np.abs(x).max()))).half() for x in (a, b, go))   # unit range
            rf = orc.corr_fwd(ah.float().numpy(), bh.float().numpy(), 20, 1, 20, 1, 2)
            q1, q2 = orc.corr_bwd(ah.float().numpy(), bh.float().numpy(), gh.float().numpy(), 20, 1, 20, 1, 2)
            oh = torch.full((B, 441, H, W), float("nan"), dtype=torch.float16, device=dev)
            fn2_capi.correlation_forward(ah.to(dev), bh.to(dev), 20, 1, 20, 1, 2, out=oh)
            h1 = torch.full((B, C, H, W), float("nan"), dtype=torch.float16, device=dev); h2 = torch.full_like(h1, float("nan"))
            fn2_capi.correlation_backward(ah.to(dev), bh.to(dev), gh.to(dev), 20, 1, 20, 1, 2, out=(h1, h2))
            for name, got, ref in (("half_fwd", oh, rf), ("half_bwd", h1, q1), ("half_bwd", h2, q2)):
                gnp = got.float().cpu().numpy()
                tol = 2.0 ** -11 * np.abs(ref) + 6.1e-5 + 4e-6 * np.abs(ref).max()     # half rounding (normal / subnormal) + summation order
                bad = np.abs(gnp - ref) > tol
                note(name, float(np.max(np.abs(gnp - ref) / tol))); assert not bad.any(), (name, B, C, H, W, float(np.abs(gnp - ref).max()))
    else:
        B, C, H, W, bil = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 3, 3, 4])), int(rng.integers(1, 130)), int(rng.integers(1, 200)), bool(rng.integers(0, 2))
        if rng.random() < 0.5: W += (-W) % 4   # tileable widths half of the time
        ks = int(rng.choice([1, 1, 1, 2, 3]))   # window sums (kernel_size > 1) now and then
        img = rng.standard_normal((B, C, H, W)).astype(np.float32)
        flow = (rng.standard_normal((B, 2, H, W)) * float(rng.choice([0.5, 3.0, 12.0]))).astype(np.float32)
        flow.reshape(-1)[rng.integers(0, flow.size, max(1, flow.size // 40))] *= 40.0
        shifted = rng.random() < 0.35
        if shifted:   # a translation under it (the backward windows follow the flow; border pile-ups)
            flow[:, 0] += np.float32(rng.uniform(-70, 70)); flow[:, 1] += np.float32(rng.uniform(-70, 70))
        gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
        lib, P, st = fn2_capi.lib(), (lambda t: __import__("ctypes").c_void_p(t.data_ptr())), None
        import ctypes
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        imd, fld, god = D(img), D(flow), D(gout)
        out = torch.full((B, C, H, W), float("nan"), device=dev)
        assert lib.fn2_resample2d_forward(P(imd), None, P(fld), P(out), B, C, H, W, H, W, ks, int(bil), st) == 0
        rf = orc.resample_fwd(img, flow, ks, bil)
        e = mx(out.cpu().numpy(), rf) / max(1.0, float(np.abs(rf).max())); note("resample_fwd", e); assert e <= 1e-5 * ks * ks + 1e-4 * (ks == 1), ("rfwd", B, C, H, W, bil, ks, e)
        gi, gf = torch.zeros(B, C, H, W, device=dev), torch.full((B, 2, H, W), float("nan"), device=dev)
        assert lib.fn2_resample2d_backward(P(imd), None, P(fld), P(god), P(gi), P(gf), B, C, H, W, H, W, ks, int(bil), st) == 0
        rgi, rgf = orc.resample_bwd(img, flow, gout, ks, bil)
        s = max(1.0, float(np.abs(rgi).max()), float(np.abs(rgf).max()))
        e = max(mx(gi.cpu().numpy(), rgi), mx(gf.cpu().numpy(), rgf)) / s; note("resample_bwd", e)
        # a translated field piles thousands of terms onto border cells: the oracle's own sequential fp32 sum is that far from fp64
        assert e <= (1e-4 if shifted else 2e-5), ("rbwd", B, C, H, W, bil, shifted, e)
        pair = rng.standard_normal((B, 2 * C, H, W)).astype(np.float32)
        got = fn2_capi.warp_diff_norm_cat(D(pair), fld, 20.0, bil).cpu().numpy()
        warped = orc.resample_fwd(np.ascontiguousarray(pair[:, C:]), flow, 1, bil)
        refc = np.concatenate((pair, warped, flow * (np.float32(1.0) / np.float32(20.0)), orc.chnorm_fwd(pair[:, :C] - warped)), axis=1)
        e = mx(got, refc); note("warp_diff_norm_cat", e); assert e <= 1e-4, ("n2", B, C, H, W, bil, e)
        # N2's backward (round 5): the one fused kernel against the oracle's composition of channelnorm_kernel.cu:63-96 and
        # resample2d_kernel.cu:75-198 on the forward's own output (C = 3 on tileable maps: the LDS-window kernel; else one lane per pixel)
        gcat = rng.standard_normal((B, 3 * C + 3, H, W)).astype(np.float32)
        want_pair = bool(rng.integers(0, 2))
        gp, gf = fn2_capi.warp_diff_norm_cat_backward(D(pair), fld, D(got), D(gcat), 20.0, bil, want_pair)
        diff = pair[:, :C] - got[:, 2 * C:3 * C]
        gdiff = orc.chnorm_bwd(diff, np.ascontiguousarray(got[:, 3 * C + 2:]), np.ascontiguousarray(gcat[:, 3 * C + 2:]))
        rgi, rgf = orc.resample_bwd(np.ascontiguousarray(pair[:, C:]), flow, np.ascontiguousarray(gcat[:, 2 * C:3 * C] - gdiff), 1, True)
        rgf = rgf + gcat[:, 3 * C:3 * C + 2] * (np.float32(1.0) / np.float32(20.0))
        s = max(1.0, float(np.abs(rgi).max()), float(np.abs(rgf).max()), float(np.abs(gdiff).max()))
        e = mx(gf.cpu().numpy(), rgf) / s
        if want_pair:
            e = max(e, mx(gp[:, :C].cpu().numpy(), gcat[:, :C] + gdiff) / s, mx(gp[:, C:].cpu().numpy(), gcat[:, C:2 * C] + rgi) / s)
        note("warp_diff_norm_cat_bwd", e); assert e <= (1e-4 if shifted else 2e-5), ("n2bwd", B, C, H, W, bil, want_pair, shifted, e)
        nimg += 1
print("soak ok: %d correlation cases, %d image cases in %.0f s; worst normalised errors %s" % (ncorr, nimg, time.time() - t0, {k: float("%.3g" % v) for k, v in worst.items()}))
