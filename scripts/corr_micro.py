"""Micro-benchmark of correlation-forward instantiations through the C ABI (fn2_correlation_forward_ex).
algo 2 = shipped MFMA kernel; 100 + 8*cfg + var = profiling instantiations (correlation_mfma.hip).
Usage: python scripts/corr_micro.py [--algos 2,100,101,...] [--iters 30] [--check]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch  # noqa: E402
import fn2_capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--algos", default="2")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--check", action="store_true")
ap.add_argument("--shape", default="8,256,48,64")
ap.add_argument("--md", type=int, default=20)
ap.add_argument("--bwd", default="")
ap.add_argument("--batch", type=int, default=40)
ap.add_argument("--in-scale", type=float, default=1.0, help="multiply in1 / in2 (the block-scaled kernels skip a scale of 1)")
ap.add_argument("--go-scale", type=float, default=1.0, help="multiply gradOutput")
ap.add_argument("--lib", default="", help="A/B runs: load this build of libflownet2_hip.so instead of the in-tree one")
a = ap.parse_args()
if a.lib:   # an A/B build (scripts/build_ablations.sh): used for the public AND the debug entry points if it exports them
    fn2_capi.LIB_PATH = os.path.abspath(a.lib)
    fn2_capi.DEBUG_LIB_PATH = os.path.abspath(a.lib)
B, C, H, W = (int(v) for v in a.shape.split(","))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
in1 = (a.in_scale * torch.randn(B, C, H, W, generator=g)).to(dev)
in2 = (a.in_scale * torch.randn(B, C, H, W, generator=g)).to(dev)
D = 2 * (a.md // 2) + 1
out = torch.empty(B, D * D, H, W, device=dev)
ref = None
res = {}
dbg = torch.zeros(256 * 2 * 16, dtype=torch.int64, device=dev)
if any(int(v) >= 100 for v in (a.algos + "," + a.bwd).split(",") if v):
    fn2_capi.debug_lib().fn2_debug_set_buffer(fn2_capi._p(dbg))
for algo in (int(v) for v in a.algos.split(",")):
    try:
        for _ in range(3):
            fn2_capi.correlation_forward(in1, in2, a.md, 1, a.md, 1, 2, algo=algo, out=out)
    except RuntimeError as e:
        print(algo, "ERR", e)
        continue
    torch.cuda.synchronize()
    evs = []
    for _ in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn2_capi.correlation_forward(in1, in2, a.md, 1, a.md, 1, 2, algo=algo, out=out)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    r = {"median_us": round(ts[len(ts) // 2], 2), "min_us": round(ts[0], 2)}
    # back-to-back launches between one event pair: the per-launch time without the event pair's own ~6 us
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.batch):
        fn2_capi.correlation_forward(in1, in2, a.md, 1, a.md, 1, 2, algo=algo, out=out)
    e.record()
    torch.cuda.synchronize()
    r["b2b_us"] = round(s.elapsed_time(e) * 1e3 / a.batch, 2)
    if a.check and (algo in (2, 3, 4, 5000, 7000) or (100 <= algo < 1000 and (algo - 100) % 256 == 0) or (1000 <= algo < 5000 and algo % 10 == 0)):
        if ref is None:
            ref = fn2_capi.correlation_forward(in1, in2, a.md, 1, a.md, 1, 2, algo=1)
        r["max_abs_vs_direct"] = float((out - ref).abs().max())
    res[algo] = r
    print(algo, r, flush=True)
    if algo == 5064:   # f16x2 timeline: stamps of wave 0 (matrix) and wave 8 (staging) of every workgroup
        torch.cuda.synchronize()
        st = dbg.cpu().view(256, 2, 16)
        base = st[:, :, 0].min(dim=1, keepdim=True).values.unsqueeze(2)   # per workgroup (the XCDs have their own counters)
        zero = st == 0
        st = (st - base).double()
        st[zero] = -1
        t0 = 0
        names = ["start", "loads issued", "t0 data ready", "t0 K done", "t0 image", "t0 stores issued", "t0 free",
                 "-", "t1 data ready", "t1 K done", "t1 image", "t1 stores issued", "t1 free", "-", "-", "end"]
        for role, rn in ((0, "matrix wave 0"), (1, "staging wave 8")):
            print("  ", rn)
            for i, nm in enumerate(names):
                v = st[:, role, i]
                v = v[v >= 0]
                if len(v):
                    print("     %-18s mean %8.0f  min %8.0f  max %8.0f   (n=%d)" % (nm, float((v - t0).mean()), float((v - t0).min()), float((v - t0).max()), len(v)))
    if algo == 3124:   # instrumented forward: s_memtime stamps of two workgroups (dispatch rounds 0 and 1 of one CU)
        torch.cuda.synchronize()
        st = out.view(-1)[:256].view(torch.int64).cpu().view(2, 8, 8)
        t0 = int(st[st > 0].min())
        for sl in range(2):
            for w in range(8):
                print("   wg", sl, "wave", w, [int(v) - t0 if v > 0 else None for v in st[sl, w]])
if a.bwd:
    gout = (a.go_scale * torch.randn(B, D * D, H, W, generator=g)).to(dev)
    refb = None
    for algo in (int(v) for v in a.bwd.split(",")):
        try:
            for _ in range(2):
                g1, g2 = fn2_capi.correlation_backward(in1, in2, gout, a.md, 1, a.md, 1, 2, algo=algo)
        except RuntimeError as e:
            print("bwd", algo, "ERR", e)
            continue
        torch.cuda.synchronize()
        evs = []
        for _ in range(a.iters):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn2_capi.correlation_backward(in1, in2, gout, a.md, 1, a.md, 1, 2, algo=algo)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        r = {"median_us": round(ts[len(ts) // 2], 2), "min_us": round(ts[0], 2)}
        if 100 <= algo < 6000 and (algo - 140) >= 0 and ((algo - 140) & 128):
            # instrumented instantiation: s_memtime stamps of one phase of each wave of one workgroup (100 MHz ticks)
            torch.cuda.synchronize()
            st = g1.view(-1)[:128].view(torch.int64).cpu().view(8, 8)
            t0 = int(st[:, :7][st[:, :7] > 0].min())
            for w in range(8):
                print("   wave", w, [int(v) - t0 if v > 0 else None for v in st[w, :7]])
        if 6000 <= algo < 8000 and ((algo - 6000) & 64):   # f16x2 backward timeline of each workgroup's first task
            torch.cuda.synchronize()
            st = dbg.cpu().view(256, 2, 16)
            base = st[:, :, 0].min(dim=1, keepdim=True).values.unsqueeze(2)
            zero = st == 0
            st = (st - base).double()
            st[zero] = -1
            names = ["start", "loads issued", "G(0) written", "barrier A", "u0 ph1 done", "u0 barrier B", "u0 ph2 done", "u0 barrier A'",
                     "u1 ph1 done", "u1 barrier B", "u1 ph2 done", "u1 barrier A'", "u loop done", "image complete", "task done", "end"]
            for role, rn in ((0, "staging wave 0"), (1, "matrix wave 8")):
                print("  ", rn)
                for i, nm in enumerate(names):
                    v = st[:, role, i]
                    v = v[v >= 0]
                    if len(v):
                        print("     %-18s mean %8.0f  min %8.0f  max %8.0f   (n=%d)" % (nm, float(v.mean()), float(v.min()), float(v.max()), len(v)))
        if a.check:
            if refb is None:
                refb = fn2_capi.correlation_backward(in1, in2, gout, a.md, 1, a.md, 1, 2, algo=1)
            r["max_abs_vs_direct"] = [float((g1 - refb[0]).abs().max()), float((g2 - refb[1]).abs().max())]
        res["bwd%d" % algo] = r
        print("bwd", algo, r, flush=True)
print(json.dumps(res))
