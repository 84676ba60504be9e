#!/usr/bin/env python
"""bench.py -- throughput of the FlowNet2 custom-layer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: starts its own N ranks (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path, forward + backward, over one batch of 8 synthetic image
pairs at 384x512 (BASELINE.json):
    Correlation  fwd+bwd  in 8x256x48x64 fp32, FlowNetC parameters (20,1,20,1,2)  (configs[1])
    Resample2d   fwd+bwd  img 8x3x384x512, flow 8x2x384x512
    ChannelNorm  fwd+bwd  8x3x384x512
all through the reference-named pybind modules (correlation_cuda / resample2d_cuda /
channelnorm_cuda -> C ABI -> gfx950 HIP kernels).  Inputs are resident in HBM before the timed
region.  N ranks = N independent batches (the path shards over the batch with no data-path
collective: weak scaling); value = image pairs processed by all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` is for the correlation forward kernel (the kernel
BASELINE.json's metric names): achieved = algorithmic bytes of one launch (SURVEY.md 8d:
2*B*C*H*W*4 read + B*441*H*W*4 written = 93 683 712 B) / mean launch duration, measured with HIP
events on the launch stream inside the timed steps (an event pair around that kernel in every fourth timed step, on
events created before the region; the other five ops are timed in a second pass over the same K steps: event pairs
around everything cost 10-25 % of a 0.19 ms step, host-dependent).  `cpu_baseline` times the CPU oracle
(oracle/, a restatement of the reference kernels; the reference itself has no CPU path) on a
bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "flownet2-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
CORR = dict(B=8, C=256, H=48, W=64, pad=20, k=1, md=20, s1=1, s2=2)
IMG = dict(B=8, C=3, H=384, W=512)


def corr_fwd_bytes(B=CORR["B"]):
    return 2 * B * CORR["C"] * CORR["H"] * CORR["W"] * 4 + B * 441 * CORR["H"] * CORR["W"] * 4


def algorithmic_bytes(B=8):
    """SURVEY.md 8(d) per-call figures (fp32)."""
    hw_c = B * CORR["C"] * CORR["H"] * CORR["W"] * 4
    out_c = B * 441 * CORR["H"] * CORR["W"] * 4
    px = B * IMG["H"] * IMG["W"] * 4
    return {
        "corr_fwd": 2 * hw_c + out_c,
        "corr_bwd": out_c + 2 * hw_c + 2 * hw_c,
        "resample_fwd": (3 + 2 + 3) * px,
        "resample_bwd": (3 + 2 + 3) * px + (3 + 2) * px,
        "chnorm_fwd": (3 + 1) * px,
        "chnorm_bwd": (3 + 1 + 1) * px + 3 * px,
    }


class HotPath:
    """Device-resident synthetic inputs + one fwd/bwd pass through the three extension modules."""

    def __init__(self, dev, seed):
        import channelnorm_cuda
        import correlation_cuda
        import resample2d_cuda
        self.m_corr, self.m_res, self.m_cn = correlation_cuda, resample2d_cuda, channelnorm_cuda
        g = torch.Generator().manual_seed(seed)
        c, i = CORR, IMG
        self.in1 = torch.randn(c["B"], c["C"], c["H"], c["W"], generator=g).to(dev)
        self.in2 = torch.randn(c["B"], c["C"], c["H"], c["W"], generator=g).to(dev)
        self.gcorr = torch.randn(c["B"], 441, c["H"], c["W"], generator=g).to(dev)
        self.img = (torch.rand(i["B"], 3, i["H"], i["W"], generator=g) - 0.5).to(dev)
        flow = torch.randn(i["B"], 2, i["H"], i["W"], generator=g) * 4.0
        idx = torch.randint(0, flow.numel(), (flow.numel() // 100,), generator=g)
        flow.view(-1)[idx] *= 20.0
        self.flow = flow.to(dev)
        self.gwarp = torch.randn(i["B"], 3, i["H"], i["W"], generator=g).to(dev)
        self.gnorm = torch.randn(i["B"], 1, i["H"], i["W"], generator=g).to(dev)
        self.cparams = (c["pad"], c["k"], c["md"], c["s1"], c["s2"], 1)
        e = self.in1.new_empty
        self.scr1, self.scr2 = e(0), e(0)
        self.out, self.g1, self.g2 = e(0), e(0), e(0)
        self.warped = torch.zeros_like(self.img)
        self.gimg = torch.zeros_like(self.img)
        self.gflow = torch.zeros_like(self.flow)
        self.norm = torch.zeros(i["B"], 1, i["H"], i["W"], device=dev)
        self.gdiff = torch.zeros_like(self.img)
        self.ev = None

    def corr_fwd(self):
        self.m_corr.forward(self.in1, self.in2, self.scr1, self.scr2, self.out, *self.cparams)

    def corr_bwd(self):
        self.m_corr.backward(self.in1, self.in2, self.scr1, self.scr2, self.gcorr, self.g1, self.g2, *self.cparams)

    def step_plain(self):
        """The step with nothing around it: seven launches (six kernels + the zero fill) through the pybind modules."""
        self.m_corr.forward(self.in1, self.in2, self.scr1, self.scr2, self.out, *self.cparams)
        self.step_rest()

    def step_rest(self):
        """Everything after the correlation forward."""
        self.m_corr.backward(self.in1, self.in2, self.scr1, self.scr2, self.gcorr, self.g1, self.g2, *self.cparams)
        self.m_res.forward(self.img, self.flow, self.warped, 1, True)
        self.m_cn.forward(self.warped, self.norm, 2)
        self.m_cn.backward(self.warped, self.norm, self.gnorm, self.gdiff, 2)
        self.gimg.zero_()   # the reference wrapper zero-fills grad_input1 every call (resample2d.py:31)
        self.m_res.backward(self.img, self.flow, self.gwarp, self.gimg, self.gflow, 1, True)

    def step(self, events=None, only=None):
        """fwd + bwd of the three layers.  `events`, if given, collects (start, stop) HIP event
        pairs around each op (or just the ops named in `only`) on the current stream."""
        if events is None and only is None:
            return self.step_plain()
        def timed(name, fn):
            if events is None or (only is not None and name not in only):
                fn()
                return
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            events.setdefault(name, []).append((s, e))
        if only != "skip_corr_fwd":
            timed("corr_fwd", self.corr_fwd)
        timed("corr_bwd", self.corr_bwd)
        timed("resample_fwd", lambda: self.m_res.forward(self.img, self.flow, self.warped, 1, True))
        timed("chnorm_fwd", lambda: self.m_cn.forward(self.warped, self.norm, 2))
        timed("chnorm_bwd", lambda: self.m_cn.backward(self.warped, self.norm, self.gnorm, self.gdiff, 2))

        def res_bwd():
            self.gimg.zero_()   # the reference wrapper zero-fills grad_input1 every call (resample2d.py:31)
            self.m_res.backward(self.img, self.flow, self.gwarp, self.gimg, self.gflow, 1, True)
        timed("resample_bwd", res_bwd)


    def module_steps(self, steps):
        """`steps` fwd+bwd passes through the nn.Module wrappers and autograd (wall clock, synchronised)."""
        from networks.channelnorm_package.channelnorm import ChannelNorm
        from networks.correlation_package.correlation import Correlation
        from networks.resample2d_package.resample2d import Resample2d
        corr, warp, norm = Correlation(*self.cparams), Resample2d(), ChannelNorm()
        a, b = self.in1.clone().requires_grad_(True), self.in2.clone().requires_grad_(True)
        img1, flow = self.img.clone().requires_grad_(True), self.flow.clone().requires_grad_(True)
        img0 = self.gwarp

        def one():
            a.grad = b.grad = img1.grad = flow.grad = None
            corr(a, b).backward(self.gcorr)
            norm(img0 - warp(img1, flow)).backward(self.gnorm)
        # the same statements through the fused row FlowNet2 actually runs at its warp sites (models.py:157-161: the image pair is
        # the network's input and gets no gradient): WarpDiffNorm, one kernel forward, one gather-only kernel backward
        from networks.resample2d_package.resample2d import WarpDiffNorm
        werr = WarpDiffNorm()
        pair = torch.cat((img0, self.img), 1)
        flow2 = self.flow.clone().requires_grad_(True)

        def one_fused():
            a.grad = b.grad = flow2.grad = None
            corr(a, b).backward(self.gcorr)
            werr(pair, flow2).backward(self.gnorm)

        # exactly the kernels of step() -- no difference op between warp and norm -- through the modules and autograd: what the
        # wrappers themselves add (allocations, the autograd nodes, the engine) to the same GPU work
        def one_same_ops():
            a.grad = b.grad = img1.grad = flow.grad = None
            corr(a, b).backward(self.gcorr)
            norm(warp(img1, flow)).backward(self.gnorm)

        def clock(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        # the host's floor under ANY module-level step of this shape: two trivial graphs (a multiply + a sum on 64 floats), each with its
        # own .backward() -- Python call, node construction, the engine's hand-over to its device thread, AccumulateGrad.  A step through
        # the modules cannot cost less than this on the host it runs on, whatever the kernels take
        tiny = torch.randn(64, device=self.in1.device, requires_grad=True)

        def floor():
            tiny.grad = None
            (tiny * 2.0).sum().backward()
            tiny.grad = None
            (tiny * 3.0).sum().backward()
        return clock(one), clock(one_fused), clock(one_same_ops), clock(floor)


def cpu_baseline(max_seconds=30.0):
    """Oracle (a C restatement of the reference kernels, OpenMP over the host cores) on a bounded
    sample: whole steps of the same workload (batch 8), repeated while the time budget allows."""
    import numpy as np
    from oracle.oracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(0)
    c, i = CORR, IMG
    nb = c["B"]
    a = rng.standard_normal((nb, c["C"], c["H"], c["W"])).astype(np.float32)
    b = rng.standard_normal((nb, c["C"], c["H"], c["W"])).astype(np.float32)
    go = rng.standard_normal((nb, 441, c["H"], c["W"])).astype(np.float32)
    img = (rng.random((nb, 3, i["H"], i["W"])) - 0.5).astype(np.float32)
    flow = (rng.standard_normal((nb, 2, i["H"], i["W"])) * 4).astype(np.float32)
    gw = rng.standard_normal((nb, 3, i["H"], i["W"])).astype(np.float32)
    gn = rng.standard_normal((nb, 1, i["H"], i["W"])).astype(np.float32)
    p = (c["pad"], c["k"], c["md"], c["s1"], c["s2"])

    def one_step():
        orc.corr_fwd(a, b, *p)
        orc.corr_bwd(a, b, go, *p)
        w = orc.resample_fwd(img, flow)
        n = orc.chnorm_fwd(w)
        orc.chnorm_bwd(w, n, gn)
        orc.resample_bwd(img, flow, gw)

    t0 = time.perf_counter()
    one_step()                      # warm-up (page faults, OpenMP pool)
    warm = time.perf_counter() - t0
    times = []
    budget = max(0.0, max_seconds - warm)
    while True:
        t0 = time.perf_counter()
        one_step()
        times.append(time.perf_counter() - t0)
        if len(times) >= 7 or sum(times) + times[-1] > budget:
            break
    med = sorted(times)[len(times) // 2]
    return {
        "value": round(nb / med, 3),
        "unit": "image-pairs/s",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": f"{len(times)} whole steps of the same workload (batch 8: corr 8x256x48x64 fwd+bwd, resample2d + "
                  f"channelnorm 8x3x384x512 fwd+bwd) after 1 warm-up step, median; fp32; OpenMP default threads = "
                  f"all {os.cpu_count()} host cores (the reference has no CPU path: this is the restated oracle)",
        "seconds_per_step": round(med, 4),
    }


# ---- a second CPU formulation of the same step (SURVEY.md 8d "no sandbagging"): whole-array PyTorch at torch's fastest thread
# count on the host (16 of 256 cores; NOT faster than the OpenMP oracle on all cores: 12.9-15.4 against 13.7-15.9 pairs/s in round 5) -- 441 shifted channel contractions with EXPLICIT backward passes for the correlation, grid_sample (border,
# align_corners: the closed form of resample2d_kernel.cu:15-72) and its autograd for the warp, the L2 norm
# and its closed-form gradient.  Tolerance-level equal to the oracle (tests/test_bench_cpu_fast.py), not bit-exact: the
# summation order is whatever the vectorised kernels choose.
def torch_corr_fwd(a, b, md=20, s2=2):
    """Per displacement row tj: one batched GEMM (W x C) @ (C x Wp) per (item, image row) against the shifted, padded row of in2,
    then its 2r + 1 diagonals x' = x + s2 * ti (5x the useful multiply-adds, but BLAS-shaped: 7x faster than 441 shifted
    elementwise contractions on the same cores)."""
    B, C, H, W = a.shape
    r = md // s2
    D = 2 * r + 1
    bp = torch.nn.functional.pad(b, (md, md, md, md))
    out = a.new_empty(B, D * D, H, W)
    At = a.permute(0, 2, 3, 1).reshape(B * H, W, C)
    idx = (torch.arange(W).view(W, 1) + s2 * torch.arange(D).view(1, D)).unsqueeze(0).expand(B * H, W, D)   # column of the padded row
    for tj in range(D):
        Brow = bp[:, :, s2 * tj:s2 * tj + H].permute(0, 2, 1, 3).reshape(B * H, C, W + 2 * md)
        d = torch.bmm(At, Brow).gather(2, idx)                                   # (item, row) x W x D
        out[:, tj * D:(tj + 1) * D] = d.view(B, H, W, D).permute(0, 3, 1, 2)
    return out / C


def torch_corr_bwd(a, b, go, md=20, s2=2):
    """Explicit backward in the same shape: per displacement row the banded matrix G[x, x'] = gO[tj, ti] at x' = x + s2 * ti,
    gradInput1 rows += (in2 row) @ G^T, padded gradInput2 rows += (in1 row) @ G."""
    B, C, H, W = a.shape
    r = md // s2
    D = 2 * r + 1
    Wp = W + 2 * md
    bp = torch.nn.functional.pad(b, (md, md, md, md))
    A = a.permute(0, 2, 1, 3).reshape(B * H, C, W)
    idx = (torch.arange(W).view(W, 1) + s2 * torch.arange(D).view(1, D)).unsqueeze(0).expand(B * H, W, D)
    g1 = a.new_zeros(B * H, C, W)
    g2p = torch.zeros_like(bp)
    for tj in range(D):
        G = a.new_zeros(B * H, W, Wp)
        G.scatter_(2, idx, go[:, tj * D:(tj + 1) * D].permute(0, 2, 3, 1).reshape(B * H, W, D))
        Brow = bp[:, :, s2 * tj:s2 * tj + H].permute(0, 2, 1, 3).reshape(B * H, C, Wp)
        g1 += torch.bmm(Brow, G.transpose(1, 2))
        g2p[:, :, s2 * tj:s2 * tj + H] += torch.bmm(A, G).view(B, H, C, Wp).permute(0, 2, 1, 3)
    return g1.view(B, H, C, W).permute(0, 2, 1, 3) / C, g2p[:, :, md:md + H, md:md + W] / C


def torch_resample_grid(flow):
    B, _, H, W = flow.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=flow.dtype), torch.arange(W, dtype=flow.dtype), indexing="ij")
    gx = (xs + flow[:, 0]) * (2.0 / max(W - 1, 1)) - 1.0
    gy = (ys + flow[:, 1]) * (2.0 / max(H - 1, 1)) - 1.0
    return torch.stack((gx, gy), dim=-1)


def torch_resample_fwd(img, flow):
    return torch.nn.functional.grid_sample(img, torch_resample_grid(flow), mode="bilinear", padding_mode="border", align_corners=True)


def torch_resample_bwd(img, flow, gw):
    img = img.detach().requires_grad_(True)
    flow = flow.detach().requires_grad_(True)
    return torch.autograd.grad(torch_resample_fwd(img, flow), (img, flow), gw)


def torch_chnorm_fwd(x):
    return x.square().sum(1, keepdim=True).sqrt()


def torch_chnorm_bwd(x, n, gn):
    return gn * x / (n + 1e-9)


def cpu_baseline_fast(max_seconds=15.0):
    """The same whole steps (batch 8) in vectorised PyTorch on every host core, median of what fits the time budget."""
    nthreads_before = torch.get_num_threads()
    # torch's intra-op pool is fastest at ~16 threads on the 256-core hosts of this pool (scripts/cpu_probe.py: the batched GEMMs
    # take 9 ms at 16 threads, 12 ms at 64; the elementwise formulation 62 / 217 ms; neither finishes in minutes at 256)
    nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    try:
        g = torch.Generator().manual_seed(0)
        c, i = CORR, IMG
        nb = c["B"]
        a = torch.randn(nb, c["C"], c["H"], c["W"], generator=g)
        b = torch.randn(nb, c["C"], c["H"], c["W"], generator=g)
        go = torch.randn(nb, 441, c["H"], c["W"], generator=g)
        img = torch.rand(nb, 3, i["H"], i["W"], generator=g) - 0.5
        flow = torch.randn(nb, 2, i["H"], i["W"], generator=g) * 4
        gw = torch.randn(nb, 3, i["H"], i["W"], generator=g)
        gn = torch.randn(nb, 1, i["H"], i["W"], generator=g)

        def one_step():
            with torch.no_grad():
                torch_corr_fwd(a, b)
                torch_corr_bwd(a, b, go)
                w = torch_resample_fwd(img, flow)
                n = torch_chnorm_fwd(w)
                torch_chnorm_bwd(w, n, gn)
            torch_resample_bwd(img, flow, gw)

        t0 = time.perf_counter()
        one_step()
        warm = time.perf_counter() - t0
        times = []
        budget = max(0.0, max_seconds - warm)
        while warm <= max_seconds:               # (a warm-up step that alone exceeds the budget is the measurement)
            t0 = time.perf_counter()
            one_step()
            times.append(time.perf_counter() - t0)
            if len(times) >= 9 or sum(times) + times[-1] > budget:
                break
        if not times:
            times = [warm]
        med = sorted(times)[len(times) // 2]
        return {"value": round(nb / med, 3), "unit": "image-pairs/s", "cores": nthreads, "host_cores": os.cpu_count(), "kind": "port",
                "sample": f"{len(times)} whole steps of the same workload (batch 8) after 1 warm-up step, median; fp32; vectorised PyTorch "
                          f"{torch.__version__} with torch.set_num_threads({nthreads}) (its fastest setting on this host, scripts/cpu_probe.py): "
                          "one batched GEMM per displacement row and image row + diagonal gather, explicit backward in the same shape "
                          "(correlation); grid_sample border/align_corners + autograd (warp); closed-form norm gradient -- a vectorised "
                          "formulation beside `cpu_baseline` (the bit-exact scalar restatement on all host cores, which is about as fast); "
                          "the reference has no CPU path",
                "seconds_per_step": round(med, 4)}
    finally:
        torch.set_num_threads(nthreads_before)


def pmc_traffic(timeout=170):
    """Fabric traffic of the graded kernel MEASURED IN THIS RUN: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one
    pass; kernel trace only alongside, as MI355X_MICROARCH.md prescribes) over a child process that launches the correlation
    forward of the bench workload a few times through the same pybind module.  Returns (dict or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not found on this box"
    child = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import torch, correlation_cuda\n"
        "g = torch.Generator().manual_seed(1234)\n"
        "a = torch.randn(%d, %d, %d, %d, generator=g).cuda(); b = torch.randn(a.shape, generator=g).cuda()\n"
        "e = a.new_empty; s1, s2, out = e(0), e(0), e(0)\n"
        "for _ in range(8): correlation_cuda.forward(a, b, s1, s2, out, %d, %d, %d, %d, %d, 1)\n"
        "torch.cuda.synchronize()\n" % (ROOT, PKG, CORR["B"], CORR["C"], CORR["H"], CORR["W"], CORR["pad"], CORR["k"], CORR["md"], CORR["s1"], CORR["s2"]))
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fn2_pmc_")
        try:
            subprocess.run([rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, "-c", child],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in files:
                for r in csv.DictReader(open(f)):
                    if "corr_fwd" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        vals.append(float(r["Counter_Value"]))
            vals = vals[2:]                      # the first launches carry the code-object load / cold caches
            if not vals:
                return None, f"rocprofv3 --pmc {counter}: no correlation-forward dispatches in its output"
            raw[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:
            return None, f"rocprofv3 --pmc {counter} failed: {exc!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # MI355X_MICROARCH.md (HBM): both counters are in KB; gfx950 tallies wide (16 B per lane) coalesced reads at HALF their bytes --
    # the kernel's inputs arrive by 16-byte buffer loads -> FETCH_SIZE x 2; WRITE_SIZE as reported (uncalibrated)
    read_b, write_b = 2.0 * raw["FETCH_SIZE"][0] * 1024, raw["WRITE_SIZE"][0] * 1024
    return ({"bytes_per_launch": read_b + write_b, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
             "FETCH_SIZE_KB_raw": raw["FETCH_SIZE"][0], "WRITE_SIZE_KB_raw": raw["WRITE_SIZE"][0], "launches_averaged": raw["FETCH_SIZE"][1]},
            "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass each, over a child process that launches "
            "the same correlation forward 8 times (first 2 dropped); FETCH_SIZE x 2 per the guide's gfx950 correction, WRITE_SIZE as reported")


def flownet2c_pass(dev, rank, world, steps, warmup):
    """the whole FlowNet2C network (harness/) around the HIP layers, bs 8 per GPU at
    384x512, synthetic data, fp32: training step (forward, MultiScale-L1 loss, backward with overlapped bucketed RCCL
    all-reduce, Adam), forward+backward alone, and inference.  Reported next to the hot-path numbers, never as `value`."""
    import dist_utils
    from harness.train import Trainer, synthetic_batch, time_steps
    tr = Trainer(dev)
    inputs, target = synthetic_batch(IMG["B"], IMG["H"], IMG["W"], dev, seed=10 + rank)

    def fwd_bwd():
        tr.model.train()
        tr.reducer.zero_grad()
        tr.reducer.reset()
        tr.criterion(tr.model(inputs), target)[0].backward()
        tr.reducer.finish()

    def phase(fn):
        """One timed phase on every rank.  A rank that fails locally says so through the same MAX all-reduce the others are
        about to enter, so that all ranks leave the pass together instead of blocking in the next collective.
        (A failure INSIDE a phase's own gradient all-reduce still needs the watchdog / the launcher to end the job.)"""
        try:
            t, err = fn(), None
        except Exception as exc:
            t, err = float("inf"), exc
        t = dist_utils.max_over_ranks(t, device=dev)
        if t == float("inf"):
            raise RuntimeError(f"FlowNet2C pass failed on a rank: {err!r}" if err is not None else "FlowNet2C pass failed on another rank")
        return t

    if world > 1:
        # N ranks each running MIOpen's kernel search at once is how a scaling run ends in the watchdog: rank 0 searches first (its
        # results land in the user find database all ranks of the node share), the others then warm up against that database
        # The staggered warm-up must be COLLECTIVE-FREE: train_step() issues the bucketed gradient all-reduces, and a
        # rank that runs them alone while the others sit in the barrier pairs a 48 MB all-reduce with the barrier's 1-element one
        # (undefined on RCCL, a size mismatch on gloo).  Forward + loss + backward with the reducer's collectives switched off
        # finds the same MIOpen kernels; the optimizer's elementwise kernels need no search.
        def warm():
            was = tr.reducer.collective
            tr.reducer.collective = False
            try:
                fwd_bwd()
                tr.infer(inputs)
                torch.cuda.synchronize()
            finally:
                tr.reducer.collective = was
        sync = (lambda: torch.distributed.barrier(device_ids=[dev.index])) if torch.distributed.get_backend() == "nccl" else torch.distributed.barrier
        if rank == 0:
            warm()
        sync()
        if rank != 0:
            warm()
        sync()
    t_train = phase(lambda: time_steps(lambda: tr.train_step(inputs, target), steps, warmup, dev))
    t_fb = phase(lambda: time_steps(fwd_bwd, steps, warmup, dev))
    t_inf = phase(lambda: time_steps(lambda: tr.infer(inputs), steps, warmup, dev))
    pairs = IMG["B"] * world
    # BASELINE.json configs[3]: the full FlowNet2 stack (CSS + SD + fusion, 162.5 M parameters), inference, fp32 and with fp16
    # convolution stacks (the custom layers keep fp32 operands); rank-local, no collective
    full = {}
    n_buckets = len(tr.reducer.buckets)
    try:
        from harness.flownet2 import FlowNet2
        tr = None                         # release the FlowNet2C replica, its gradients and optimizer state
        torch.manual_seed(3)
        net = FlowNet2().to(dev).eval()
        with torch.no_grad():
            t32 = time_steps(lambda: net(inputs), steps, warmup, dev)
            net16, in16 = net.half(), inputs.half()
            t16 = time_steps(lambda: net16(in16), steps, warmup, dev)
        t32 = dist_utils.max_over_ranks(t32, device=dev)
        t16 = dist_utils.max_over_ranks(t16, device=dev)
        full = {"flownet2_inference_ms_fp32": round(t32 * 1e3, 3), "flownet2_inference_pairs_per_s_fp32": round(pairs / t32, 1),
                "flownet2_inference_ms_fp16": round(t16 * 1e3, 3), "flownet2_inference_pairs_per_s_fp16": round(pairs / t16, 1)}
    except Exception as exc:
        full = {"flownet2_error": repr(exc)}
    return {**full, "model": "FlowNet2C (39 175 298 parameters, random init), bs 8 per GPU @384x512, fp32, synthetic",
            "steps": steps, "warmup": warmup,
            "train_step_ms": round(t_train * 1e3, 3), "train_image_pairs_per_s": round(pairs / t_train, 1),
            "fwd_bwd_ms": round(t_fb * 1e3, 3), "fwd_bwd_image_pairs_per_s": round(pairs / t_fb, 1),
            "inference_ms": round(t_inf * 1e3, 3), "inference_image_pairs_per_s": round(pairs / t_inf, 1),
            "grad_buckets": n_buckets, "parallelism": f"dp{world}: replica per GPU, bucketed all-reduce overlapped with backward"}


def _parse_cpulist(text):
    cores = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cores.update(range(int(lo), int(hi or lo) + 1))
    return cores


def rank_cpu_cores(local_rank, local_world, available, gpu_cpulists=None):
    """The host cores rank `local_rank` of `local_world` pins itself to (pure function; tests/test_bench_launch.py).  A 0.19 ms step is
    launched by one host thread: a rank whose thread shares a core with another rank's (or with the box's housekeeping) becomes
    "the slowest rank" of a max-over-ranks timing.  `gpu_cpulists[d]` = the cores next to GPU d (its PCI device's local_cpulist) or
    None: ranks whose GPUs sit on the same NUMA node split that node's cores evenly; without topology the available cores are
    split evenly.  Never returns an empty set (falls back to everything available)."""
    avail = sorted(available)
    if local_world <= 1 or not avail:
        return set(avail)
    pool, idx, n = avail, local_rank, local_world
    if gpu_cpulists and local_rank < len(gpu_cpulists) and gpu_cpulists[local_rank]:
        near = sorted(set(gpu_cpulists[local_rank]) & set(avail))
        peers = [r for r in range(min(local_world, len(gpu_cpulists))) if gpu_cpulists[r] == gpu_cpulists[local_rank]]
        if near and len(near) >= len(peers):
            pool, idx, n = near, peers.index(local_rank), len(peers)
    per = len(pool) // n
    if per < 1:
        return set(avail)
    return set(pool[idx * per:(idx + 1) * per])


def gpu_identity(index):
    """Where THIS rank's GPU sits: PCI bus id, the cores next to it, its XGMI hive (KFD topology).  Every field is best effort
    (None when the box does not expose it): the N-GPU line records it so that a slow rank can be named by slot, not by number."""
    info = {"hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"), "rocr_visible_devices": os.environ.get("ROCR_VISIBLE_DEVICES"),
            "pci_bus_id": None, "numa_node": None, "local_cpulist": None, "xgmi_hive_id": None}
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        info["pci_bus_id"] = bdf
        base = "/sys/bus/pci/devices/" + bdf
        if os.path.exists(base + "/numa_node"):
            info["numa_node"] = int(open(base + "/numa_node").read())
        if os.path.exists(base + "/local_cpulist"):
            info["local_cpulist"] = open(base + "/local_cpulist").read().strip()
        loc = (pr.pci_bus_id << 8) | (pr.pci_device_id << 3)
        topo = "/sys/class/kfd/kfd/topology/nodes"
        for node in sorted(os.listdir(topo)) if os.path.isdir(topo) else []:
            try:
                props = dict(ln.split(None, 1) for ln in open(os.path.join(topo, node, "properties")).read().splitlines() if " " in ln)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0 and int(props.get("location_id", "-1")) == loc and int(props.get("domain", "0")) == pr.pci_domain_id:
                info["xgmi_hive_id"] = props.get("hive_id", "").strip() or None
                break
    except Exception as exc:
        info["error"] = repr(exc)
    return info


def pin_rank_to_cores(local_rank, local_world):
    """sched_setaffinity for this rank (N > 1 only); returns the core list it now runs on, or None when nothing was changed."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("FN2_BENCH_NO_PIN") == "1":
        return None
    try:
        lists = []
        for d in range(min(local_world, torch.cuda.device_count())):
            cl = gpu_identity(d).get("local_cpulist")
            lists.append(frozenset(_parse_cpulist(cl)) if cl else None)
        cores = rank_cpu_cores(local_rank, local_world, os.sched_getaffinity(0), lists)
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(len(cores), 8)))
        return sorted(cores)
    except Exception:
        return None


def launch_plan(gpus, device_count, share):
    """How `--gpus N` is started when no launcher set WORLD_SIZE: the per-rank environments (one process per GPU, RCCL over
    127.0.0.1 rendezvous) or an error string.  Pure function: tests/test_bench_launch.py covers it on CPU."""
    if gpus < 1:
        return None, "--gpus must be >= 1"
    if gpus == 1:
        return [], None                       # run in this process
    if device_count < gpus and not share:
        return None, (f"--gpus {gpus} but only {device_count} GPU(s) visible (FN2_BENCH_SHARE_GPU=1 runs the N-rank control "
                      "flow on one GPU over gloo as a dry run)")
    return [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "MASTER_ADDR": "127.0.0.1",
             "LOCAL_WORLD_SIZE": str(gpus)} for r in range(gpus)], None


def self_launch(argv, gpus, share):
    """`python bench.py --gpus N` (N > 1, no torchrun): re-runs this script as N ranks, one per GPU, rank 0 inheriting stdout
    so that the ONE JSON line is its line (the reference scales with one command too: main.py:187-201).  Returns the exit code."""
    import subprocess
    import tempfile
    plan, err = launch_plan(gpus, torch.cuda.device_count(), share)
    if err:
        print("[bench] " + err, file=sys.stderr, flush=True)
        return 2
    # rendezvous through a fresh file (dist_utils.init_from_env, FN2_INIT_FILE): no port to lose between choosing and binding it
    rdv_dir = tempfile.mkdtemp(prefix="fn2_bench_rdv_")
    rdv = os.path.join(rdv_dir, "store")
    procs = []
    for env_r in plan:
        env = dict(os.environ, **env_r, FN2_INIT_FILE=rdv)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if env_r["RANK"] == "0" else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    deadline = time.time() + float(os.environ.get("FN2_BENCH_LAUNCH_TIMEOUT", "1500"))
    live = list(procs)
    while live:
        for pr in list(live):
            r = pr.poll()
            if r is not None:
                live.remove(pr)
                if r != 0 and rc == 0:
                    rc = r
        if (rc != 0 or time.time() > deadline) and live:   # one rank failed (or the job hangs): stop exactly the ranks we started
            for pr in live:
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    try:
        if os.path.exists(rdv):
            os.remove(rdv)
        os.rmdir(rdv_dir)
    except OSError:
        pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=("auto", "on", "off"),
                    help="replay a hipGraph of the step instead of launching it eagerly; auto = eager at one GPU (the documented "
                         "headline), graph replay for --gpus N > 1: one launch per step, so that a noisy host core cannot make a "
                         "rank the slowest of a max-over-ranks timing")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-fast-seconds", type=float, default=12.0)
    ap.add_argument("--model", choices=("auto", "on", "off"), default="auto",
                    help="also time the whole FlowNet2C network around the layers (extra key `flownet2c`; never `value`); auto = on "
                         "for one GPU, off for --gpus N > 1 (N ranks each running MIOpen's kernel search under the watchdog is how a "
                         "scaling run ends with a truncated line)")
    ap.add_argument("--pmc", choices=("auto", "on", "off"), default="auto",
                    help="measure the graded kernel's fabric traffic in THIS run (two rocprofv3 --pmc passes over a tiny child "
                         "process, one GPU only); auto = on where rocprofv3 is installed")
    ap.add_argument("--placement", choices=("auto", "on", "off"), default="auto",
                    help="untimed set-up: re-place the step's buffers when the step is > 10 %% above the sum of its kernels (auto), always try (on), never (off)")
    ap.add_argument("--model-steps", type=int, default=20)
    ap.add_argument("--model-warmup", type=int, default=5)
    ap.add_argument("--model-timeout", type=float, default=240.0, help="seconds after which the FlowNet2C pass is abandoned")
    args = ap.parse_args()

    import dist_utils
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP kernels are the only implementation"
    # FN2_BENCH_SHARE_GPU=1: dry run of the N-rank control flow on a box with ONE GPU (all ranks on cuda:0, gloo instead of
    # RCCL) -- checks the code path, measures nothing; the JSON line says so
    share = os.environ.get("FN2_BENCH_SHARE_GPU") == "1"
    if args.gpus != 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus, share))
    rank, world, local_rank = dist_utils.init_from_env(backend="gloo" if share else None)   # "nccl" = RCCL; one process per GPU
    dist = torch.distributed if world > 1 else None
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pinned = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), local_world)   # before any worker thread exists

    hp = HotPath(dev, seed=1234 + rank)   # every rank its own batch: no cross-GPU dependence
    # set-up, before the W warm-up steps: code objects loaded, allocator settled, clocks up (a cold MI355X runs the first
    # ~50 steps 7 % slower than steady state; measured 0.328 -> 0.305 ms on the same box)
    for _ in range(100):
        hp.step()
    torch.cuda.synchronize()
    # Placement of the step's buffers, chosen in the UNTIMED set-up (round 6).  About every second box of the pool runs the raw step 20-25 %
    # slower at identical kernel times; on such a box the same kernels through the autograd modules -- whose tensors lie elsewhere -- ran at
    # the fast boxes' speed (0.184 against 0.249 ms, profiles/r06_a_*): where the buffers lie matters there, and only there (on a fast box
    # four placements are within 1 %, profiles/r06_c_placement_probe.log).  So: time the step as allocated; if it is more than 10 % above
    # the sum of its kernels' own durations (the slow-box signature), move every tensor to a fresh allocation -- up to three times -- and
    # keep the fastest placement.  Same tensors, same kernels, same K timed steps afterwards; what was tried is in the JSON line.
    def _clock_steps(n=40):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            hp.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / n * 1e3
    placement = {"tried_ms_per_step": [round(_clock_steps(), 4)], "kernel_sum_ms": None, "chosen": 0}
    if args.placement != "off":
        ev_ = {}
        for _ in range(10):
            hp.step(ev_)
        torch.cuda.synchronize()
        ksum = sum(sum(s_.elapsed_time(e_) for s_, e_ in v) / len(v) for v in ev_.values())
        placement["kernel_sum_ms"] = round(ksum, 4)
        if args.placement == "on" or placement["tried_ms_per_step"][0] > 1.10 * ksum:
            names = [n_ for n_, t_ in vars(hp).items() if torch.is_tensor(t_) and t_.numel()]
            best = {n_: getattr(hp, n_) for n_ in names}
            held = []
            for trial in range(3):
                held.append({n_: getattr(hp, n_) for n_ in names})           # the old blocks stay allocated: the clones must land elsewhere
                for n_ in names:
                    setattr(hp, n_, getattr(hp, n_).clone())
                for _ in range(20):
                    hp.step()
                placement["tried_ms_per_step"].append(round(_clock_steps(), 4))
                if placement["tried_ms_per_step"][-1] < min(placement["tried_ms_per_step"][:-1]):
                    best = {n_: getattr(hp, n_) for n_ in names}
                    placement["chosen"] = trial + 1
            for n_, t_ in best.items():
                setattr(hp, n_, t_)
            del held
            torch.cuda.empty_cache()
    for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
        hp.step()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            if share:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    # The step is seven short launches (0.19 ms of kernels).  The timed region launches them eagerly with an event
    # pair around the graded kernel (correlation forward) in every fourth step; event pairs around all six ops cost 40 us per
    # step, so the other kernels are timed in a second pass.  --graph replays a hipGraph of the step instead (no faster
    # than eager launches once the device is warm: 0.305 ms either way; its roofline then comes from the second pass).
    graph = None
    if args.graph == "on" or (args.graph == "auto" and world > 1):
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                hp.step()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as exc:   # capture unsupported on this stack: say so and time eager launches
            print(f"[bench] hipGraph capture failed ({exc!r}); timing eager launches", file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()

    EV_EVERY = 4
    ev_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)]
    for s_, e_ in ev_pool:      # the underlying HIP events are created by their first record: do that here
        s_.record(); e_.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the graded kernel is timed INSIDE the timed region: an event pair around it in every fourth step, on events created and recorded once
    # before the region (creating a HIP event costs the host more than launching a kernel; with a pair per step the host of some boxes
    # could not keep the 0.18 ms step fed: 0.208 against 0.178 ms, profiles/r06_q_*)
    roof_events = {"corr_fwd": ev_pool}
    for i in range(args.steps):
        if graph is not None:
            graph.replay()
        elif i % EV_EVERY == 0:
            s_, e_ = ev_pool[i // EV_EVERY]
            s_.record(); hp.corr_fwd(); e_.record()
            hp.step_rest()
        else:
            hp.step_plain()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    elapsed = dist_utils.max_over_ranks(elapsed, device=dev)   # the step time is the slowest rank's
    # per-rank consistency check: every rank's own time for its K steps and a checksum of its results
    chk = float(hp.out.double().sum().item() + hp.g1.double().sum().item() + hp.norm.double().sum().item())
    # two probes of THIS rank's GPU, so that an N-GPU line explains itself (the step time is the slowest rank's): the graded kernel
    # alone, back to back (operands cache-resident), and a 256 MiB device copy
    def rank_probes():
        ts = []
        for _ in range(9):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); hp.corr_fwd(); e_.record(); torch.cuda.synchronize()
            ts.append(s_.elapsed_time(e_) * 1e3)
        a_, b_ = torch.empty(1 << 26, device=dev), torch.empty(1 << 26, device=dev)
        b_.copy_(a_)
        cs = []
        for _ in range(5):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); b_.copy_(a_); e_.record(); torch.cuda.synchronize()
            cs.append(s_.elapsed_time(e_))
        return sorted(ts)[len(ts) // 2], 2 * a_.numel() * 4 / (min(cs) * 1e-3) / 1e9
    probe_us, probe_copy = rank_probes()
    mine = torch.tensor([own_elapsed, chk, float(torch.cuda.current_device()), probe_us, probe_copy], dtype=torch.float64, device=dev)
    if dist is not None:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    per_rank = [{"rank": r, "device": int(t[2].item()), "ms_per_step": round(float(t[0].item()) / args.steps * 1e3, 4),
                 "image_pairs_per_s": round(CORR["B"] * args.steps / float(t[0].item()), 1),
                 "corr_fwd_us_warm": round(float(t[3].item()), 1), "copy_GBps_torch": round(float(t[4].item()), 1),
                 "finite": bool(torch.isfinite(t[1]).item())} for r, t in enumerate(allr)]
    # where each rank ran: the GPU's slot (PCI bus id, NUMA node, XGMI hive) and the host cores the rank pinned itself to
    ident = dict(gpu_identity(torch.cuda.current_device()), cpu_cores=(f"{pinned[0]}-{pinned[-1]} ({len(pinned)})" if pinned else None),
                 host=os.uname().nodename, pid=os.getpid())
    if dist is not None:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    else:
        idents = [ident]
    for pr_, id_ in zip(per_rank, idents):
        pr_.update(id_)
    # how far the slowest rank is behind the fastest: `value` = N x pairs / slowest rank's time, so this is the run's scaling
    # efficiency relative to N copies of its own fastest rank (the driver computes efficiency across runs from `value` itself)
    rank_balance = round(min(float(t[0].item()) for t in allr) / max(float(t[0].item()) for t in allr), 4)

    # Per-kernel durations: the same K steps once more with a HIP event pair around every op on the launch stream
    # (same kernels, same inputs; rocprofv3's per-kernel averages of this command agree).
    events = {}
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        hp.step(events)
    torch.cuda.synchronize()
    eager_elapsed = time.perf_counter() - t1

    # The same step through the shipped autograd wrappers (Correlation / Resample2d / ChannelNorm modules, with the
    # difference op of models.py:135 between warp and norm): what a training script pays, allocations included.
    mod_elapsed, mod_fused_elapsed, mod_same_elapsed, mod_floor_elapsed = hp.module_steps(args.steps)

    if rank == 0:
        per_op_ms = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in events.items()}
        if roof_events.get("corr_fwd") and graph is None:   # roofline: the durations recorded inside the timed region
            v = roof_events["corr_fwd"]
            per_op_ms["corr_fwd"] = sum(s.elapsed_time(e) for s, e in v) / len(v)
        ab = algorithmic_bytes(CORR["B"])
        kernels = {k: {"ms": round(ms, 5), "algorithmic_bytes": ab[k],
                       "achieved_GBps": round(ab[k] / (ms * 1e-3) / 1e9, 2),
                       "frac_of_8TBps": round(ab[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                   for k, ms in per_op_ms.items()}
        cf = kernels["corr_fwd"]
        # HBM-side traffic of the graded kernel: NOT measured in this process (PMC counters need their own rocprofv3 --pmc
        # passes, scripts/gpu_traffic.sh); the tracked result of those passes is quoted with its source
        traffic, traffic_src, traffic_detail = None, None, None
        if world == 1 and args.pmc != "off":
            traffic_detail, traffic_src = pmc_traffic()
            if traffic_detail is not None:
                traffic = traffic_detail["bytes_per_launch"]
        tpath = os.path.join(ROOT, "profiles", "corr_fwd_hbm_traffic.json")
        if traffic is None and os.path.exists(tpath):
            why_not = traffic_src
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("bytes_per_launch")
                traffic_src = ("quoted from profiles/corr_fwd_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                               "passes, FETCH_SIZE doubled per the guide's gfx950 correction; kernel source at commit "
                               f"{tj.get('commit', 'unrecorded')}), not measured in this run"
                               + (f" ({why_not})" if why_not else ""))
            except Exception:
                traffic = None
        # what an event pair adds around ONE short kernel (launch latency + inter-packet gaps): a 4-byte fill
        tiny = torch.zeros(1, device=dev)
        nev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s_, e_ in nev:
            s_.record(); tiny.zero_(); e_.record()
        torch.cuda.synchronize()
        event_floor_ms = sorted(s_.elapsed_time(e_) for s_, e_ in nev)[len(nev) // 2]
        pairs = CORR["B"] * args.steps * world
        # empirical streaming ceiling of this box: device-to-device copy of 1 GiB, read + write bytes -- a float4
        # grid-stride kernel (fn2_debug_stream_copy in the debug library; the guide quotes 6.29 TB/s for such a copy), best of a
        # few grid sizes and of temporal / non-temporal accesses; torch's copy_ for comparison
        src = torch.empty(1 << 28, device=dev, dtype=torch.float32)
        dst = torch.empty_like(src)

        def best_gbs(fn, reps=5):
            fn()
            cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for s_, e_ in cev:
                s_.record(); fn(); e_.record()
            torch.cuda.synchronize()
            return 2 * src.numel() * 4 / (min(s_.elapsed_time(e_) for s_, e_ in cev) * 1e-3) / 1e9

        copy_torch = best_gbs(lambda: dst.copy_(src))
        copy_gbs, copy_kernel = copy_torch, "torch Tensor.copy_"
        try:
            import ctypes
            import fn2_capi
            dl = fn2_capi.debug_lib()
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for blocks in (2048, 4096, 8192, 16384):
                for nt in (0, 1):
                    g_ = best_gbs(lambda: fn2_capi.check(dl.fn2_debug_stream_copy(
                        ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(src.numel() * 4),
                        blocks, nt, st), "fn2_debug_stream_copy"), reps=3)
                    if g_ > copy_gbs:
                        copy_gbs, copy_kernel = g_, f"float4 grid-stride copy, {blocks} x 256 lanes, {'non-' if nt else ''}temporal"
        except Exception as exc:   # debug library not built: keep torch's number and say so
            copy_kernel += f" (fn2_debug_stream_copy unavailable: {exc!r})"
        del src, dst
        # what this box sustains on the matrix pipe (register-only f16 MFMA stream on every SIMD): the boxes of the pool differ by
        # up to 20 % on the whole step; with the copy ceiling this tells a slow clock from slow memory
        mfma_tflops = None
        try:
            import ctypes
            import fn2_capi
            dl = fn2_capi.debug_lib()
            sink = torch.zeros(1024, device=dev)
            flop = ctypes.c_double(0.0)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            run = lambda: fn2_capi.check(dl.fn2_debug_mfma_probe(ctypes.c_void_p(sink.data_ptr()), 4000, 512, ctypes.byref(flop), st),
                                         "fn2_debug_mfma_probe")
            run()
            # 40 launches back to back (~0.4 s of sustained matrix load): the rate of the first and of the last five
            pev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for s_, e_ in pev:
                s_.record(); run(); e_.record()
            torch.cuda.synchronize()
            rate = lambda evs: round(flop.value / (sum(s_.elapsed_time(e_) for s_, e_ in evs) / len(evs) * 1e-3) / 1e12, 1)
            mfma_tflops = {"first5": rate(pev[:5]), "last5": rate(pev[-5:])}
        except Exception:
            mfma_tflops = None
        # where workgroups run: the tile orders of the kernels assume "workgroup b on XCD b % 8" (for speed only)
        xcd_census = None
        try:
            import ctypes
            import fn2_capi
            dl = fn2_capi.debug_lib()
            ids = torch.full((4096,), -1, dtype=torch.int32, device=dev)
            fn2_capi.check(dl.fn2_debug_xcc_census(ctypes.c_void_p(ids.data_ptr()), 4096, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           "fn2_debug_xcc_census")
            torch.cuda.synchronize()
            ids = ids.cpu()
            xcd_census = {"workgroups": 4096, "xcds_seen": int(ids.unique().numel()),
                          "not_on_xcd_b_mod_8": int((ids != (torch.arange(4096, dtype=torch.int32) % 8)).sum())}
        except Exception:
            xcd_census = None
        # cold-operand sensitivity of the graded kernel on this box: median HIP-event time back-to-back (operands cache-resident)
        # and right after a 1 GiB fill (operands evicted from every cache level)
        junk = torch.empty(1 << 28, device=dev)
        def corr_us(thrash):
            ts = []
            for _ in range(9):
                if thrash:
                    junk.fill_(1.0)
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(); hp.corr_fwd(); e_.record(); torch.cuda.synchronize()
                ts.append(s_.elapsed_time(e_) * 1e3)
            return round(sorted(ts)[len(ts) // 2], 1)
        cold = {"corr_fwd_us_warm": corr_us(False), "corr_fwd_us_after_1GiB_fill": corr_us(True)}
        del junk
        line = {
            "metric": "image-pairs/sec, FlowNet2 custom-layer hot path (Correlation + Resample2d + ChannelNorm) fwd+bwd @ 384x512 "
                      "bs8 -- the whole FlowNet2C network's fwd+bwd image-pairs/sec is flownet2c.fwd_bwd_image_pairs_per_s; "
                      "corr-layer HBM GB/s vs roofline is `roofline`",
            "value": round(pairs / elapsed, 3),
            "unit": "image-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "launch": "hipGraph replay of the step" if graph is not None else "eager launches",
            "placement_autotune": placement,   # untimed set-up: step time per buffer placement tried, the one the timed steps ran with
            "ms_per_step_eager_with_events": round(eager_elapsed / args.steps * 1e3, 4),
            "ms_per_step_autograd_modules": round(mod_elapsed / args.steps * 1e3, 4),
            "ms_per_step_autograd_fused_rows": round(mod_fused_elapsed / args.steps * 1e3, 4),   # Correlation + WarpDiffNorm (flow gradient only)
            # the kernels of `ms_per_step` and nothing else (no difference op), through nn.Module + autograd (C++ nodes since round 6)
            "ms_per_step_autograd_modules_same_kernels": round(mod_same_elapsed / args.steps * 1e3, 4),
            # ... and what two .backward() calls of trivial graphs cost on this host (no kernels to speak of): the floor under both
            "ms_per_step_autograd_host_floor": round(mod_floor_elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "FlowNet2 custom-layer hot path fwd+bwd, bs8 @384x512: Correlation 8x256x48x64 "
                            "(20,1,20,1,2) [BASELINE configs[1]] + Resample2d 8x3x384x512 + ChannelNorm 8x3x384x512",
                "pairs_per_step_per_gpu": CORR["B"],
                "sharding": "batch over ranks, no data-path collective",
            },
            "configs1_correlation_only": {   # BASELINE.json configs[1]: correlation layer alone, fwd+bwd, 8x256x48x64 fp32
                "ms_fwd_bwd": round(kernels["corr_fwd"]["ms"] + kernels["corr_bwd"]["ms"], 5),
                "image_pairs_per_s_per_gpu": round(CORR["B"] / ((kernels["corr_fwd"]["ms"] + kernels["corr_bwd"]["ms"]) * 1e-3), 1),
            },
            "roofline": {
                "kernel": "correlation forward (corr_fwd_f16x2)",
                "bound": "hbm",
                "achieved": cf["achieved_GBps"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(cf["achieved_GBps"] / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_detail": traffic_detail,
                "traffic_over_algorithmic": round(traffic / cf["algorithmic_bytes"], 4) if traffic else None,
                "frac_of_copy_ceiling": round(cf["achieved_GBps"] / copy_gbs, 4),
                "copy_ceiling_GBps": round(copy_gbs, 1),
                "copy_ceiling_kernel": copy_kernel,
                "copy_ceiling_torch_copy_GBps": round(copy_torch, 1),
                "launch_ms": cf["ms"],
                "event_pair_floor_ms": round(event_floor_ms, 5),   # the same event pair around a 4-byte fill: rocprofv3's kernel
                                                                  # durations are shorter than launch_ms by about this much
                "algorithmic_bytes": cf["algorithmic_bytes"],
            },
            # the same kernel against the matrix-core roof: 3 f16 MFMAs (two-term operand split) per 16x16x32 block product;
            # 480 tasks with neighbour rows inside the image x 8 channel steps x 8 matrix waves x 33 MFMAs x 16384 FLOP
            "roofline_mfma": (lambda mfma_flop: {
                "kernel": "correlation forward (corr_fwd_f16x2)", "bound": "mfma",
                "achieved": round(mfma_flop / (cf["ms"] * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(mfma_flop / (cf["ms"] * 1e-3) / 1e12 / 2500.0, 4),
                "issued_f16_flop_per_launch": mfma_flop,
                "useful_fp32_flop_per_launch": 2 * CORR["B"] * CORR["H"] * CORR["W"] * 441 * CORR["C"],
                "note": "issued = 3 products of the two-term f16 split x the 4x4-block band (44 of 64 block pairs per "
                        "row-block pair); the kernel is bound by instruction issue and LDS, not by the matrix pipe (DESIGN.md 4.1)"})(
                480 * 8 * 8 * 33 * 16384),
            "kernels": kernels,
            "per_rank": per_rank,
            "rank_balance_fastest_over_slowest": rank_balance,
            "box": {"mfma_probe_TFLOPs": mfma_tflops, "copy_ceiling_GBps": round(copy_gbs, 1), **cold, "xcd_placement": xcd_census,
                    "note": "probes of THIS box: a register-only f16 MFMA stream on every SIMD (first / last five of 40 back-to-back launches; dense "
                            "peak 2500), the streaming copy, and the graded kernel timed alone (operands cache-resident / after a 1 GiB "
                            "fill).  About every second box of the pool runs the whole step 20 % slower (0.23-0.24 vs 0.195-0.205 ms at the "
                            "closing build of round 3, eager and as a hipGraph alike) although all of these probes agree within a few percent across boxes "
                            "(DESIGN.md 5)"},
        }
        assert len(per_rank) == world and all(r["finite"] for r in per_rank), per_rank
        if share:
            line["dry_run_shared_gpu"] = True
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            line["gpu_over_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
            # a vectorised-PyTorch formulation of the same step beside the bit-exact restatement
            try:
                line["cpu_baseline_fast"] = cpu_baseline_fast(args.cpu_fast_seconds)
                line["gpu_over_cpu_fast"] = round(line["value"] / line["cpu_baseline_fast"]["value"], 1)
            except Exception as exc:
                line["cpu_baseline_fast"] = {"error": repr(exc)}
    else:
        line = None

    # The whole FlowNet2C network around the layers (extra key, never `value`).  All ranks take part (gradient all-reduce);
    # a watchdog guarantees the one JSON line even if this pass stalls: it prints the hot-path line and ends the process.
    if args.model == "on" or (args.model == "auto" and world == 1):
        import threading

        def give_up():
            if rank == 0:
                line["flownet2c"] = {"error": f"no result within {args.model_timeout} s"}
                print(json.dumps(line), flush=True)
            os._exit(0)

        dog = threading.Timer(args.model_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            model_line = flownet2c_pass(dev, rank, world, args.model_steps, args.model_warmup)
        except Exception as exc:   # the hot-path line must not depend on MIOpen finding its kernels
            print(f"[bench] FlowNet2C pass failed: {exc!r}", file=sys.stderr, flush=True)
            model_line = {"error": repr(exc)}
        dog.cancel()
        if rank == 0:
            line["flownet2c"] = model_line
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
