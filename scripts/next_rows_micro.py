"""Timing of the fused "next" rows against the unfused statement sequences of the reference models
on the same HIP layers.  Prints one JSON object; run on the GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import torch.nn.functional as F
from networks.correlation_package.correlation import Correlation, CorrelationLeakyReLUCat
from networks.resample2d_package.resample2d import Resample2d, WarpDiffNorm, WarpDiffNormCat
from networks.channelnorm_package.channelnorm import ChannelNorm

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return round(ts[len(ts) // 2], 1)


res = {}
with torch.no_grad():
    # N1: FlowNetC.py:86-92 at bs 8 @ 384x512 -> conv3 maps 8 x 256 x 48 x 64, conv_redir 8 x 32 x 48 x 64
    a = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    b = torch.randn(8, 256, 48, 64, generator=g).to(dev)
    redir = torch.randn(8, 32, 48, 64, generator=g).to(dev)
    corr, fused = Correlation(20, 1, 20, 1, 2, 1), CorrelationLeakyReLUCat(20, 1, 20, 1, 2, 0.1)
    res["N1_unfused_us"] = timeit(lambda: torch.cat((redir, F.leaky_relu(corr(a, b), 0.1, inplace=True)), 1))
    res["N1_fused_us"] = timeit(lambda: fused(a, b, redir))
    assert torch.equal(fused(a, b, redir), torch.cat((redir, F.leaky_relu(corr(a, b), 0.1)), 1))
    # N2: models.py:133-138 at bs 8 @ 384x512
    x = torch.rand(8, 6, 384, 512, generator=g).to(dev)
    flow = (torch.randn(8, 2, 384, 512, generator=g) * 4.0).to(dev)
    rs, cn, wd = Resample2d(), ChannelNorm(), WarpDiffNormCat(20.0)

    def unfused():
        r = rs(x[:, 3:, :, :], flow)
        d = x[:, :3, :, :] - r
        return torch.cat((x, r, flow / 20.0, cn(d)), dim=1)

    res["N2_unfused_us"] = timeit(unfused)
    res["N2_fused_us"] = timeit(lambda: wd(x, flow))
    assert torch.equal(wd(x, flow), unfused())
    res["N2_algorithmic_bytes"] = (6 + 2 + 12) * 8 * 384 * 512 * 4
    res["N2_fused_GBps"] = round(res["N2_algorithmic_bytes"] / (res["N2_fused_us"] * 1e-6) / 1e9, 1)
# N3: losses.py:52-86 (MultiScale, L1) + EPE at bs 8 @ 384x512, forward + backward to the five predictions
from losses_fused import MultiScaleL1
target = (torch.randn(8, 2, 384, 512, generator=g) * 5.0).to(dev)
outs = [(torch.randn(8, 2, 384 // (4 << i), 512 // (4 << i), generator=g) * 0.3).to(dev).requires_grad_(True) for i in range(5)]
weights = [0.32 / 2 ** i for i in range(5)]
pools = [torch.nn.AvgPool2d(4 << i, 4 << i) for i in range(5)]


def ref_loss():
    for o in outs:
        o.grad = None
    t = 0.05 * target
    loss, epe = 0, 0
    for i, o in enumerate(outs):
        ti = pools[i](t)
        epe = epe + weights[i] * torch.norm(ti - o, p=2, dim=1).mean()
        loss = loss + weights[i] * torch.abs(o - ti).mean()
    loss.backward()
    return loss, epe


crit = MultiScaleL1()


def fused_loss():
    for o in outs:
        o.grad = None                # (as an optimizer's zero_grad(set_to_none=True) does: no accumulation kernels in the timing)
    loss, epe = crit(tuple(outs), target)
    loss.backward()
    return loss, epe


# what ANY two-launch autograd graph costs at module level on this host (Python call, node construction, the engine's hand-over to its
# device thread, AccumulateGrad): a tiny multiply + sum and their backward -- the floor under N3's forward + backward, whose two kernels
# take ~20 us on the GPU
tiny = torch.randn(64, device=dev, requires_grad=True)


def floor_graph():
    tiny.grad = None
    (tiny * 2.0).sum().backward()


res["autograd_two_op_floor_us"] = timeit(floor_graph)
res["N3_unfused_us"] = timeit(ref_loss)
res["N3_fused_us"] = timeit(fused_loss)
# N2, training half (round 5): the backward of models.py:133-138 -- autograd through cat / div / ChannelNorm / sub / Resample2d against
# the one fused kernel; the image pair without a gradient (FlowNet2's case: it is the network's input) and with one
gcat = torch.randn(8, 12, 384, 512, generator=g).to(dev)
for need_x in (False, True):
    xl, fl = x.clone().requires_grad_(need_x), flow.clone().requires_grad_(True)
    r = rs(xl[:, 3:], fl)
    unf = torch.cat((xl, r, fl / 20.0, cn(xl[:, :3] - r)), dim=1)
    xl2, fl2 = x.clone().requires_grad_(need_x), flow.clone().requires_grad_(True)
    fus = wd(xl2, fl2)
    key = "N2_bwd_pair_grad_" if need_x else "N2_bwd_flow_only_"
    res[key + "unfused_us"] = timeit(lambda: unf.backward(gcat, retain_graph=True))
    res[key + "fused_us"] = timeit(lambda: fus.backward(gcat, retain_graph=True))
# N2 without the concat (models.py:157-161, :170-174): ||first image - warped|| and its flow gradient
we = WarpDiffNorm()
gn = torch.randn(8, 1, 384, 512, generator=g).to(dev)
with torch.no_grad():
    res["N2n_unfused_us"] = timeit(lambda: cn(x[:, :3] - rs(x[:, 3:], flow)))
    res["N2n_fused_us"] = timeit(lambda: we(x, flow))
fl = flow.clone().requires_grad_(True)
unf = cn(x[:, :3] - rs(x[:, 3:], fl))
fl2 = flow.clone().requires_grad_(True)
fus = we(x, fl2)
res["N2n_bwd_unfused_us"] = timeit(lambda: unf.backward(gn, retain_graph=True))
res["N2n_bwd_fused_us"] = timeit(lambda: fus.backward(gn, retain_graph=True))
print(json.dumps(res))
