"""On a box of the pool's SLOW class (raw step >= 0.215 ms) -- and only there, a fast box exits at once -- what makes the
back-to-back step slow?  Round 6 observation: on such a box the same kernels through the autograd modules took 0.184 ms against 0.249 ms
for the raw pybind calls, and event pairs around every op made the raw step FASTER (0.239).  Variants of the same seven launches:
as is; re-placed buffers; an event record / an empty kernel / a host sync between launches; through a hipGraph; fresh outputs per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-pytorch_amd")]
import torch
import bench

dev = torch.device("cuda:0")
hp = bench.HotPath(dev, 1234)


def clock(fn, n=200, warm=60):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = clock(hp.step_plain)
print("raw step %.4f ms" % base, flush=True)
if base < 0.215 and "--force" not in sys.argv:
    print("fast box: nothing to probe")
    sys.exit(0)
ops = [lambda: hp.m_corr.forward(hp.in1, hp.in2, hp.scr1, hp.scr2, hp.out, *hp.cparams),
       lambda: hp.m_corr.backward(hp.in1, hp.in2, hp.scr1, hp.scr2, hp.gcorr, hp.g1, hp.g2, *hp.cparams),
       lambda: hp.m_res.forward(hp.img, hp.flow, hp.warped, 1, True),
       lambda: hp.m_cn.forward(hp.warped, hp.norm, 2),
       lambda: hp.m_cn.backward(hp.warped, hp.norm, hp.gnorm, hp.gdiff, 2),
       lambda: hp.gimg.zero_(),
       lambda: hp.m_res.backward(hp.img, hp.flow, hp.gwarp, hp.gimg, hp.gflow, 1, True)]
names = ["corr_fwd", "corr_bwd", "res_fwd", "cn_fwd", "cn_bwd", "fill", "res_bwd"]
evs = [torch.cuda.Event() for _ in range(8)]
tiny = torch.zeros(1, device=dev)


def with_between(between, after=None):
    def f():
        for i, op in enumerate(ops):
            op()
            if after is None or i in after:
                between(i)
    return f


print("event record after every launch        %.4f ms" % clock(with_between(lambda i: evs[i].record())), flush=True)
print("4-byte fill after every launch         %.4f ms" % clock(with_between(lambda i: tiny.zero_())), flush=True)
for k in range(len(ops) - 1):
    print("event record after %-9s only      %.4f ms" % (names[k], clock(with_between(lambda i: evs[i].record(), after={k}))), flush=True)
print("host sync after the correlation kernels %.4f ms" % clock(with_between(lambda i: torch.cuda.synchronize(), after={1})), flush=True)
# other orders of the same launches
for order in ([2, 3, 4, 5, 6, 0, 1], [0, 2, 1, 3, 4, 5, 6], [0, 2, 3, 1, 4, 5, 6], [5, 0, 1, 2, 3, 4, 6]):
    def f(order=order):
        for i in order:
            ops[i]()
    print("order %s  %.4f ms" % ([names[i] for i in order], clock(f)), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    hp.step_plain()
print("hipGraph replay                        %.4f ms" % clock(g.replay), flush=True)
for n, t in list(vars(hp).items()):
    if torch.is_tensor(t) and t.numel():
        setattr(hp, n, t.clone())
print("re-placed buffers                      %.4f ms" % clock(hp.step_plain), flush=True)
ev = {}
for _ in range(20):
    hp.step(ev)
torch.cuda.synchronize()
print("per kernel (event pairs):", {k: round(sum(s.elapsed_time(e) for s, e in v) / len(v) * 1e3, 1) for k, v in ev.items()})
