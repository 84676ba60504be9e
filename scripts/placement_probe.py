"""Does the step time of bench.py's HotPath depend on WHERE its buffers lie?  (Round 6 observation: on a box of the slow class the raw
step took 0.249 ms while the same kernels through the autograd modules -- freshly allocated outputs -- took 0.184 ms.)  Times the raw step
for several placements of the same tensors on one box: as allocated, after cloning every tensor, behind spacers of various sizes, with the
outputs freshly allocated each step.  Prints one line per placement with the tensors' addresses modulo 2 MiB / 1 GiB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-pytorch_amd")]
import torch
import bench

dev = torch.device("cuda:0")


def clock(hp, n=200):
    for _ in range(50):
        hp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        hp.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def per_kernel(hp, n=40):
    ev = {}
    for _ in range(n):
        hp.step(ev)
    torch.cuda.synchronize()
    return {k: round(sum(s.elapsed_time(e) for s, e in v) / len(v) * 1e3, 1) for k, v in ev.items()}


def addrs(hp):
    names = ["in1", "in2", "gcorr", "out", "g1", "g2", "img", "flow", "warped", "gimg", "gflow", "norm", "gdiff", "gwarp", "gnorm"]
    return " ".join(f"{n}:{getattr(hp, n).data_ptr() >> 21 & 0x1ff:03x}" for n in names if getattr(hp, n).numel())


hp = bench.HotPath(dev, 1234)
print("as allocated      %.4f ms  %s" % (clock(hp), per_kernel(hp)), flush=True)
print("   2 MiB-page index mod 512:", addrs(hp))
keep = []
for trial in range(3):
    for n, t in list(vars(hp).items()):
        if torch.is_tensor(t) and t.numel():
            keep.append(t)                      # the old block stays allocated: the clone must land elsewhere
            setattr(hp, n, t.clone())
    print("cloned (%d)        %.4f ms  %s" % (trial + 1, clock(hp), per_kernel(hp)), flush=True)
    print("   2 MiB-page index mod 512:", addrs(hp))
del keep
torch.cuda.empty_cache()
for mb in (1, 7, 64, 333, 1024):
    spacer = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    hp2 = bench.HotPath(dev, 1234)
    print("behind %4d MiB    %.4f ms  %s" % (mb, clock(hp2), per_kernel(hp2)), flush=True)
    del hp2, spacer
    torch.cuda.empty_cache()
# outputs freshly allocated every step (what the autograd modules do)
hp3 = bench.HotPath(dev, 1234)
def fresh_step():
    e = hp3.in1.new_empty
    hp3.out, hp3.g1, hp3.g2 = e(0), e(0), e(0)
    hp3.warped, hp3.gimg, hp3.gflow = torch.empty_like(hp3.img), torch.empty_like(hp3.img), torch.empty_like(hp3.flow)
    hp3.norm, hp3.gdiff = torch.empty_like(hp3.gnorm), torch.empty_like(hp3.img)
    bench.HotPath.step(hp3)
for _ in range(50):
    fresh_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    fresh_step()
torch.cuda.synchronize()
print("fresh outputs     %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
