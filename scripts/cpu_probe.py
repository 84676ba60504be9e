"""Which vectorised-PyTorch formulation / thread count is fastest for the correlation on this host (bench.py's cpu_baseline_fast)."""
import os, sys, time
import torch
B, C, H, W = 2, 256, 48, 64
g = torch.Generator().manual_seed(0)
a = torch.randn(B, C, H, W, generator=g); b = torch.randn(B, C, H, W, generator=g)
bp = torch.nn.functional.pad(b, (20, 20, 20, 20))
def f_mulsum():
    out = a.new_empty(B, 441, H, W)
    for tj in range(21):
        for ti in range(21):
            out[:, tj * 21 + ti] = (a * bp[:, :, 2 * tj:2 * tj + H, 2 * ti:2 * ti + W]).sum(1)
    return out / C
def f_rows():   # one displacement row at a time: a (B,C,H,1,W) * unfolded in2 rows (B,C,H,21,W)
    out = a.new_empty(B, 441, H, W)
    win = bp.unfold(3, W, 2)                      # B, C, Hp, 21, W   (view)
    for tj in range(21):
        out[:, tj * 21:(tj + 1) * 21] = torch.einsum("bchw,bchtw->bthw", a, win[:, :, 2 * tj:2 * tj + H])
    return out / C
def f_bmm():    # per (n, y, tj): (W x C) @ (C x Wp), then the 21 diagonals x' = x + 2 ti
    out = a.new_empty(B, 441, H, W)
    At = a.permute(0, 2, 3, 1).reshape(B * H, W, C)                       # (n,y) x W x C
    idx = (torch.arange(W).view(1, W) + 2 * torch.arange(21).view(21, 1))   # 21 x W: column of the padded row
    for tj in range(21):
        Brow = bp[:, :, 2 * tj:2 * tj + H].permute(0, 2, 1, 3).reshape(B * H, C, W + 40)
        M = torch.bmm(At, Brow)                                              # (n,y) x W x Wp
        d = M.gather(2, idx.t().unsqueeze(0).expand(B * H, W, 21))           # (n,y) x W x 21
        out[:, tj * 21:(tj + 1) * 21] = d.view(B, H, W, 21).permute(0, 3, 1, 2)
    return out / C
ref = None
for nt in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32,64").split(",")]:
    torch.set_num_threads(nt)
    for name, fn in (("mul+sum", f_mulsum), ("unfold einsum rows", f_rows), ("bmm + diagonals", f_bmm)):
        t0 = time.perf_counter(); o = fn(); t1 = time.perf_counter(); o = fn(); t2 = time.perf_counter()
        if ref is None: ref = o
        print("threads %3d  %-20s first %.3f s  second %.3f s  (B=%d)  max |d| %.1e" % (nt, name, t1 - t0, t2 - t1, B, float((o - ref).abs().max())), flush=True)
        if t2 - t1 > 20: break
