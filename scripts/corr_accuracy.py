"""Accuracy of the correlation-forward kernels against an fp64 reference (torch, on the GPU).
Usage: python scripts/corr_accuracy.py --algos 1,2,2000"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import torch.nn.functional as F
import fn2_capi

ap = argparse.ArgumentParser()
ap.add_argument("--algos", default="1,2")
ap.add_argument("--shape", default="8,256,48,64")
a = ap.parse_args()
B, C, H, W = (int(v) for v in a.shape.split(","))
dev = torch.device("cuda:0")


def ref64(in1, in2, md=20, s2=2):
    a64, b64 = in1.double(), in2.double()
    p2 = F.pad(b64, (md, md, md, md))
    dr = md // s2
    outs = []
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            y0, x0 = md + tj * s2, md + ti * s2
            outs.append((a64 * p2[:, :, y0:y0 + H, x0:x0 + W]).mean(1, keepdim=True))
    return torch.cat(outs, 1)


g = torch.Generator().manual_seed(0)
cases = {
    "N(0,1)": (torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)),
    "leaky_relu(N(0,1))": (F.leaky_relu(torch.randn(B, C, H, W, generator=g), 0.1), F.leaky_relu(torch.randn(B, C, H, W, generator=g), 0.1)),
    "N(0,1)*100": (100 * torch.randn(B, C, H, W, generator=g), 100 * torch.randn(B, C, H, W, generator=g)),
    "N(0,1)*1e-3": (1e-3 * torch.randn(B, C, H, W, generator=g), 1e-3 * torch.randn(B, C, H, W, generator=g)),
    "N(0,1)*1e-6 x N(0,1)*1e-8": (1e-6 * torch.randn(B, C, H, W, generator=g), 1e-8 * torch.randn(B, C, H, W, generator=g)),
    "lognormal magnitudes": (torch.randn(B, C, H, W, generator=g) * torch.exp(3 * torch.randn(B, C, H, W, generator=g)),
                             torch.randn(B, C, H, W, generator=g) * torch.exp(3 * torch.randn(B, C, H, W, generator=g))),
}
for name, (x1, x2) in cases.items():
    x1, x2 = x1.to(dev), x2.to(dev)
    r = ref64(x1, x2)
    scale = float(r.abs().max())
    print(f"{name}: max|ref| = {scale:.4g}")
    for algo in (int(v) for v in a.algos.split(",")):
        out = fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, algo=algo)
        err = (out.double() - r).abs()
        print(f"   algo {algo:5d}: max abs err {float(err.max()):.3e}  (rel to max|ref| {float(err.max()) / scale:.2e})  rms err {float(err.pow(2).mean().sqrt()):.3e}")
