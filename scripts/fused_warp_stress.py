import sys, torch
sys.path.insert(0, "flownet2-pytorch_amd")
from networks.resample2d_package.resample2d import Resample2d, WarpDiffNormCat, WarpDiffNorm
from networks.channelnorm_package.channelnorm import ChannelNorm
dev = torch.device("cuda:0")
bad = 0
for shape in ((2, 128, 192), (8, 384, 512), (2, 96, 160)):
    B, H, W = shape
    for seed in range(6):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, 6, H, W, generator=g).to(dev)
        f0 = (torch.randn(B, 2, H, W, generator=g) * (0.5 if seed % 2 else 6.0)).to(dev)
        if seed % 3 == 0: f0[:, 0] += 17.0
        gcat = torch.randn(B, 12, H, W, generator=g).to(dev)
        gn = torch.randn(B, 1, H, W, generator=g).to(dev)
        f = f0.clone().requires_grad_(True)
        res = Resample2d()(x[:, 3:], f)
        unf = torch.cat((x, res, f / 20.0, ChannelNorm()(x[:, :3] - res)), dim=1)
        unf.backward(gcat)
        fu = f0.clone().requires_grad_(True)
        une = ChannelNorm()(x[:, :3] - Resample2d()(x[:, 3:], fu)); une.backward(gn)
        for rep in range(40):
            f2 = f0.clone().requires_grad_(True)
            out = WarpDiffNormCat(20.0)(x, f2); out.backward(gcat)
            f3 = f0.clone().requires_grad_(True)
            o3 = WarpDiffNorm()(x, f3); o3.backward(gn)
            # some unrelated allocations / kernels in between, as a network would have
            tmp = torch.randn(1 + (rep * 7919) % 50000, device=dev).sum()
            if not (torch.equal(out, unf.detach()) and torch.equal(f2.grad, f.grad) and torch.equal(o3, une.detach()) and torch.equal(f3.grad, fu.grad)):
                bad += 1
                print("MISMATCH", shape, seed, rep, float((f2.grad - f.grad).abs().max()), float((f3.grad - fu.grad).abs().max()), float((out - unf.detach()).abs().max()))
print("stress done, mismatches:", bad)
