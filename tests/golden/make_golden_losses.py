"""Generates tests/golden/multiscale_*.npz from the REFERENCE's own losses.py.

Run in the dev container only (needs /root/reference):
    python tests/golden/make_golden_losses.py

losses.py is plain PyTorch and imports on the CPU.  For seeded predictions / targets the fixture stores what the reference's
MultiScale(args, norm='L1' | 'L2') (losses.py:52-86) returns for the training tuple -- [lossvalue, epevalue] -- the gradient of
lossvalue with respect to every prediction (autograd through the reference module), and the values of the single-tensor branch
(:80-83), L1Loss (:28-38) and L2Loss (:40-50).  tests/test_losses_pin.py checks the oracle against the fixture on the CPU and
tests/test_gpu_parity.py::test_multiscale_golden checks the fused HIP loss against it on the GPU, where /root/reference is absent.
"""
import importlib.util
import os

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/losses.py"

CASES = [("2x64x128", 2, 64, 128, 11), ("1x128x192", 1, 128, 192, 12), ("3x64x64_zero_diff", 3, 64, 64, 13)]


def load_reference_losses():
    spec = importlib.util.spec_from_file_location("reference_losses", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_inputs(B, H, W, seed, zero_diff=False):
    g = torch.Generator().manual_seed(seed)
    target = torch.randn(B, 2, H, W, generator=g) * 5.0
    outs = [torch.randn(B, 2, H // (4 << i), W // (4 << i), generator=g) * 0.3 for i in range(5)]
    if zero_diff:   # predictions that equal the pooled target exactly on a few pixels: sign(0) = 0 and the 2-norm's zero gradient
        for i in range(5):
            ti = torch.nn.functional.avg_pool2d(0.05 * target, 4 << i, 4 << i)
            outs[i][:, :, ::2, ::2] = ti[:, :, ::2, ::2]
    return target, outs


def main():
    ref = load_reference_losses()
    for name, B, H, W, seed in CASES:
        target, outs = make_inputs(B, H, W, seed, "zero" in name)
        d = dict(target=target.numpy())
        for i, o in enumerate(outs):
            d[f"out{i}"] = o.numpy()
        for norm in ("L1", "L2"):
            crit = ref.MultiScale(None, norm=norm)
            leaves = [o.clone().requires_grad_(True) for o in outs]
            loss, epe = crit(tuple(leaves), target)
            loss.backward()
            d[f"loss_{norm}"] = np.float32(loss.item())
            d[f"epe_{norm}"] = np.float32(epe.item())
            for i, leaf in enumerate(leaves):
                d[f"grad_{norm}_{i}"] = leaf.grad.numpy()
            full = torch.nn.functional.interpolate(outs[0], size=(H, W), mode="nearest")
            one = crit(full, target)
            d[f"single_{norm}"] = np.array([one[0].item(), one[1].item()], np.float32)
        full = torch.nn.functional.interpolate(outs[0], size=(H, W), mode="nearest")
        d["l1loss"] = np.array([v.item() for v in ref.L1Loss(None)(full, target)], np.float32)
        d["l2loss"] = np.array([v.item() for v in ref.L2Loss(None)(full, target)], np.float32)
        d["weights"] = ref.MultiScale(None).loss_weights.numpy()
        np.savez_compressed(os.path.join(OUT, f"multiscale_{name}.npz"), **d)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in d.items() if k.startswith(("loss", "epe"))})


if __name__ == "__main__":
    main()
