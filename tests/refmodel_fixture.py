"""Shared by tests/golden/make_golden_ref_models.py (dev container, with /root/reference) and tests/test_reference_model_golden.py
(GPU box, without it): the deterministic weights and inputs of the model-level fixture, and the oracle-backed CPU stand-ins of the
three extension modules the reference's wrappers import by bare name.  Test infrastructure; nothing here is product code."""
import sys
import types
import zlib

import numpy as np
import torch

SEED = 20260930
B, H, W = 8, 384, 512          # BASELINE.json: bs 8 @ 384 x 512


def fill_state_dict(model, seed=SEED):
    """Every tensor of model.state_dict() from its NAME and shape alone (numpy generator seeded by (seed, crc32(name))): the reference
    classes here and the harness classes on the GPU box get identical weights whatever order their constructors run in.  He-style
    scale on the fan-in so that activations neither vanish nor explode through 11 convolutions; biases N(0, 0.02)."""
    sd = model.state_dict()
    for name, t in sd.items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if t.dim() == 4:
            transposed = "deconv" in name or "upsampled_flow" in name           # ConvTranspose2d: (in, out, kh, kw), stride 2
            fan = (t.shape[0] * t.shape[2] * t.shape[3] / 4.0) if transposed else (t.shape[1] * t.shape[2] * t.shape[3])
            v = rng.standard_normal(tuple(t.shape)) * np.sqrt(2.0 / fan)
        else:
            v = rng.standard_normal(tuple(t.shape)) * 0.02
        t.copy_(torch.from_numpy(v.astype(np.float32)))
    return model


def make_inputs(seed=SEED, b=B, h=H, w=W):
    """An image pair with structure at several scales (a warp of white noise would turn every rounding difference of a flow into a
    large difference of the warped image) and a true displacement between the frames; inputs B x 3 x 2 x H x W in [0, 255], target
    flow B x 2 x H x W for the training step."""
    rng = np.random.default_rng([seed, 1])
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.zeros((b, 3, 2, h, w))
    target = np.zeros((b, 2, h, w))
    for n in range(b):
        dx, dy = rng.uniform(-6, 6, 2)
        target[n, 0], target[n, 1] = dx, dy
        for c in range(3):
            acc = np.zeros((h, w))
            for _ in range(6):
                fx, fy = rng.uniform(0.01, 0.12, 2)
                acc += rng.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + rng.uniform(0, 2 * np.pi))
            img[n, c, 0] = acc
            img[n, c, 1] = _shift(acc, dx, dy)                                   # the second frame: the first, displaced
    img = (img - img.min()) / (img.max() - img.min()) * 255.0
    img += rng.uniform(-2.0, 2.0, img.shape)                                     # mild sensor noise
    return torch.from_numpy(np.clip(img, 0, 255).astype(np.float32)), torch.from_numpy(target.astype(np.float32))


def _shift(plane, dx, dy):
    h, w = plane.shape
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    xs = np.clip(np.rint(xx - dx).astype(int), 0, w - 1)
    ys = np.clip(np.rint(yy - dy).astype(int), 0, h - 1)
    return plane[ys, xs]


def checksum(t):
    a = t.detach().cpu().numpy().astype(np.float64).reshape(-1)
    return float(np.sum(a * (np.arange(1, a.size + 1, dtype=np.float64) % 7.0)))


def state_checksum(model):
    """Order-independent checksum of a state dict (the harness classes register their modules in another order than the reference's)."""
    sd = model.state_dict()
    return float(sum(checksum(sd[k]) * (1 + zlib.crc32(k.encode()) % 13) for k in sorted(sd)))


def param_digest(t):
    """[sum, sum |.|, L2 norm] in fp64 + the first 16 values of a parameter gradient."""
    a = t.detach().cpu().numpy().astype(np.float64).reshape(-1)
    head = np.zeros(16)
    head[:min(16, a.size)] = a[:16]
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], head])


# ---- oracle-backed stand-ins of correlation_cuda / resample2d_cuda / channelnorm_cuda (CPU tensors, the reference bindings' positional
# signatures and in-place conventions: correlation_cuda.cc:10-167, resample2d_cuda.cc:6-24, channelnorm_cuda.cc:6-25)
def install_oracle_extensions(oracle):
    def np32(t):
        return np.ascontiguousarray(t.detach().numpy(), dtype=np.float32)

    def put(dst, arr):
        dst.resize_(*arr.shape)
        dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)))

    corr = types.ModuleType("correlation_cuda")

    def c_fwd(in1, in2, r1, r2, out, pad, k, md, s1, s2, cm):
        put(out, oracle.corr_fwd(np32(in1), np32(in2), pad, k, md, s1, s2))
        return 1

    def c_bwd(in1, in2, r1, r2, gout, g1, g2, pad, k, md, s1, s2, cm):
        a, b = oracle.corr_bwd(np32(in1), np32(in2), np32(gout), pad, k, md, s1, s2)
        put(g1, a)
        put(g2, b)
        return 1
    corr.forward, corr.backward = c_fwd, c_bwd

    res = types.ModuleType("resample2d_cuda")

    def r_fwd(in1, in2, out, ks, bilinear):
        put(out, oracle.resample_fwd(np32(in1), np32(in2), ks, bool(bilinear)))
        return 1

    def r_bwd(in1, in2, gout, g1, g2, ks, bilinear):
        a, b = oracle.resample_bwd(np32(in1), np32(in2), np32(gout), ks, bool(bilinear))
        g1.add_(torch.from_numpy(a))          # accumulated into (resample2d_kernel.cu:119), arrives zero-filled
        put(g2, b)
        return 1
    res.forward, res.backward = r_fwd, r_bwd

    cn = types.ModuleType("channelnorm_cuda")

    def n_fwd(in1, out, deg):
        put(out, oracle.chnorm_fwd(np32(in1)))
        return 1

    def n_bwd(in1, out, gout, g1, deg):
        put(g1, oracle.chnorm_bwd(np32(in1), np32(out), np32(gout.contiguous())))
        return 1
    cn.forward, cn.backward = n_fwd, n_bwd
    saved = {k: sys.modules.get(k) for k in ("correlation_cuda", "resample2d_cuda", "channelnorm_cuda")}
    sys.modules.update(correlation_cuda=corr, resample2d_cuda=res, channelnorm_cuda=cn)
    return saved
