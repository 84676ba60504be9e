"""Generates tests/golden/fullsize_*.npz from the REFERENCE's own kernels (oracle/_ref/libfn2_ref.so, dev container only) at the
BASELINE shape of Resample2d and ChannelNorm: 8 x 3 x 384 x 512 with the SURVEY's flow (N(0, 4^2) px, 1 % of the entries x 20) --
resample2d_kernel.cu / channelnorm_kernel.cu under the CPU SIMT shim, a few seconds per kernel.
    python tests/golden/make_golden_full_size.py

The inputs are NOT stored: first draws of numpy's default_rng(SEED) (order: img, flow, the outlier indices, gout, gnorm), with a
checksum so that a test can tell a different generator from a wrong kernel.  Of every result the fixture keeps ROWS (8 image rows
of every plane, in full) and the float64 sum and sum of absolute values of every (batch, channel) plane.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260925
B, C, H, W = 8, 3, 384, 512
ROWS = [0, 16, 31, 32, 191, 256, 352, 383]


def make_inputs():
    rng = np.random.default_rng(SEED)
    img = rng.uniform(-0.5, 0.5, (B, C, H, W)).astype(np.float32)
    flow = (rng.standard_normal((B, 2, H, W)) * 4.0).astype(np.float32)
    flow.reshape(-1)[rng.integers(0, flow.size, flow.size // 100)] *= np.float32(20.0)
    gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
    gnorm = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    img[0, :, 5, 7] = 0.0                       # an exactly-zero pixel: ChannelNorm's backward gives 0 there, not nan
    return img, flow, gout, gnorm


def checksum(*arrays):
    return np.array([float(np.sum(a.astype(np.float64) * np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape) % 7.0)) for a in arrays])


def keep(d, name, a):
    a64 = a.astype(np.float64)
    d[name + "_rows"] = a[:, :, ROWS]
    d[name + "_sum"] = a64.sum(axis=(2, 3))
    d[name + "_abs"] = np.abs(a64).sum(axis=(2, 3))


def main():
    ref = Oracle(ref=True)
    img, flow, gout, gnorm = make_inputs()
    d = dict(seed=np.int64(SEED), shape=np.array([B, C, H, W], np.int32), rows=np.array(ROWS, np.int32),
             input_checksum=checksum(img, flow, gout, gnorm))
    keep(d, "warp", ref.resample_fwd(img, flow, 1, True))
    keep(d, "warp_nearest", ref.resample_fwd(img, flow, 1, False))
    gi, gf = ref.resample_bwd(img, flow, gout, 1, True)
    keep(d, "gimg", gi)
    keep(d, "gflow", gf)
    n = ref.chnorm_fwd(img)
    keep(d, "norm", n)
    keep(d, "gnorm_in", ref.chnorm_bwd(img, n, gnorm))
    path = os.path.join(OUT, "fullsize_resample_chnorm_8x3x384x512.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
