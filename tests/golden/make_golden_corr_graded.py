"""Generates tests/golden/corrgraded_1x64x48x64.npz from the REFERENCE's own kernels (oracle/_ref/libfn2_ref.so, dev container only)
at the GRADED geometry: FlowNetC's parameters on a 48 x 64 map -- the real task table of the f16x2 correlation kernels (every (row
group, neighbour row block) pair, ragged displacement ranges at all four borders), one batch item, 64 channels.  VERDICT r4 next #7.

The reference's device code runs under the CPU SIMT shim (one fibre per CUDA thread): about four minutes for this case.
    python tests/golden/make_golden_corr_graded.py

To keep the fixture small the inputs are NOT stored: they are the first draws of numpy's default_rng(seed_of(C)) (in1, in2, gout, in
this order, standard normal, float32; two channels of in1 / in2 rescaled as in make_golden_corr_f16x2.py), and the fixture holds a
checksum of them so that a test can tell a different generator from a wrong kernel.  Of the reference's results it keeps
  out    : 32 of the 441 displacement planes in full (PLANES: the four corners, the centre, the band edges, a spread of the rest)
           and the float64 sum and sum of squares of EVERY plane,
  g1, g2 : 8 of the 64 channels in full (CHANNELS, including the two rescaled ones) and the float64 sums of every channel.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260924
# (name suffix, B, C, H, W): 64 channels = the smallest channel count the kernels take (two 32-channel steps, one 64-channel group);
# 256 channels = FlowNetC's conv3 (eight steps of the forward, four channel-group tasks per row group in the backward; ~20 minutes)
# 56 x 128 (round 6): Sintel-size conv3 -- the task tables of the WIDE kernels (column windows, row-group pairs per B row block,
# seven row groups = an odd count, windows that end at the right border)
CASES = [("1x64x48x64", 1, 64, 48, 64), ("1x256x48x64", 1, 256, 48, 64), ("1x64x56x128", 1, 64, 56, 128)]
PLANES = sorted(set([0, 20, 420, 440, 220, 10, 210, 230, 430, 21, 41, 399, 419] + list(range(7, 441, 23))))[:32]
CHANNELS = [0, 3, 7, 17, 31, 32, 48, 63]


def seed_of(C, W=64):
    if W != 64:
        return SEED + 1000 + W
    return SEED if C == 64 else SEED + C      # (the 64-channel fixture was drawn from SEED itself)


def make_inputs(B, C, H, W):
    rng = np.random.default_rng(seed_of(C, W))
    in1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    in2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    gout = rng.standard_normal((B, 441, H, W)).astype(np.float32)
    in1[0, 3] *= 30.0; in2[0, 7] *= 1e-3         # channel magnitudes apart: the backward's per-channel scales
    return in1, in2, gout


def checksum(*arrays):
    return np.array([float(np.sum(a.astype(np.float64) * np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape) % 7.0)) for a in arrays])


def main():
    only = sys.argv[1:]
    for name, B, C, H, W in CASES:
        if not only or name in only:
            one(name, B, C, H, W)


def one(name, B, C, H, W):
    ref = Oracle(ref=True)
    in1, in2, gout = make_inputs(B, C, H, W)
    channels = CHANNELS if C == 64 else [0, 3, 7, 63, 64, 130, 200, 255]
    pad, k, md, s1, s2 = 20, 1, 20, 1, 2        # FlowNetC.py:28
    t = time.time()
    out = ref.corr_fwd(in1, in2, pad, k, md, s1, s2)
    print("forward %.0f s" % (time.time() - t), out.shape, flush=True)
    t = time.time()
    g1, g2 = ref.corr_bwd(in1, in2, gout, pad, k, md, s1, s2)
    print("backward %.0f s" % (time.time() - t), flush=True)
    o64 = out.astype(np.float64)
    d = dict(seed=np.int64(seed_of(C, W)), shape=np.array([B, C, H, W], np.int32), params=np.array([pad, k, md, s1, s2], np.int32),
             input_checksum=checksum(in1, in2, gout), planes=np.array(PLANES, np.int32), channels=np.array(channels, np.int32),
             out_planes=out[:, PLANES], out_sum=o64.sum(axis=(0, 2, 3)), out_sumsq=(o64 * o64).sum(axis=(0, 2, 3)),
             g1_channels=g1[:, channels], g2_channels=g2[:, channels],
             g1_sum=g1.astype(np.float64).sum(axis=(0, 2, 3)), g2_sum=g2.astype(np.float64).sum(axis=(0, 2, 3)),
             g1_abs=np.abs(g1.astype(np.float64)).sum(axis=(0, 2, 3)), g2_abs=np.abs(g2.astype(np.float64)).sum(axis=(0, 2, 3)))
    path = os.path.join(OUT, f"corrgraded_{name}.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
