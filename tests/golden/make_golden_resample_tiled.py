"""Generates tests/golden/resample_tiled_*.npz from the REFERENCE's own kernels (oracle/_ref/libfn2_ref.so = the reference's
resample2d_kernel.cu compiled against the CPU SIMT shim; dev container only: `make -C oracle ref`), at shapes the TILED HIP
kernels take (C = 3, W % 4 == 0, H >= 16, W >= 32): ragged tiles, a translation of tens of pixels under the flow (the
backward windows follow the flow; clamped runs pile onto border cells), noise and a few far outliers.  A seed of its own,
so that tests/golden/make_golden.py's fixtures do not move.
    python tests/golden/make_golden_resample_tiled.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = Oracle(ref=True)
    rng = np.random.default_rng(20260922)
    for (name, B, C, H, W, sigma, shift) in [("tiled_translation", 1, 3, 40, 96, 2.0, (25.0, -18.0)),
                                             ("tiled_noise", 1, 3, 33, 68, 4.0, (0.0, 0.0))]:
        img = rng.uniform(-0.5, 0.5, (B, C, H, W)).astype(np.float32)
        flow = (rng.standard_normal((B, 2, H, W)) * sigma).astype(np.float32)
        flow[:, 0] += np.float32(shift[0]); flow[:, 1] += np.float32(shift[1])
        flow.reshape(-1)[rng.integers(0, flow.size, 12)] *= 20.0   # far outliers: border clamps, out-of-window corners
        flow[0, :, 0, 0] = 0.0                                      # exact-integer coordinates
        gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
        d = dict(img=img, flow=flow, gout=gout)
        for bil in (1, 0):
            d[f"out_bil{bil}"] = ref.resample_fwd(img, flow, 1, bool(bil))
        gimg, gflow = ref.resample_bwd(img, flow, gout, 1, True)
        d.update(gimg=gimg, gflow=gflow)
        np.savez_compressed(os.path.join(OUT, f"resample_{name}.npz"), **d)
        print("resample", name, os.path.getsize(os.path.join(OUT, f"resample_{name}.npz")), "bytes")


if __name__ == "__main__":
    main()
