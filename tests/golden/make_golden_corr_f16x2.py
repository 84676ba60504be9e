"""Generates tests/golden/corr_f16x2_*.npz from the REFERENCE's own kernels (oracle/_ref/libfn2_ref.so, dev container only) at a
shape FN2_CORR_AUTO sends to the graded matrix-core kernels (FlowNetC's parameters, C % 64 == 0, H even, W % 8 == 0): the f16x2
forward and backward pinned DIRECTLY to the reference's device code, not only through the restated oracle.  A seed of its own.
    python tests/golden/make_golden_corr_f16x2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = Oracle(ref=True)
    rng = np.random.default_rng(20260923)
    for (name, B, C, H, W) in [("f16x2_c64_4x8", 1, 64, 4, 8)]:
        pad, k, md, s1, s2 = 20, 1, 20, 1, 2        # FlowNetC.py:28
        in1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
        in2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
        in1[0, 3] *= 30.0; in2[0, 7] *= 1e-3         # channel magnitudes apart: the backward's per-channel scales
        out = ref.corr_fwd(in1, in2, pad, k, md, s1, s2)
        gout = rng.standard_normal(out.shape).astype(np.float32)
        g1, g2 = ref.corr_bwd(in1, in2, gout, pad, k, md, s1, s2)
        np.savez_compressed(os.path.join(OUT, f"corr_{name}.npz"), in1=in1, in2=in2, out=out, gout=gout, g1=g1, g2=g2,
                            params=np.array([pad, k, md, s1, s2], np.int32))
        print("corr", name, out.shape, os.path.getsize(os.path.join(OUT, f"corr_{name}.npz")), "bytes")


if __name__ == "__main__":
    main()
