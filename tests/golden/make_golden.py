"""Generates tests/golden/*.npz from the REFERENCE's own kernels.

Run in the dev container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py

oracle/_ref/libfn2_ref.so is the reference's CUDA device code (the three *_kernel.cu files under
/root/reference/networks/*_package/) compiled against the CPU SIMT shim in oracle/simt/ and launched
with the reference's own grid/block geometry.  Inputs are seeded; every array the reference
produced is stored next to its inputs, so the fixtures can pin both the restated oracle (CPU
tests) and the HIP kernels (GPU tests) on machines where /root/reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (name, B, C, H, W, pad, k, md, s1, s2, dtype)
CORR_CASES = [
    ("flownetc_params_tiny", 1, 32, 6, 8, 20, 1, 20, 1, 2, np.float32),     # FlowNetC.py:28 parameters
    ("flownetc_params_8x8", 2, 16, 8, 8, 20, 1, 20, 1, 2, np.float32),
    ("md4_s2", 2, 40, 6, 8, 4, 1, 4, 1, 2, np.float32),
    ("md3_s1", 1, 33, 5, 7, 3, 1, 3, 1, 1, np.float32),
    ("pad_lt_md", 1, 8, 10, 10, 2, 1, 4, 1, 2, np.float32),
    ("stride1_2_fwd_only", 1, 16, 8, 8, 4, 1, 4, 2, 2, np.float32),
    ("md6_s3_f64", 1, 12, 7, 9, 6, 1, 6, 1, 3, np.float64),
]


def main():
    ref = Oracle(ref=True)
    rng = np.random.default_rng(20260921)
    for (name, B, C, H, W, pad, k, md, s1, s2, dt) in CORR_CASES:
        in1 = rng.standard_normal((B, C, H, W)).astype(dt)
        in2 = rng.standard_normal((B, C, H, W)).astype(dt)
        out = ref.corr_fwd(in1, in2, pad, k, md, s1, s2)
        d = dict(in1=in1, in2=in2, out=out, params=np.array([pad, k, md, s1, s2], np.int32))
        if s1 == 1:  # the reference's backward indexes out of bounds for stride1 != 1
            gout = rng.standard_normal(out.shape).astype(dt)
            g1, g2 = ref.corr_bwd(in1, in2, gout, pad, k, md, s1, s2)
            d.update(gout=gout, g1=g1, g2=g2)
        np.savez_compressed(os.path.join(OUT, f"corr_{name}.npz"), **d)
        print("corr", name, out.shape)

    for (name, B, C, Hi, Wi, H, W, scale) in [("warp_small", 2, 3, 12, 16, 12, 16, 3.0),
                                               ("warp_rect_flow_smaller", 1, 2, 10, 14, 8, 12, 2.0)]:
        img = rng.uniform(-0.5, 0.5, (B, C, Hi, Wi)).astype(np.float32)
        flow = (rng.standard_normal((B, 2, H, W)) * scale).astype(np.float32)
        flow.reshape(-1)[rng.integers(0, flow.size, 6)] *= 20.0  # force border clamps
        flow[0, :, 0, 0] = 0.0                                    # exact-integer coordinates
        flow[0, 0, 1, 1], flow[0, 1, 1, 1] = 2.0, -1.0
        gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
        d = dict(img=img, flow=flow, gout=gout)
        for bil in (1, 0):
            d[f"out_bil{bil}"] = ref.resample_fwd(img, flow, 1, bool(bil))
        gimg, gflow = ref.resample_bwd(img, flow, gout, 1, True)
        d.update(gimg=gimg, gflow=gflow)
        np.savez_compressed(os.path.join(OUT, f"resample_{name}.npz"), **d)
        print("resample", name)

    for (name, B, C, H, W, dt) in [("c3", 2, 3, 9, 12, np.float32), ("c2", 1, 2, 8, 8, np.float32),
                                   ("c5_f64", 1, 5, 6, 7, np.float64)]:
        x = rng.standard_normal((B, C, H, W)).astype(dt)
        x[0, :, 0, 0] = 0.0          # zero pixel: fwd 0, bwd 0 (no NaN)
        x[0, :, 1, 1] = [3.0, 4.0] + [0.0] * (C - 2)  # (3,4) -> 5
        out = ref.chnorm_fwd(x)
        gout = rng.standard_normal(out.shape).astype(dt)
        gin = ref.chnorm_bwd(x, out, gout)
        np.savez_compressed(os.path.join(OUT, f"chnorm_{name}.npz"), x=x, out=out, gout=gout, gin=gin)
        print("chnorm", name)


if __name__ == "__main__":
    main()
