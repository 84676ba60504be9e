"""Generates tests/golden/flowvis_*.npy (+ flowvis_inputs.npz) with the REFERENCE's own flow2img / make_color_wheel
(/root/reference/utils/flow_utils.py:72-204), imported from where it lies.  Dev container only:
    python tests/golden/make_golden_flowvis.py
The fixtures pin oracle/flo_oracle.py (restatement) and flownet2-pytorch_amd/utils/flow_utils.py (product) value for value."""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
from utils import flow_utils as ref  # noqa: E402


def main():
    rng = np.random.default_rng(20260923)
    cases = {}
    cases["noise_24x32"] = (rng.standard_normal((24, 32, 2)) * 4).astype(np.float32)
    yy, xx = np.mgrid[-1:1:33j, -1:1:41j]
    cases["radial_33x41"] = np.stack((xx * 30, yy * 30), -1).astype(np.float32)             # every direction, radius 0 .. max
    a = (rng.standard_normal((9, 11, 2)) * 1e-3).astype(np.float32)
    a[0, 0] = (2e7, 0.0); a[3, 4] = (0.0, -3e9); a[5, 5] = (np.nan, 1.0); a[8, 10] = (0.0, 0.0)   # unknown flow, nan, exact zero
    cases["specials_9x11"] = a
    cases["f64_6x7"] = rng.standard_normal((6, 7, 2)) * 100                                 # float64 input
    cases["axis_1x8"] = np.array([[[1, 0], [-1, 0], [0, 1], [0, -1], [1, 1], [-1, -1], [1, -1], [-1, 1]]], np.float32)
    for k, v in cases.items():
        with np.errstate(all="ignore"):
            np.save(os.path.join(OUT, "flowvis_%s.npy" % k), ref.flow2img(np.array(v, copy=True)))    # the reference edits its argument
    np.save(os.path.join(OUT, "flowvis_wheel.npy"), ref.make_color_wheel())
    np.savez(os.path.join(OUT, "flowvis_inputs.npz"), **cases)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("flowvis_")))


if __name__ == "__main__":
    main()
