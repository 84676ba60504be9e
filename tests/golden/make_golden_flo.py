"""Generates tests/golden/flo_*.flo (+ flo_inputs.npz) with the REFERENCE's own writer, utils/flow_utils.writeFlow
(/root/reference/utils/flow_utils.py:28-57), imported from where it lies.  Dev container only:
    python tests/golden/make_golden_flo.py
The fixtures pin oracle/flo_oracle.py (restatement) and flownet2-pytorch_amd/utils/flow_utils.py (product) byte for byte."""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
from utils import flow_utils as ref  # noqa: E402


def main():
    rng = np.random.default_rng(20260922)
    cases = {}
    a = (rng.standard_normal((5, 7, 2)) * 30).astype(np.float32)
    a[0, 0] = (np.inf, -np.inf); a[1, 2] = (np.nan, -0.0); a[4, 6] = (1e-45, 3.4028235e38)     # specials, subnormal, max
    cases["f32_5x7"] = a
    cases["f64_3x4"] = rng.standard_normal((3, 4, 2)) * 1e3                                        # float64 input: rounded on write
    cases["chw_view_6x9"] = np.ascontiguousarray(rng.standard_normal((2, 6, 9)).astype(np.float32))   # written through .transpose(1, 2, 0)
    cases["sep_u_4x4"] = rng.standard_normal((4, 4)).astype(np.float32)
    cases["sep_v_4x4"] = rng.standard_normal((4, 4)).astype(np.float32)
    cases["one_1x1"] = np.array([[[1.5, -2.25]]], np.float32)
    ref.writeFlow(os.path.join(OUT, "flo_f32_5x7.flo"), cases["f32_5x7"])
    ref.writeFlow(os.path.join(OUT, "flo_f64_3x4.flo"), cases["f64_3x4"])
    ref.writeFlow(os.path.join(OUT, "flo_chw_view_6x9.flo"), cases["chw_view_6x9"].transpose(1, 2, 0))    # main.py:388
    ref.writeFlow(os.path.join(OUT, "flo_sep_4x4.flo"), cases["sep_u_4x4"], cases["sep_v_4x4"])
    ref.writeFlow(os.path.join(OUT, "flo_one_1x1.flo"), cases["one_1x1"])
    np.savez(os.path.join(OUT, "flo_inputs.npz"), **cases)
    back = ref.readFlow(os.path.join(OUT, "flo_f32_5x7.flo"))
    np.save(os.path.join(OUT, "flo_f32_5x7_readback.npy"), back)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("flo_")))


if __name__ == "__main__":
    main()
