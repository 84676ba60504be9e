"""CPU restatement of the reference's Middlebury .flo writer / reader -- TEST INFRASTRUCTURE ONLY (see oracle.py's header).

Follows utils/flow_utils.py statement by statement:
  writeFlow :28-57  tag TAG_CHAR = float32 202021.25 (:5, :47), int32 width then int32 height (:48-49), a float64 matrix
                    height x 2*width with u in the even and v in the odd columns (:51-53), written as float32 (:54);
  readFlow  :7-26   float32 magic compared with 202021.25 (:15-16), int32 w, int32 h (:20-21), 2*w*h float32 (:23),
                    np.resize to (h, w, 2) (:26).
Pinned against files written by the reference's own functions: tests/golden/flo_*.flo (tests/golden/make_golden_flo.py).
"""
import numpy as np


def flo_bytes(uv, v=None):
    """The exact bytes writeFlow(filename, uv, v) puts into the file."""
    n_bands = 2
    if v is None:
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    assert u.shape == v.shape
    height, width = u.shape
    tmp = np.zeros((height, width * n_bands))                      # float64, as the reference (:51)
    tmp[:, np.arange(width) * 2] = u
    tmp[:, np.arange(width) * 2 + 1] = v
    return (np.array([202021.25], np.float32).tobytes() + np.array(width).astype(np.int32).tobytes() +
            np.array(height).astype(np.int32).tobytes() + tmp.astype(np.float32).tobytes())


def flo_parse(buf):
    """readFlow on a byte string: (h, w, 2) float32, or None for a wrong magic number."""
    magic = np.frombuffer(buf, np.float32, count=1)
    if 202021.25 != magic:
        return None
    w = int(np.frombuffer(buf, np.int32, count=1, offset=4)[0])
    h = int(np.frombuffer(buf, np.int32, count=1, offset=8)[0])
    n = min(2 * w * h, (len(buf) - 12) // 4)
    return np.resize(np.frombuffer(buf, np.float32, count=n, offset=12), (h, w, 2))
