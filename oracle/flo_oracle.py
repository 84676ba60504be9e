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


# ---------------------------------------------------------------------------------------------------------------------
# Middlebury colour coding of a flow field (utils/flow_utils.py:72-204), restated pixel by pixel.  Pinned against images
# produced by the reference's own flow2img: tests/golden/flowvis_*.npy (tests/golden/make_golden_flowvis.py).
def wheel_oracle():
    """make_color_wheel :157-204: 55 rows (RY 15, YG 6, GC 4, CB 11, BM 13, MR 6) of (R, G, B) in 0..255."""
    rows = []
    for n, fixed, ramp, rising in ((15, 0, 1, True), (6, 1, 0, False), (4, 1, 2, True), (11, 2, 1, False), (13, 2, 0, True), (6, 0, 2, False)):
        for i in range(n):
            rgb = [0.0, 0.0, 0.0]
            rgb[fixed] = 255.0
            r = np.floor(255 * i / n)
            rgb[ramp] = r if rising else 255 - r
            rows.append(rgb)
    return np.array(rows)


def flow2img_oracle(flow):
    """flow2img :72-109 + compute_color :112-154 for one (H, W, 2) array, without the in-place edits of the caller's array."""
    flow = np.array(flow, copy=True)
    u, v = flow[:, :, 0], flow[:, :, 1]
    unknown = (abs(u) > 1e7) | (abs(v) > 1e7)                                            # :81-84
    u[unknown] = 0
    v[unknown] = 0
    with np.errstate(all="ignore"):
        maxrad = max(-1, np.max(np.sqrt(u ** 2 + v ** 2)))                                # :96-97
        un = u / maxrad + np.finfo(float).eps                                             # :98-99 (float64 from here on with NumPy >= 2)
        vn = v / maxrad + np.finfo(float).eps
    wheel = wheel_oracle()
    ncols = wheel.shape[0]
    h, w = un.shape
    img = np.zeros((h, w, 3), np.uint8)
    for y in range(h):
        for x in range(w):
            a, b = un[y, x], vn[y, x]
            nan = bool(np.isnan(a) or np.isnan(b))                                       # :123-124
            if nan:
                a = b = np.float64(0)
            rad = np.sqrt(a ** 2 + b ** 2)                                               # :129
            fk = (np.arctan2(-b, -a) / np.pi + 1) / 2 * (ncols - 1) + 1                  # :131-133
            k0 = int(np.floor(fk))
            k1 = 1 if k0 + 1 == ncols + 1 else k0 + 1                                    # :137-138
            f = fk - k0
            for c in range(3):
                col = (1 - f) * (wheel[k0 - 1, c] / 255) + f * (wheel[k1 - 1, c] / 255)    # :141-145
                col = 1 - rad * (1 - col) if rad <= 1 else col * 0.75                     # :147-151
                img[y, x, c] = np.uint8(np.floor(255 * col * (1 - nan)))                  # :152
            if unknown[y, x]:
                img[y, x] = 0                                                            # :105-106
    return img
