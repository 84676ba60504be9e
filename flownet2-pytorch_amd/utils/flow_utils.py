"""Middlebury ``.flo`` files: the inference output format of the reference.

Same names, arguments and error behaviour as the reference's ``utils/flow_utils.py`` (``readFlow`` :7-26, ``writeFlow``
:28-57) -- byte work, so the files are BIT-IDENTICAL to the reference's (tests/test_flo.py against fixtures written by the
reference's own functions):

    bytes 0..3    float32 202021.25 (the tag 'PIEH'), little endian
    bytes 4..7    int32 width, bytes 8..11 int32 height
    then          height x width x (u, v) float32, row-major, u and v interleaved

``writeFlow`` forms the interleaved float32 image in one pass instead of a float64 scratch matrix filled by two strided
assignments (:51-54; the detour through float64 is exact for float32 and float64 inputs alike, so the bytes agree).
``flow2img`` / ``compute_color`` / ``make_color_wheel`` / ``visulize_flow_file`` (the reference's spelling) are its Middlebury
colour coding of a flow field (:62-204), value for value (tests/test_flo.py against images made by the reference's functions).
``save_flows`` is what the reference's inference loop does per batch (main.py:385-389: one ``.data.cpu().numpy()``
+ transpose + writeFlow per item): the batch is interleaved on the device and crosses PCIe once.
"""
import os
import struct

import numpy as np

TAG_FLOAT = 202021.25
_HEADER = struct.Struct("<fii")


def readFlow(fn):
    """Reads a .flo file -> float32 array (height, width, 2); ``None`` (and a message) for a wrong magic number, as the
    reference does (:17-19)."""
    with open(fn, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) < 4 or struct.unpack("<f", head[:4])[0] != TAG_FLOAT:
            print("Magic number incorrect. Invalid .flo file")
            return None
        if len(head) < _HEADER.size:          # the tag is right but width / height are cut off
            print("Truncated header. Invalid .flo file")
            return None
        _, w, h = _HEADER.unpack(head)
        data = np.fromfile(f, np.float32, count=2 * int(w) * int(h))
    return np.resize(data, (int(h), int(w), 2))      # np.resize like the reference: a short file repeats its data


def writeFlow(filename, uv, v=None):
    """Writes optical flow to ``filename``.  ``uv``: (H, W, 2) with v None, or u of shape (H, W) next to ``v``."""
    if v is None:
        assert uv.ndim == 3
        assert uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    assert u.shape == v.shape
    height, width = u.shape
    out = np.empty((height, width, 2), dtype=np.float32)
    out[:, :, 0] = u        # same value conversion as the reference's float64 matrix + astype(float32)
    out[:, :, 1] = v
    with open(filename, "wb") as f:
        f.write(_HEADER.pack(TAG_FLOAT, width, height))
        out.tofile(f)


def save_flows(folder, flows, start_index=0, pattern="%06d.flo"):
    """Writes a batch of flows (torch tensor or array, B x 2 x H x W, any device) as ``folder/000000.flo`` ... -- the files
    main.py:385-389 produces item by item.  The (u, v) interleave happens where the tensor lives (one permuted copy on the
    GPU), then ONE device-to-host transfer; returns the paths."""
    import torch
    if isinstance(flows, torch.Tensor):
        assert flows.dim() == 4 and flows.shape[1] == 2, "flows: B x 2 x H x W"
        host = flows.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous().cpu().numpy()
    else:
        flows = np.asarray(flows)
        assert flows.ndim == 4 and flows.shape[1] == 2, "flows: B x 2 x H x W"
        host = np.ascontiguousarray(flows.transpose(0, 2, 3, 1), dtype=np.float32)
    os.makedirs(folder, exist_ok=True)
    paths = []
    b, h, w, _ = host.shape
    head = _HEADER.pack(TAG_FLOAT, w, h)
    for i in range(b):
        path = os.path.join(folder, pattern % (start_index + i))
        with open(path, "wb") as f:
            f.write(head)
            host[i].tofile(f)
        paths.append(path)
    return paths


# ---------------------------------------------------------------------------------------------------------------------
# Middlebury colour coding (reference utils/flow_utils.py:62-204): direction -> hue on a 55-entry colour wheel, magnitude
# (relative to the largest in the image) -> saturation.  Whole-array numpy, all three channels at once; the arithmetic and
# its order are the reference's, so the uint8 images are identical.
_WHEEL_SEGMENTS = (          # (entries, channel held at 255, channel that ramps, ramp rises?)   RY YG GC CB BM MR  (:162-167)
    (15, 0, 1, True), (6, 1, 0, False), (4, 1, 2, True), (11, 2, 1, False), (13, 2, 0, True), (6, 0, 2, False))


def make_color_wheel():
    """55 x 3 float64 table of (R, G, B) in 0..255 (:157-204)."""
    wheel = np.zeros((sum(seg[0] for seg in _WHEEL_SEGMENTS), 3))
    row = 0
    for n, fixed, ramp, rising in _WHEEL_SEGMENTS:
        steps = np.floor(255 * np.arange(0, n) / n)
        wheel[row:row + n, fixed] = 255
        wheel[row:row + n, ramp] = steps if rising else 255 - steps
        row += n
    return wheel


def compute_color(u, v):
    """Colour image (H, W, 3; float64 holding uint8 values, like the reference's) of the normalised flow components u, v
    (:112-154).  Unlike the reference the arguments are not edited."""
    nan = np.isnan(u) | np.isnan(v)
    u = np.where(nan, 0, u)
    v = np.where(nan, 0, v)
    wheel = make_color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1) + 1          # 1 .. ncols: position on the wheel
    k0 = np.floor(fk).astype(int)
    k1 = np.where(k0 + 1 == ncols + 1, 1, k0 + 1)
    f = (fk - k0)[..., None]
    col = (1 - f) * (wheel[k0 - 1] / 255) + f * (wheel[k1 - 1] / 255)    # (H, W, 3)
    inside = (rad <= 1)[..., None]
    col = np.where(inside, 1 - rad[..., None] * (1 - col), col * 0.75)
    return np.uint8(np.floor(255 * col * (1 - nan)[..., None])).astype(np.float64)


def flow2img(flow_data):
    """(H, W, 2) flow -> (H, W, 3) uint8 colour image (:72-109).  Components beyond 1e7 in magnitude mark unknown flow
    (black).  The reference zeroes those entries in the caller's array; this function leaves ``flow_data`` untouched."""
    u = np.array(flow_data[:, :, 0], copy=True)
    v = np.array(flow_data[:, :, 1], copy=True)
    unknown = (abs(u) > 1e7) | (abs(v) > 1e7)
    u[unknown] = 0
    v[unknown] = 0
    with np.errstate(all="ignore"):                      # an all-zero flow divides 0 by 0 (nan -> black), as in the reference
        maxrad = max(-1, np.max(np.sqrt(u ** 2 + v ** 2)))
        eps = np.finfo(float).eps                        # a float64 scalar: float32 flows continue in float64 from here on
        img = compute_color(u / maxrad + eps, v / maxrad + eps)
    img[unknown] = 0
    return np.uint8(img)


def visulize_flow_file(flow_filename, save_dir=None):
    """Colour image of a .flo file; with ``save_dir`` also written there as ``<name>-vis.png`` (:62-70, the reference's spelling
    of the name).  Returns the image."""
    img = flow2img(readFlow(flow_filename))
    if save_dir:
        from PIL import Image
        base = os.path.basename(flow_filename)
        Image.fromarray(img).save(os.path.join(save_dir, "%s-vis.png" % base[:-4]))
    return img
