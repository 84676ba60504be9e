"""Middlebury ``.flo`` files: the inference output format of the reference.

Same names, arguments and error behaviour as the reference's ``utils/flow_utils.py`` (``readFlow`` :7-26, ``writeFlow``
:28-57) -- byte work, so the files are BIT-IDENTICAL to the reference's (tests/test_flo.py against fixtures written by the
reference's own functions):

    bytes 0..3    float32 202021.25 (the tag 'PIEH'), little endian
    bytes 4..7    int32 width, bytes 8..11 int32 height
    then          height x width x (u, v) float32, row-major, u and v interleaved

``writeFlow`` forms the interleaved float32 image in one pass instead of a float64 scratch matrix filled by two strided
assignments (:51-54; the detour through float64 is exact for float32 and float64 inputs alike, so the bytes agree).
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
