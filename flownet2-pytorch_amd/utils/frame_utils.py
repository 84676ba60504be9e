"""Frame reader of the data path (reference utils/frame_utils.py:6-20): images as H x W x 3 uint8 arrays (alpha dropped), .flo
flows as float32 H x W x 2.  ``scipy.misc.imread``, which the reference calls, no longer exists; PIL decodes instead."""
from os.path import splitext

import numpy as np

from . import flow_utils


def read_gen(file_name):
    ext = splitext(file_name)[-1]
    if ext in (".png", ".jpeg", ".ppm", ".jpg"):
        from PIL import Image
        with Image.open(file_name) as im:
            arr = np.asarray(im.convert("RGBA") if im.mode in ("RGBA", "LA", "P") else im.convert("RGB"))
        return arr[:, :, :3] if arr.shape[2] > 3 else arr
    if ext in (".bin", ".raw"):
        return np.load(file_name)
    if ext == ".flo":
        flow = flow_utils.readFlow(file_name)
        if flow is None:
            raise ValueError(f"{file_name}: not a Middlebury .flo file")
        return flow.astype(np.float32)
    return []
