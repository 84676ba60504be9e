"""Real-data input path of the harness ('synthetic + real datasets'): the two folder layouts of the
reference's datasets.py that need no benchmark download logic, with the reference's cropping and tensor conventions --

* ``ImagesFromFolder``  (datasets.py:316-365)  consecutive frames of one folder -> pairs, zero target;
* ``FlyingChairs``      (datasets.py:113-169)  ``*.ppm`` pairs + ``*.flo`` ground truth (also what ``write_synthetic_folder``
  produces, so the whole path -- image decode, .flo decode, crop, batch, train step -- runs without external data).

An item is ``([images 3 x 2 x H x W float32 in 0..255], [flow 2 x H x W float32])`` exactly as the reference returns it
(``np.array(images).transpose(3, 0, 1, 2)``).  Sizes are rounded down to multiples of 64 for inference (:338-340).
Everything else of datasets.py (Sintel / Things / ChairsSDHom directory walking) is out of scope."""
import os
import random
from glob import glob
from os.path import join

import numpy as np
import torch
import torch.utils.data as data

from utils import flow_utils, frame_utils


class _StaticCrop:
    """A crop window chosen ONCE per sample and applied to both images and the flow (datasets.py:13-28: the reference's
    StaticRandomCrop draws the corner at construction, StaticCenterCrop centres the window)."""

    def __init__(self, image_size, crop_size, centred):
        (h, w), (th, tw) = image_size, crop_size
        top, left = ((h - th) // 2, (w - tw) // 2) if centred else (random.randint(0, h - th), random.randint(0, w - tw))
        self.rows, self.cols = slice(top, top + th), slice(left, left + tw)

    def __call__(self, img):
        return img[self.rows, self.cols, :]


def StaticRandomCrop(image_size, crop_size):
    return _StaticCrop(image_size, crop_size, centred=False)


def StaticCenterCrop(image_size, crop_size):
    return _StaticCrop(image_size, crop_size, centred=True)


class _PairFolder(data.Dataset):
    def __init__(self, image_list, flow_list, is_cropped, crop_size, inference_size, replicates):
        self.image_list, self.flow_list = image_list, flow_list
        self.is_cropped, self.crop_size, self.replicates = is_cropped, tuple(crop_size), replicates
        self.size = len(image_list)
        if self.size == 0:
            raise FileNotFoundError("no image pairs found")
        self.frame_size = frame_utils.read_gen(image_list[0][0]).shape
        rs = list(inference_size)
        if rs[0] < 0 or rs[1] < 0 or self.frame_size[0] % 64 or self.frame_size[1] % 64:
            rs = [(self.frame_size[0] // 64) * 64, (self.frame_size[1] // 64) * 64]
        self.render_size = rs

    def __getitem__(self, index):
        index = index % self.size
        images = [frame_utils.read_gen(f) for f in self.image_list[index]]
        image_size = images[0].shape[:2]
        cropper = StaticRandomCrop(image_size, self.crop_size) if self.is_cropped else StaticCenterCrop(image_size, self.render_size)
        images = np.array([cropper(im) for im in images]).transpose(3, 0, 1, 2)
        images = torch.from_numpy(images.astype(np.float32))
        if self.flow_list is None:
            return [images], [torch.zeros(images.size()[0:1] + (2,) + images.size()[-2:])]
        flow = cropper(frame_utils.read_gen(self.flow_list[index])).transpose(2, 0, 1)
        return [images], [torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32))]

    def __len__(self):
        return self.size * self.replicates


class ImagesFromFolder(_PairFolder):
    def __init__(self, root, iext="png", is_cropped=False, crop_size=(384, 512), inference_size=(-1, -1), replicates=1):
        images = sorted(glob(join(root, "*." + iext)))
        super().__init__([[images[i], images[i + 1]] for i in range(len(images) - 1)], None, is_cropped, crop_size,
                         inference_size, replicates)


class FlyingChairs(_PairFolder):
    def __init__(self, root, is_cropped=False, crop_size=(384, 512), inference_size=(-1, -1), replicates=1):
        images, flows = sorted(glob(join(root, "*.ppm"))), sorted(glob(join(root, "*.flo")))
        assert len(images) // 2 == len(flows)
        super().__init__([[images[2 * i], images[2 * i + 1]] for i in range(len(flows))], flows, is_cropped, crop_size,
                         inference_size, replicates)


def write_synthetic_folder(root, pairs=4, height=128, width=192, seed=0):
    """A FlyingChairs-layout folder of synthetic pairs: smooth random textures, the second frame = the first warped by a smooth
    random flow (nearest-neighbour backward warp), ground truth as .flo."""
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:height, 0:width]
    for i in range(pairs):
        coarse = rng.random((height // 16 + 2, width // 16 + 2, 3))
        img1 = np.kron(coarse, np.ones((16, 16, 1)))[:height, :width]
        img1 = (255 * (0.7 * img1 + 0.3 * rng.random((height, width, 3)))).astype(np.uint8)
        flow = np.stack([np.full((height, width), rng.uniform(-6, 6)) + 2 * np.sin(ys / 17.0 + i),
                         np.full((height, width), rng.uniform(-6, 6)) + 2 * np.cos(xs / 23.0 - i)], -1).astype(np.float32)
        sx = np.clip(np.rint(xs - flow[..., 0]).astype(int), 0, width - 1)
        sy = np.clip(np.rint(ys - flow[..., 1]).astype(int), 0, height - 1)
        img2 = img1[sy, sx]
        Image.fromarray(img1).save(join(root, "%05d_img1.ppm" % i))
        Image.fromarray(img2).save(join(root, "%05d_img2.ppm" % i))
        flow_utils.writeFlow(join(root, "%05d_flow.flo" % i), flow)
    return root
