"""SURVEY.md 8f N4, 'real datasets': the folder datasets of harness/data.py keep the reference's conventions
(datasets.py:113-169, :316-365; utils/frame_utils.py:6-20) and feed a training step."""
import numpy as np
import pytest
import torch


def test_folder_datasets_follow_reference_conventions(tmp_path):
    from PIL import Image
    from harness import data
    from utils import flow_utils, frame_utils
    root = data.write_synthetic_folder(str(tmp_path / "chairs"), pairs=3, height=136, width=200, seed=1)
    ds = data.FlyingChairs(root)                                   # inference: centre crop to multiples of 64 (:138-140)
    assert len(ds) == 3 and ds.render_size == [128, 192] and ds.frame_size == (136, 200, 3)
    (images,), (flow,) = ds[1]
    assert images.dtype == torch.float32 and tuple(images.shape) == (3, 2, 128, 192) and tuple(flow.shape) == (2, 128, 192)
    raw1 = np.asarray(Image.open(tmp_path / "chairs" / "00001_img1.ppm"))
    assert np.array_equal(images[:, 0].numpy(), raw1[4:132, 4:196].transpose(2, 0, 1).astype(np.float32))   # StaticCenterCrop (:23-28)
    gt = flow_utils.readFlow(str(tmp_path / "chairs" / "00001_flow.flo"))
    assert np.array_equal(flow.numpy(), gt[4:132, 4:196].transpose(2, 0, 1))
    assert 0.0 <= float(images.min()) and float(images.max()) <= 255.0
    # training crops: random window, the same for both frames and the flow (:13-21, :151-156)
    tr = data.FlyingChairs(root, is_cropped=True, crop_size=(64, 96), replicates=2)
    assert len(tr) == 6
    (im,), (fl,) = tr[4]
    assert tuple(im.shape) == (3, 2, 64, 96) and tuple(fl.shape) == (2, 64, 96)
    full1 = np.asarray(Image.open(tmp_path / "chairs" / "00001_img1.ppm")).astype(np.float32)
    hits = [(y, x) for y in range(136 - 64 + 1) for x in range(200 - 96 + 1)
            if np.array_equal(full1[y:y + 64, x:x + 96, 0], im[0, 0].numpy())]
    assert len(hits) >= 1
    y, x = hits[0]
    assert np.array_equal(fl.numpy(), flow_utils.readFlow(str(tmp_path / "chairs" / "00001_flow.flo"))[y:y + 64, x:x + 96].transpose(2, 0, 1))
    # frames-only folder: consecutive pairs, zero target (:329-333, :357)
    for i in range(4):
        Image.fromarray(np.full((64, 128, 4 if i == 2 else 3), 10 * i, np.uint8)).save(tmp_path / ("f%02d.png" % i))
    seq = data.ImagesFromFolder(str(tmp_path), iext="png")
    assert len(seq) == 3
    (im,), (zero,) = seq[2]
    assert tuple(im.shape) == (3, 2, 64, 128) and float(im[:, 0].mean()) == 20.0 and float(im[:, 1].mean()) == 30.0   # alpha dropped
    assert tuple(zero.shape) == (3, 2, 64, 128)[:1] + (2, 64, 128) and float(zero.abs().sum()) == 0.0
    assert frame_utils.read_gen("x.unknown") == []
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)
    (batch,), (target,) = next(iter(loader))
    assert tuple(batch.shape) == (2, 3, 2, 128, 192) and tuple(target.shape) == (2, 2, 128, 192)


@pytest.mark.gpu
def test_train_step_on_folder_data(tmp_path, dev):
    """Disk -> DataLoader -> FlowNet2C training step on the HIP layers -> .flo files of the inferred flows."""
    from harness import data
    from harness.train import Trainer
    from utils import flow_utils
    root = data.write_synthetic_folder(str(tmp_path / "chairs"), pairs=4, height=128, width=192, seed=2)
    loader = torch.utils.data.DataLoader(data.FlyingChairs(root), batch_size=2, shuffle=False)
    tr = Trainer(dev, seed=3)
    losses = []
    for _ in range(2):
        for (images,), (flow,) in loader:
            loss, epe = tr.train_step(images.to(dev), flow.to(dev))
            losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    (images,), _ = next(iter(loader))
    paths = flow_utils.save_flows(str(tmp_path / "out"), tr.infer(images.to(dev)))
    back = flow_utils.readFlow(paths[1])
    assert back.shape == (128, 192, 2) and np.isfinite(back).all()
