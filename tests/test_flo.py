"""SURVEY.md 8f N3, second half: the Middlebury .flo writer / reader (reference utils/flow_utils.py:7-57).  Byte work: the bar is
bit-exact.  The golden files were written by the reference's own writeFlow (tests/golden/make_golden_flo.py); the oracle
restatement and the product must reproduce them byte for byte."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from oracle import flo_oracle
from utils import flow_utils


def _golden(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


CASES = [("flo_f32_5x7.flo", "f32_5x7", None), ("flo_f64_3x4.flo", "f64_3x4", None), ("flo_chw_view_6x9.flo", "chw_view_6x9", "chw"),
         ("flo_sep_4x4.flo", "sep_u_4x4", "sep"), ("flo_one_1x1.flo", "one_1x1", None)]


@pytest.mark.parametrize("fname,key,kind", CASES)
def test_writer_bit_exact_against_reference_files(tmp_path, fname, key, kind):
    inp = np.load(os.path.join(GOLDEN, "flo_inputs.npz"))
    want = _golden(fname)
    if kind == "sep":
        args = (inp["sep_u_4x4"], inp["sep_v_4x4"])
    elif kind == "chw":
        args = (inp[key].transpose(1, 2, 0),)          # a non-contiguous view, as main.py:388 passes it
        assert not args[0].flags["C_CONTIGUOUS"]
    else:
        args = (inp[key],)
    assert flo_oracle.flo_bytes(*args) == want                         # the restatement is pinned
    path = str(tmp_path / "out.flo")
    flow_utils.writeFlow(path, *args)
    assert open(path, "rb").read() == want                             # the product writes the reference's bytes


def test_reader_matches_reference_readback(tmp_path):
    want = np.load(os.path.join(GOLDEN, "flo_f32_5x7_readback.npy"))
    got = flow_utils.readFlow(os.path.join(GOLDEN, "flo_f32_5x7.flo"))
    assert got.dtype == np.float32 and got.shape == (5, 7, 2)
    assert got.tobytes() == want.tobytes()                             # bit-exact incl. nan / -0.0 / inf
    assert flo_oracle.flo_parse(_golden("flo_f32_5x7.flo")).tobytes() == want.tobytes()
    bad = tmp_path / "bad.flo"
    bad.write_bytes(b"\x00\x00\x00\x00" + _golden("flo_f32_5x7.flo")[4:])
    assert flow_utils.readFlow(str(bad)) is None                       # wrong magic: None, as the reference (:17-19)
    assert flo_oracle.flo_parse(bad.read_bytes()) is None
    with pytest.raises(AssertionError):
        flow_utils.writeFlow(str(tmp_path / "x.flo"), np.zeros((3, 3, 3), np.float32))   # the reference's asserts (:39-40)
    with pytest.raises(AssertionError):
        flow_utils.writeFlow(str(tmp_path / "x.flo"), np.zeros((3, 3), np.float32), np.zeros((3, 4), np.float32))


def test_full_size_round_trip_and_batch_writer(tmp_path):
    """BASELINE size (384 x 512): write -> read is the identity on the float32 bits; save_flows writes exactly what the
    reference's per-item loop (main.py:385-389) writes; empty flows are legal."""
    import torch
    rng = np.random.default_rng(3)
    flows = (rng.standard_normal((3, 2, 384, 512)) * 20).astype(np.float32)
    paths = flow_utils.save_flows(str(tmp_path / "out"), torch.from_numpy(flows), start_index=8)
    assert [os.path.basename(p) for p in paths] == ["000008.flo", "000009.flo", "000010.flo"]
    for i, p in enumerate(paths):
        item = flows[i].transpose(1, 2, 0)
        assert open(p, "rb").read() == flo_oracle.flo_bytes(item)
        back = flow_utils.readFlow(p)
        assert back.tobytes() == np.ascontiguousarray(item).tobytes()
        assert os.path.getsize(p) == 12 + 384 * 512 * 2 * 4
    flow_utils.writeFlow(str(tmp_path / "empty.flo"), np.zeros((0, 5, 2), np.float32))
    assert open(tmp_path / "empty.flo", "rb").read() == flo_oracle.flo_bytes(np.zeros((0, 5, 2), np.float32))
    assert flow_utils.readFlow(str(tmp_path / "empty.flo")).shape == (0, 5, 2)


# ---------------------------------------------------------------- Middlebury colour coding (utils/flow_utils.py:62-204)
FLOWVIS_CASES = ("noise_24x32", "radial_33x41", "specials_9x11", "f64_6x7", "axis_1x8")


@pytest.mark.parametrize("case", FLOWVIS_CASES)
def test_flow2img_matches_the_reference_images(case):
    """Product and oracle against images produced by the reference's own flow2img (tests/golden/make_golden_flowvis.py):
    identical uint8 values -- every direction and radius, unknown-flow markers, nan, exact zeros, float64 input."""
    flow = np.load(os.path.join(GOLDEN, "flowvis_inputs.npz"))[case]
    want = np.load(os.path.join(GOLDEN, "flowvis_%s.npy" % case))
    keep = flow.copy()
    got = flow_utils.flow2img(flow)
    assert got.dtype == np.uint8 and got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(flow, keep, equal_nan=True)                  # unlike the reference, the argument is left alone
    assert np.array_equal(flo_oracle.flow2img_oracle(flow), want)


def test_color_wheel_and_edge_cases(tmp_path):
    wheel = flow_utils.make_color_wheel()
    assert wheel.shape == (55, 3) and np.array_equal(wheel, np.load(os.path.join(GOLDEN, "flowvis_wheel.npy")))
    assert np.array_equal(flo_oracle.wheel_oracle(), wheel)
    assert (flow_utils.flow2img(np.zeros((4, 5, 2), np.float32)) == 0).all()           # 0 / 0 -> nan -> black, as the reference
    one = flow_utils.flow2img(np.array([[[3.0, 0.0]]], np.float32))                    # a single vector is its own maximum: radius 1 + eps
    assert one.shape == (1, 1, 3) and np.array_equal(one, flo_oracle.flow2img_oracle(np.array([[[3.0, 0.0]]], np.float32)))
    img = flow_utils.compute_color(np.array([[0.5, np.nan]]), np.array([[0.0, 0.2]]))
    assert img.dtype == np.float64 and img.shape == (1, 2, 3) and (img[0, 1] == 0).all() and (img == np.floor(img)).all()
    # .flo file -> colour image -> PNG next to it (the reference's visulize_flow_file, its spelling)
    flow = np.load(os.path.join(GOLDEN, "flowvis_inputs.npz"))["radial_33x41"]
    flow_utils.writeFlow(str(tmp_path / "r.flo"), flow)
    img = flow_utils.visulize_flow_file(str(tmp_path / "r.flo"), str(tmp_path))
    assert np.array_equal(img, np.load(os.path.join(GOLDEN, "flowvis_radial_33x41.npy")))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "r-vis.png").convert("RGB")), img)


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="reference checkout not present")
def test_flow2img_against_the_live_reference():
    import sys
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "/root/reference")
    try:
        import importlib
        ref = importlib.import_module("utils.flow_utils") if "utils.flow_utils" not in sys.modules else None
    finally:
        sys.path.remove("/root/reference")
    if ref is None or not hasattr(ref, "TAG_CHAR"):     # `utils` already resolved to the product package in this process
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_flow_utils", "/root/reference/utils/flow_utils.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    rng = np.random.default_rng(77)
    for shape, scale in (((17, 23, 2), 1.0), ((8, 8, 2), 1e-6), ((30, 5, 2), 400.0)):
        flow = (rng.standard_normal(shape) * scale).astype(np.float32)
        flow[0, 0] = (5e7, 0.0)
        with np.errstate(all="ignore"):
            want = ref.flow2img(flow.copy())
        assert np.array_equal(flow_utils.flow2img(flow), want)


@pytest.mark.gpu
def test_save_flows_from_device_tensor(tmp_path, dev):
    """The batch writer fed with a device tensor (the network's output): interleave on the GPU, one transfer, same bytes."""
    import torch
    g = torch.Generator().manual_seed(4)
    flows = torch.randn(4, 2, 96, 128, generator=g) * 15
    flows[0, 0, 0, 0] = float("nan"); flows[1, 1, 5, 5] = float("inf")
    paths = flow_utils.save_flows(str(tmp_path / "gpu"), flows.to(dev))
    for i, p in enumerate(paths):
        assert open(p, "rb").read() == flo_oracle.flo_bytes(flows[i].numpy().transpose(1, 2, 0))
    half = flow_utils.save_flows(str(tmp_path / "gpu16"), flows.to(dev).half())     # fp16 network output: widened, not reinterpreted
    assert open(half[2], "rb").read() == flo_oracle.flo_bytes(flows[2].half().float().numpy().transpose(1, 2, 0))


def test_truncated_header_is_rejected(tmp_path, capsys):
    """a right tag with width / height cut off prints the reference's kind of message and returns None (the reference
    raises struct.error there, utils/flow_utils.py:13-21); read_gen turns None into a ValueError instead of an AttributeError."""
    import struct
    from utils import flow_utils, frame_utils
    f = tmp_path / "short.flo"
    f.write_bytes(struct.pack("<f", flow_utils.TAG_FLOAT) + b"\x05\x00")
    assert flow_utils.readFlow(str(f)) is None
    assert "Invalid .flo file" in capsys.readouterr().out
    import pytest
    with pytest.raises(ValueError):
        frame_utils.read_gen(str(f))
