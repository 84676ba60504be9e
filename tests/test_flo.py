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
