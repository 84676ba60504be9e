"""Model-level golden vectors from the reference's UNMODIFIED models.py / networks/*.py / losses.py.

The reference's Python cannot travel to the GPU box in any form, and its custom layers have no CPU path; so HERE (dev container)
the reference's own classes -- models.FlowNet2C, models.FlowNet2, losses.MultiScale, and the reference's own wrapper modules
networks/{correlation,resample2d,channelnorm}_package/*.py -- run on the CPU on top of stand-ins of the three extension modules
that answer with the oracle (bit-exact to the reference's CUDA kernels, tests/test_oracle_vs_ref.py), at the BASELINE shape
bs 8 @ 384 x 512 with weights and inputs that depend on names and a seed only (tests/refmodel_fixture.py).  What they produce is
stored; tests/test_reference_model_golden.py rebuilds weights and inputs on the GPU box, runs harness.FlowNet2C / FlowNet2 on the
HIP layers and compares.

    python tests/golden/make_golden_ref_models.py          # -> tests/golden/refmodels_8x384x512.npz   (a few minutes of CPU)
"""
import importlib
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import refmodel_fixture as fx  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

REF = "/root/reference"


def main():
    assert os.path.isdir(os.path.join(REF, "networks")), "needs the reference checkout"
    fx.install_oracle_extensions(Oracle())
    sys.path.insert(0, REF)
    models = importlib.import_module("models")
    losses = importlib.import_module("losses")
    assert models.__file__.startswith(REF) and losses.__file__.startswith(REF)
    import networks.correlation_package.correlation as rc
    assert rc.__file__.startswith(REF)                                            # the reference's own wrapper, unmodified
    args = SimpleNamespace(rgb_max=255.0, fp16=False)
    inputs, target = fx.make_inputs()
    out = {"seed": fx.SEED, "shape": np.array([fx.B, fx.H, fx.W]), "input_checksum": np.array([fx.checksum(inputs), fx.checksum(target)])}
    torch.set_num_threads(8)

    # ---- FlowNet2C: inference, then one forward + MultiScale-L1 loss + backward (models.py:187-253, losses.py:52-86)
    t0 = time.time()
    net = fx.fill_state_dict(models.FlowNet2C(args))
    out["flownet2c_param_checksum"] = np.array([fx.state_checksum(net)])
    net.eval()
    with torch.no_grad():
        y = net(inputs)
    assert tuple(y.shape) == (fx.B, 2, fx.H, fx.W)
    out["flownet2c_flow_sub4"] = y[:, :, ::4, ::4].numpy().copy()
    out["flownet2c_flow_plane_sums"] = y.double().sum(dim=(2, 3)).numpy()
    out["flownet2c_flow_plane_abs_sums"] = y.double().abs().sum(dim=(2, 3)).numpy()
    print("FlowNet2C inference", round(time.time() - t0, 1), "s; |flow| max", float(y.abs().max()), flush=True)
    net.train()
    crit = losses.MultiScale(args, startScale=4, numScales=5, l_weight=0.32, norm="L1")
    flows = net(inputs)
    loss, epe = crit(flows, target)
    loss.backward()
    for i, f in enumerate(flows):
        out[f"flownet2c_train_flow{i}"] = f.detach().numpy().copy()
    out["flownet2c_loss_epe"] = np.array([float(loss.detach()), float(epe.detach())])
    names = [n for n, _ in net.named_parameters()]
    out["flownet2c_grad_names"] = np.array(names)
    out["flownet2c_grad_digest"] = np.stack([fx.param_digest(p.grad) for _, p in net.named_parameters()])
    print("FlowNet2C fwd+bwd", round(time.time() - t0, 1), "s; loss", float(loss), "epe", float(epe), flush=True)
    del net, flows, loss

    # ---- FlowNet2 (CSS + SD + fusion, models.py:25-185): inference through all four warp sites and the ChannelNorm
    t0 = time.time()
    net = fx.fill_state_dict(models.FlowNet2(args)).eval()
    out["flownet2_param_checksum"] = np.array([fx.state_checksum(net)])
    with torch.no_grad():
        y = net(inputs)
    assert tuple(y.shape) == (fx.B, 2, fx.H, fx.W)
    out["flownet2_flow_sub4"] = y[:, :, ::4, ::4].numpy().copy()
    out["flownet2_flow_plane_sums"] = y.double().sum(dim=(2, 3)).numpy()
    out["flownet2_flow_plane_abs_sums"] = y.double().abs().sum(dim=(2, 3)).numpy()
    print("FlowNet2 inference", round(time.time() - t0, 1), "s; |flow| max", float(y.abs().max()), flush=True)
    path = os.path.join(HERE, "refmodels_8x384x512.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
