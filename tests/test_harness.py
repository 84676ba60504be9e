"""the FlowNet2C harness -- network definition (state-dict compatible with the reference's class),
bucketed gradient all-reduce (world_size-2 gloo on CPU), one training step and inference on the GPU."""
import importlib
import os
import socket
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from conftest import PKG, ROOT

REF = "/root/reference"


def test_flownet2c_parameter_table():
    from harness.flownet2c import FlowNet2C
    net = FlowNet2C()
    assert sum(p.numel() for p in net.parameters()) == 39_175_298          # FlowNetC.py:11
    sd = net.state_dict()
    assert sd["conv1.0.weight"].shape == (64, 3, 7, 7) and sd["conv3_1.0.weight"].shape == (256, 473, 3, 3)
    assert sd["deconv2.0.weight"].shape == (386, 64, 4, 4) and sd["upsampled_flow3_to_2.bias"].shape == (2,)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="reference checkout not present")
def test_flownet2c_state_dict_matches_reference_class():
    """Key names and shapes equal those of the reference's models.FlowNet2C, so its checkpoints load unchanged."""
    from harness.flownet2c import FlowNet2C
    ours = {k: tuple(v.shape) for k, v in FlowNet2C().state_dict().items()}
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "models"]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        ref_models = importlib.import_module("models")
        ref = ref_models.FlowNet2C(SimpleNamespace(rgb_max=255.0, fp16=False))
        theirs = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert ours == theirs
        net = FlowNet2C()
        net.load_state_dict(ref.state_dict())                               # strict
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved_mods]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="reference checkout not present")
def test_flownet2_state_dict_matches_reference_class():
    """The full stack (FlowNetC + 2 x FlowNetS + FlowNetSD + FlowNetFusion): same keys and shapes as models.FlowNet2."""
    from harness.flownet2 import FlowNet2
    ours = FlowNet2()
    assert sum(p.numel() for p in ours.parameters()) == 162_518_834          # models.py:27
    mine = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "models"]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        ref_models = importlib.import_module("models")
        ref = ref_models.FlowNet2(SimpleNamespace(rgb_max=255.0, fp16=False))
        theirs = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert mine == theirs
        ours.load_state_dict(ref.state_dict())
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved_mods]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


# ---------------------------------------------------------------- bucketed all-reduce, world_size 2 on gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _small_net():
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(8, 8, 3, padding=1),
                               torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(8, 2, 3, padding=1))


def _ddp_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import dist_utils
    from harness.ddp import BucketedGradAllReduce
    dist_utils.init_from_env(backend="gloo")
    torch.manual_seed(7 + rank)                       # different initial weights: the broadcast must fix that
    net = _small_net()
    dist_utils.broadcast_state(net, src=0)
    plain = _small_net()
    plain.load_state_dict(net.state_dict())
    red = BucketedGradAllReduce(net, bucket_bytes=1024)   # several buckets
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randn(4, 2, 8, 8, generator=g)
    lo, hi = dist_utils.shard_bounds(4, world, rank)
    for step in range(2):
        red.zero_grad()
        red.reset()
        (net(x[lo:hi]) - y[lo:hi]).abs().mean().backward()
        red.finish()
        # expected: mean over ranks of the per-rank gradients of the same replica
        plain.zero_grad()
        (plain(x[lo:hi]) - y[lo:hi]).abs().mean().backward()
        for pn, pp in zip(net.parameters(), plain.parameters()):
            mine = pp.grad.clone()
            dist.all_reduce(mine)
            assert torch.allclose(pn.grad, mine / world, rtol=1e-6, atol=1e-7), step
            assert pn.grad.data_ptr() == red._grad_ptr(pn)          # still a view into its bucket
    dist.barrier()
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_bucketed_allreduce_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


# ---------------------------------------------------------------- GPU: the network around the HIP layers
def _corr_torch(a, b):
    """Independent formulation of FlowNetC's cost volume (kernel 1, displacement 20, stride2 2): 441 shifted products."""
    H, W = a.shape[2:]
    bp = F.pad(b, (20, 20, 20, 20))
    outs = []
    for tj in range(21):
        for ti in range(21):
            dy, dx = 2 * (tj - 10), 2 * (ti - 10)
            outs.append((a * bp[:, :, 20 + dy:20 + dy + H, 20 + dx:20 + dx + W]).mean(1, keepdim=True))
    return torch.cat(outs, 1)


@pytest.mark.gpu
def test_flownet2c_forward_paths_agree(dev):
    from harness.flownet2c import FlowNet2C
    from harness.train import synthetic_batch
    torch.manual_seed(1)
    net = FlowNet2C().to(dev).eval()
    inputs, _ = synthetic_batch(2, 128, 192, dev, seed=4)
    with torch.no_grad():
        fused = net(inputs)                                   # fused LeakyReLU + concat epilogue
    assert tuple(fused.shape) == (2, 2, 128, 192) and torch.isfinite(fused).all()
    train_fused = net(inputs).detach()                        # grad mode on: the same fused op, differentiable (round 4)
    assert float((train_fused - fused).abs().max()) <= 1e-5 * float(fused.abs().max())   # (the convolutions may pick other kernels)
    net.fused_training = False
    unfused = net(inputs).detach()                            # Correlation + LeakyReLU + cat as FlowNetC.py:86-92 writes them
    scale = float(fused.abs().max())
    assert float((fused - unfused).abs().max()) <= 1e-5 * scale
    # the same network with the cost volume formed by plain PyTorch ops
    class TorchCorr(torch.nn.Module):
        def forward(self, a, b):
            return _corr_torch(a, b)
    hip_corr = net.corr
    net.corr = TorchCorr()
    try:
        ref = net(inputs).detach()
    finally:
        net.corr = hip_corr
    assert float((unfused - ref).abs().max()) <= 2e-4 * scale


@pytest.mark.gpu
def test_flownet2c_sintel_size_paths_agree(dev):
    """A Sintel-size frame pair (1024 x 448: conv3 maps are 56 x 128, i.e. wider than the 64 pixels one tile row of the f16x2
    kernels holds): the fused inference path, the separate modules and a plain-torch cost volume give the same flow, and the
    Correlation module's gradients on features of that size are those of torch autograd."""
    from harness.flownet2c import FlowNet2C
    from harness.train import synthetic_batch
    from networks.correlation_package.correlation import Correlation
    torch.manual_seed(12)
    net = FlowNet2C().to(dev).eval()
    inputs, _ = synthetic_batch(1, 448, 1024, dev, seed=13)
    with torch.no_grad():
        fused = net(inputs)
    assert tuple(fused.shape) == (1, 2, 448, 1024) and torch.isfinite(fused).all()
    net.fused_training = False
    unfused = net(inputs).detach()
    scale = float(fused.abs().max())
    assert float((fused - unfused).abs().max()) <= 1e-5 * scale

    class TorchCorr(torch.nn.Module):
        def forward(self, a, b):
            return _corr_torch(a, b)
    hip_corr = net.corr
    net.corr = TorchCorr()
    try:
        ref = net(inputs).detach()
    finally:
        net.corr = hip_corr
    assert float((unfused - ref).abs().max()) <= 2e-4 * scale
    g = torch.Generator().manual_seed(14)
    a = torch.randn(1, 64, 56, 128, generator=g).to(dev).requires_grad_()
    b = torch.randn(1, 64, 56, 128, generator=g).to(dev).requires_grad_()
    go = torch.randn(1, 441, 56, 128, generator=g).to(dev) * 1e-6            # training-size gradOutput
    Correlation(20, 1, 20, 1, 2, 1)(a, b).backward(go)
    g1, g2 = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    _corr_torch(a, b).backward(go)
    for got, want in ((g1, a.grad), (g2, b.grad)):
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.gpu
def test_flownet2c_train_steps(dev):
    from harness.train import Trainer, synthetic_batch
    tr = Trainer(dev, lr=1e-4, seed=2, bucket_bytes=16 << 20)
    assert len(tr.reducer.buckets) >= 5
    inputs, target = synthetic_batch(2, 128, 192, dev, seed=5)
    before = [p.detach().clone() for p in list(tr.model.parameters())[:3]]
    losses = [float(tr.train_step(inputs, target)[0]) for _ in range(4)]
    assert all(l == l and l < 1e6 for l in losses)
    assert losses[-1] < losses[0]                             # same batch four times: Adam must make progress
    assert any(not torch.equal(b, p) for b, p in zip(before, list(tr.model.parameters())[:3]))
    flow = tr.infer(inputs)
    assert tuple(flow.shape) == (2, 2, 128, 192) and torch.isfinite(flow).all()


@pytest.mark.gpu
def test_flownet2_full_stack_inference(dev):
    """FlowNet2 (CSS + SD + fusion): the fused no-grad path (correlation epilogue, WarpDiffNormCat) against the same network
    through the separate Correlation / Resample2d / ChannelNorm modules; fp16 convolution stacks stay close to fp32."""
    from harness.flownet2 import FlowNet2
    from harness.train import synthetic_batch
    torch.manual_seed(3)
    net = FlowNet2().to(dev).eval()
    # the random-init stack amplifies: scale the weights down so that flows stay within the image
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    inputs, _ = synthetic_batch(2, 128, 192, dev, seed=6)
    with torch.no_grad():
        fused = net(inputs)
    assert tuple(fused.shape) == (2, 2, 128, 192) and torch.isfinite(fused).all()
    net.fused_training = False
    unfused = net(inputs).detach()                    # grad mode, fused_training off: separate modules
    scale = max(float(unfused.abs().max()), 1e-6)
    assert float((fused - unfused).abs().max()) <= 1e-3 * scale
    half = net.half()
    with torch.no_grad():
        out16 = half(inputs.half())
    assert out16.dtype == torch.float16 and torch.isfinite(out16).all()


@pytest.mark.gpu
def test_flownet2_hip_layers_vs_torch_stand_ins(dev):
    """Second half of the transitive pin of SURVEY 8 row (g) (the reference's models cannot travel to the GPU box):
    tests/test_harness_pin.py shows harness.FlowNet2 == the reference's models.FlowNet2 when both use plain-torch layers (CPU);
    here the SAME harness network runs on the GPU once with the HIP Correlation / Resample2d / ChannelNorm and once with
    those plain-torch layers: the flows agree, so the HIP layers inside the reference's architecture give the reference's
    result."""
    from harness.flownet2 import FlowNet2
    from harness.train import synthetic_batch

    class TorchResample(torch.nn.Module):
        def forward(self, img, flow):
            B, _, H, W = flow.shape
            ys, xs = torch.meshgrid(torch.arange(H, dtype=flow.dtype, device=flow.device),
                                    torch.arange(W, dtype=flow.dtype, device=flow.device), indexing="ij")
            gx = (xs + flow[:, 0]) * (2.0 / (W - 1)) - 1.0
            gy = (ys + flow[:, 1]) * (2.0 / (H - 1)) - 1.0
            return F.grid_sample(img, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="border", align_corners=True)

    class TorchNorm(torch.nn.Module):
        def forward(self, x):
            return x.pow(2).sum(1, keepdim=True).sqrt()

    class TorchCorr(torch.nn.Module):
        def forward(self, a, b):
            return _corr_torch(a, b)

    torch.manual_seed(8)
    net = FlowNet2().to(dev).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)                                # keeps the random-init stack's flows inside the image
    inputs, _ = synthetic_batch(2, 128, 192, dev, seed=9)
    net.fused_training = False
    hip = net(inputs).detach()                         # grad mode, fused_training off: the separate HIP modules
    saved = (net.flownetc.corr, net.channelnorm, net.resample1, net.resample2, net.resample3, net.resample4)
    net.flownetc.corr, net.channelnorm = TorchCorr(), TorchNorm()
    net.resample1 = net.resample2 = net.resample3 = net.resample4 = TorchResample()
    try:
        ref = net(inputs).detach()
    finally:
        net.flownetc.corr, net.channelnorm, net.resample1, net.resample2, net.resample3, net.resample4 = saved
    scale = max(float(ref.abs().max()), 1e-6)
    assert float((hip - ref).abs().max()) <= 1e-3 * scale, (float((hip - ref).abs().max()), scale)


def _capture_fused_rows(modules):
    """Forward pre/post hooks on the fused-row modules of a network: for every call, the input tensors (detached copies), the output, the
    gradient that arrives at the output and the gradients the row sends to each differentiable input -- taken on a VIEW of the input
    made for this call, so that what other consumers of the same tensor contribute (flow_s2 also feeds the fusion network) is not in it."""
    calls, handles = [], []

    def pre(mod, args):
        rec = {"mod": mod, "inputs": [a.detach().clone() for a in args], "gin": [None] * len(args)}
        views = []
        for k, a in enumerate(args):
            v = a.view_as(a)
            if v.requires_grad:
                v.register_hook(lambda g, rec=rec, k=k: rec["gin"].__setitem__(k, g.detach().clone()))
            views.append(v)
        mod._fn2_rec = rec
        return tuple(views)

    def post(mod, args, out):
        rec = mod._fn2_rec
        rec["out"] = out.detach().clone()
        if out.requires_grad:
            out.register_hook(lambda g, rec=rec: rec.__setitem__("gout", g.detach().clone()))
        calls.append(rec)

    for m in modules:
        handles += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
    return calls, handles


@pytest.mark.gpu
def test_fused_rows_replayed_inside_one_training_pass(dev):
    """the pin of the fused training rows without MIOpen between the two sides and without a tolerance to tune.
    ONE fused training pass of harness.FlowNet2 (WarpDiffNormCat at models.py:133-138 / :145-150, WarpDiffNorm at :157-161 / :170-174) and
    of harness.FlowNet2C (CorrelationLeakyReLUCat, FlowNetC.py:86-92) is recorded -- the tensors that entered each fused row, the
    gradient that came back into it, the gradients it sent on --, and exactly those tensors are replayed through the UNFUSED HIP layers
    under autograd: outputs and flow / feature gradients must be bit-identical."""
    from harness.flownet2 import FlowNet2
    from harness.flownet2c import FlowNet2C
    from harness.train import synthetic_batch
    from networks.channelnorm_package.channelnorm import ChannelNorm
    from networks.correlation_package.correlation import Correlation
    from networks.resample2d_package.resample2d import Resample2d
    torch.manual_seed(11)
    net = FlowNet2().to(dev).train()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    inputs, target = synthetic_batch(2, 128, 192, dev, seed=4)
    calls, handles = _capture_fused_rows([net.warp_cat, net.warp_err])
    net.fused_training = True
    out = net(inputs)
    (out - target).abs().mean().backward()
    for h in handles:
        h.remove()
    assert [type(c["mod"]).__name__ for c in calls] == ["WarpDiffNormCat", "WarpDiffNormCat", "WarpDiffNorm", "WarpDiffNorm"]
    rs, cn = Resample2d(), ChannelNorm()
    for c in calls:
        x, flow = c["inputs"]
        assert c["gin"][0] is None and c["gin"][1] is not None and "gout" in c        # the pair is the network's input: flow gradient only
        fl = flow.clone().requires_grad_(True)
        warped = rs(x[:, 3:], fl)
        norm = cn(x[:, :3] - warped)
        unfused = torch.cat((x, warped, fl / net.div_flow, norm), 1) if type(c["mod"]).__name__ == "WarpDiffNormCat" else norm
        assert torch.equal(unfused.detach(), c["out"]), type(c["mod"]).__name__
        unfused.backward(c["gout"])
        assert float(c["gin"][1].abs().max()) > 0
        assert torch.equal(fl.grad, c["gin"][1]), (type(c["mod"]).__name__, float((fl.grad - c["gin"][1]).abs().max()))
    # FlowNet2C: the fused correlation epilogue (LeakyReLU + concat) and its backward (mask pass + correlation backward kernels)
    torch.manual_seed(12)
    netc = FlowNet2C().to(dev).train()
    calls, handles = _capture_fused_rows([netc.corr_fused])
    flows = netc(inputs)
    sum(f.abs().mean() for f in flows).backward()
    for h in handles:
        h.remove()
    assert len(calls) == 1
    c = calls[0]
    a, b, redir = (t.clone().requires_grad_(True) for t in c["inputs"])
    unfused = torch.cat((redir, F.leaky_relu(Correlation(20, 1, 20, 1, 2, 1)(a, b), 0.1)), 1)
    assert torch.equal(unfused.detach(), c["out"])
    unfused.backward(c["gout"])
    for k, t in enumerate((a, b, redir)):
        assert c["gin"][k] is not None and float(c["gin"][k].abs().max()) > 0
        assert torch.equal(t.grad, c["gin"][k]), (k, float((t.grad - c["gin"][k]).abs().max()))


@pytest.mark.gpu
def test_flownet2_trains_through_the_fused_warp(dev):
    """harness.FlowNet2 in grad mode runs WarpDiffNormCat (one kernel forward, one kernel backward) at its two
    warp-concat sites (models.py:133-138, :145-150); parameter gradients of one backward pass agree with the same network
    composed of the separate Resample2d / ChannelNorm modules under autograd (`fused_training = False`)."""
    from harness.flownet2 import FlowNet2
    from harness.train import synthetic_batch
    torch.manual_seed(11)
    net = FlowNet2().to(dev).train()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    inputs, target = synthetic_batch(2, 128, 192, dev, seed=4)
    grads = {}
    # (two discarded passes first: MIOpen picks its convolution algorithms on the first calls of a shape, and a pass that ran with
    # other algorithms than the next one differs from it by more than any of the layers under test)
    for key, fused in (("warm-up", True), ("warm-up 2", False), ("fused", True), ("unfused", False), ("unfused again", False)):
        net.fused_training = fused
        net.zero_grad(set_to_none=True)
        out = net(inputs)
        (out - target).abs().mean().backward()
        grads[key] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        assert torch.isfinite(out).all()
    assert grads["fused"].keys() == grads["unfused"].keys() and len(grads["fused"]) > 100

    def rel_l2(a, b):
        num = sum(float(((a[n] - b[n]).double() ** 2).sum()) for n in a) ** 0.5
        return num / sum(float((b[n].double() ** 2).sum()) for n in a) ** 0.5

    def worst(a, b):
        return max(float((a[n] - b[n]).abs().max()) / max(float(b[n].abs().max()), 1e-12) for n in a)
    # The forward of this network is not bit-reproducible from pass to pass (MIOpen), and a bilinear warp's gradient jumps where a
    # sample crosses an integer coordinate: now and then ONE pixel flips between two passes and a small bias gradient moves by up to
    # 0.8 % -- between two unfused passes as often as between a fused and an unfused one (scripts/fused_training_probe.py).  The bar is
    # therefore the relative L2 distance over ALL parameter gradients (1.6-2.5e-7 between any two passes, 4e-7 ... 4e-5 with a flip; a
    # wrong fused backward would show at 1e-2 or more), with the per-parameter maximum as a loose second check.  What the fused rows themselves
    # contribute is pinned bit for bit at the layer boundary: test_warp_diff_norm_cat_backward, test_warp_diff_norm.
    noise = rel_l2(grads["unfused again"], grads["unfused"])
    diff = rel_l2(grads["fused"], grads["unfused"])
    assert diff <= max(1e-3, 10.0 * noise), (diff, noise)
    assert worst(grads["fused"], grads["unfused"]) <= 5e-2
    # The tight bar: every parameter's gradient within 10x what two UNFUSED passes differ by on that parameter.  It
    # holds whenever no bilinear sample flips between the passes; a flip (either pair of passes shows them equally often,
    # profiles/r06_forward_reproducibility.log names the op that makes the forward differ from pass to pass) is reported as an
    # expected failure, not hidden behind a wide tolerance: what the fused rows compute is pinned bit for bit by
    # test_fused_rows_replayed_inside_one_training_pass on the tensors of one and the same pass.
    scale = {n: max(float(grads["unfused"][n].abs().max()), 1e-30) for n in grads["unfused"]}
    per_noise = {n: float((grads["unfused again"][n] - grads["unfused"][n]).abs().max()) for n in grads["unfused"]}
    per_diff = {n: float((grads["fused"][n] - grads["unfused"][n]).abs().max()) for n in grads["unfused"]}
    # one pair of passes is a poor estimate of a single parameter's noise (two passes now and then agree to the bit on most of the
    # network): a parameter's noise level is at least the network-wide median of the relative noise, and at least 1e-5 of its gradient
    rel = sorted(per_noise[n] / scale[n] for n in per_noise)
    typical = max(rel[len(rel) // 2], 1e-5)
    level = {n: max(per_noise[n], typical * scale[n]) for n in per_noise}
    loose = [n for n in per_diff if per_diff[n] > 10.0 * level[n]]
    floor = level
    assert any(float(g.abs().max()) > 0 for n, g in grads["fused"].items() if n.startswith("flownetc."))   # gradient reaches the first net through the warp
    if loose:
        pytest.xfail(f"{len(loose)} of {len(per_diff)} parameter gradients beyond 10x the unfused-vs-unfused noise (a bilinear sample "
                     f"flipped between two passes; worst {max(loose, key=lambda n: per_diff[n] / max(per_noise[n], floor[n]))}); the gross bar above held")
