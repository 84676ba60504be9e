"""Which parameter differs when harness.FlowNet2's fused and unfused training passes disagree by more than run-to-run noise?
(tests/test_harness.py::test_flownet2_trains_through_the_fused_warp: about one process in ten.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
from harness.flownet2 import FlowNet2
from harness.train import synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(11)
net = FlowNet2().to(dev).train()
with torch.no_grad():
    for p in net.parameters():
        p.mul_(0.5)
inputs, target = synthetic_batch(2, 128, 192, dev, seed=4)
# earlier tests of the suite ran the same network in eval mode first
if len(sys.argv) > 1 and sys.argv[1] == "eval-first":
    net.eval()
    with torch.no_grad():
        net(inputs)
    net.train()
grads, outs, hooks = {}, {}, {}
for key, fused in (("warm-up", True), ("warm-up 2", False), ("fused", True), ("unfused", False), ("unfused again", False)):
    net.fused_training = fused
    net.zero_grad(set_to_none=True)
    out = net(inputs)
    (out - target).abs().mean().backward()
    grads[key] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    outs[key] = out.detach().clone()
def rel(a, b):
    return {n: float((a[n] - b[n]).abs().max()) / max(float(b[n].abs().max()), 1e-12) for n in a}
noise = rel(grads["unfused again"], grads["unfused"]); diff = rel(grads["fused"], grads["unfused"])
wn = max(noise, key=noise.get); wd = max(diff, key=diff.get)
print("outputs equal: fused/unfused %s, unfused/unfused %s" % (torch.equal(outs["fused"], outs["unfused"]), torch.equal(outs["unfused"], outs["unfused again"])))
print("noise %.2e (%s)   fused vs unfused %.2e (%s)" % (noise[wn], wn, diff[wd], wd))
top = sorted(diff, key=diff.get, reverse=True)[:6]
print("  ", [(n, "%.1e" % diff[n]) for n in top])
def l2(a, b):
    num = sum(float(((a[n] - b[n]).double() ** 2).sum()) for n in a) ** 0.5
    den = sum(float((b[n].double() ** 2).sum()) for n in a) ** 0.5
    return num / den
print("global relative L2: noise %.2e   fused vs unfused %.2e" % (l2(grads["unfused again"], grads["unfused"]), l2(grads["fused"], grads["unfused"])))
