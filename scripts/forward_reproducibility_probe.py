"""Which op makes harness.FlowNet2's training forward differ from pass to pass?  ("find and name the op".)
Forward hooks on every leaf module record a checksum of each output; two identical passes (same weights, same inputs, fused rows
replaced by the unfused layers or not) are compared module by module, in execution order.  The first module whose output differs is
the source; everything downstream differs as a consequence.  Run on the GPU box; prints a short report."""
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


def one_pass(fused, backward):
    log = []
    hooks = []
    for name, m in net.named_modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(lambda mod, a, out, name=name: log.append((name, type(mod).__name__, out.detach().clone() if torch.is_tensor(out) else None))))
    net.fused_training = fused
    net.zero_grad(set_to_none=True)
    out = net(inputs)
    if backward:
        (out - target).abs().mean().backward()
    for h in hooks:
        h.remove()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None} if backward else {}
    return log, out.detach().clone(), grads


for fused in (False, True):
    one_pass(fused, True)          # MIOpen picks its algorithms on the first calls of a shape
for fused in (False, True):
    runs = [one_pass(fused, True) for _ in range(6)]
    base = runs[0]
    print(f"--- fused_training={fused}: 6 identical training passes, compared with the first")
    for k, (log, out, grads) in enumerate(runs[1:], 1):
        first = next(((i, n, t) for i, ((n, t, a), (_, _, b)) in enumerate(zip(log, base[0])) if a is not None and not torch.equal(a, b)), None)
        ndiff = sum(1 for (n, t, a), (_, _, b) in zip(log, base[0]) if a is not None and not torch.equal(a, b))
        gdiff = [n for n in grads if not torch.equal(grads[n], base[2][n])]
        print(f"pass {k}: forward outputs equal: {torch.equal(out, base[1])}; modules whose output differs: {ndiff} of {len(log)}"
              + (f"; first: #{first[0]} {first[1]} ({first[2]})" if first else "")
              + f"; parameter gradients that differ: {len(gdiff)} of {len(grads)}")
    if all(torch.equal(r[1], base[1]) for r in runs[1:]):
        # the forward IS reproducible: then the backward is what varies -- name the parameter gradients that do
        names = sorted({n for r in runs[1:] for n in r[2] if not torch.equal(r[2][n], base[2][n])})
        print("forward bit-reproducible over 6 passes; gradients that vary between passes:", len(names), "e.g.", names[:6])
        worst = max(((float((r[2][n] - base[2][n]).abs().max()) / max(float(base[2][n].abs().max()), 1e-30), n) for r in runs[1:] for n in names), default=(0.0, "-"))
        print("largest relative difference of a parameter gradient between two passes: %.2e (%s)" % worst)
