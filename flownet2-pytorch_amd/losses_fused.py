"""FlowNet2's MultiScale training loss (reference losses.py:52-86, norm 'L1' or 'L2') and its EPE metric as one
fused pass (fn2_multiscale_loss) instead of five AvgPool2d passes over the target and ~35 small launches.

    criterion = MultiScale(args)                    # startScale=4, numScales=5, l_weight=0.32, norm='L1', like the reference
    loss, epe = criterion(outputs, target)          # outputs: the tuple of 5 predictions FlowNetC returns in training
    loss.backward()                                 # d loss / d outputs[i]; the target gets no gradient

Class names, constructor arguments and return values follow losses.py (MultiScale :52-86, L1Loss :28-38, L2Loss :40-50, EPE :11-12)
so that `--loss=MultiScale` style code finds the same objects.  A single full-resolution tensor as `output` (what the models
return in eval(); losses.py:80-83) takes the plain PyTorch route.  Pinned against the reference's own classes by
tests/test_losses_pin.py (CPU) and tests/golden/multiscale_*.npz (GPU)."""
import torch
from torch import nn
from torch.autograd import Function

import fn2_capi
import multiscale_loss_cuda  # built by flownet2-pytorch_amd/build.py (csrc/binding/multiscale_loss_cuda.cpp); no fallback on purpose


def EPE(input_flow, target_flow):
    """losses.py:11-12."""
    return torch.norm(target_flow - input_flow, p=2, dim=1).mean()


class MultiScaleFunction(Function):
    """The node in Python over the ctypes view of the C ABI (rounds 3-5).  Since round 6 GPU tensors take the C++ node
    ``multiscale_loss_cuda.apply`` instead (one launch forward -- loss and metric written by the kernel --, one launch backward, no
    host arrays marshalled per call); this class remains for callers of its static methods and as the specification the CPU pin test
    drives with the kernel call replaced (tests/test_losses_pin.py)."""

    @staticmethod
    def forward(ctx, target, start_scale, div_flow, weights, coef, norm, *outputs):
        # which predictions want a gradient: asked of autograd, not of the (no-grad) contiguous copies made below
        ctx.need = tuple(ctx.needs_input_grad[6:])
        outputs = [o.contiguous() for o in outputs]
        sums, grads = fn2_capi.multiscale_l1_epe(outputs, target.contiguous(), weights, start_scale, div_flow,
                                                 want_grads=any(ctx.need), norm=norm)
        both = (sums * coef).view(2, -1).sum(dim=1)     # [sum_i w_i mean|out_i - t_i| (losses.py:78), sum_i w_i mean||.||_2 (:77)]
        loss, epe = (both[0], both[1]) if norm == 1 else (both[1].clone(), both[1])   # L2(): the expression of EPE (:21-26)
        if any(ctx.need):
            ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(epe)
        return loss, epe

    @staticmethod
    def backward(ctx, grad_loss, _grad_epe):
        # d loss / d out_i for grad_loss = 1 was written by the forward pass; scale out of place, so that a second backward
        # over the same graph (retain_graph, gradient scalers) sees the saved gradients unchanged
        return (None,) * 6 + tuple(g * grad_loss if n else None for g, n in zip(ctx.saved_tensors, ctx.need))


class MultiScale(nn.Module):
    """losses.py:52-86.  `args` is stored and unused, as in the reference."""

    def __init__(self, args=None, startScale=4, numScales=5, l_weight=0.32, norm='L1'):
        super().__init__()
        self.args = args
        self.startScale, self.numScales, self.div_flow = startScale, numScales, 0.05   # (:60)
        self.loss_weights = [l_weight / 2 ** s for s in range(numScales)]               # (:57)
        self.l_type = norm
        self.loss_labels = ['MultiScale-' + self.l_type, 'EPE']
        self._norm = 1 if norm == 'L1' else 2        # (:62-65: anything but 'L1' selects L2())
        self._coef = {}     # (device, shapes) -> w_i / N_i for the L1 sums followed by w_i / (N_i / 2) for the 2-norm sums

    def forward(self, output, target):
        if torch.is_tensor(output):   # eval(): one full-resolution prediction (losses.py:80-83)
            epe = EPE(output, target)
            return [torch.abs(output - target).mean() if self._norm == 1 else torch.norm(output - target, p=2, dim=1).mean(), epe]
        assert isinstance(output, (tuple, list)) and len(output) == self.numScales
        if target.is_cuda:
            loss, epe = multiscale_loss_cuda.apply(target, list(output), self.startScale, self.div_flow, self.loss_weights, self._norm)
            return [loss, epe]
        key = (target.device, tuple(o.numel() for o in output))
        coef = self._coef.get(key)
        if coef is None:
            n = [max(o.numel(), 1) for o in output]
            vals = [w / k for w, k in zip(self.loss_weights, n)] + [w / (k / 2) for w, k in zip(self.loss_weights, n)]
            coef = self._coef[key] = torch.tensor(vals, dtype=torch.float32, device=target.device)
        loss, epe = MultiScaleFunction.apply(target, self.startScale, self.div_flow, tuple(self.loss_weights), coef, self._norm, *output)
        return [loss, epe]


class MultiScaleL1(MultiScale):
    """MultiScale(norm='L1') with the keyword arguments of rounds 1-4 (no `args`, settable div_flow)."""

    def __init__(self, startScale=4, numScales=5, l_weight=0.32, div_flow=0.05):
        super().__init__(None, startScale, numScales, l_weight, 'L1')
        self.div_flow = div_flow


class L1Loss(nn.Module):
    """losses.py:28-38."""

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.loss_labels = ['L1', 'EPE']

    def forward(self, output, target):
        return [torch.abs(output - target).mean(), EPE(output, target)]


class L2Loss(nn.Module):
    """losses.py:40-50."""

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.loss_labels = ['L2', 'EPE']

    def forward(self, output, target):
        return [torch.norm(output - target, p=2, dim=1).mean(), EPE(output, target)]
