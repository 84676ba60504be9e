"""FlowNet2's MultiScale training loss (reference losses.py:52-86, norm='L1') and its EPE metric as one
fused pass (fn2_multiscale_l1_epe) instead of five AvgPool2d passes over the target and ~35 small launches.

    criterion = MultiScaleL1()                      # startScale=4, numScales=5, l_weight=0.32, like the reference
    loss, epe = criterion(outputs, target)          # outputs: the tuple of 5 predictions FlowNetC returns in training
    loss.backward()                                 # d loss / d outputs[i]; the target gets no gradient

A single full-resolution tensor as `output` (what the models return in eval(); reference losses.py:80-83) takes the plain
PyTorch route: [mean |output - target|, EPE(output, target)]."""
import torch
from torch import nn
from torch.autograd import Function

import fn2_capi


class MultiScaleL1Function(Function):
    @staticmethod
    def forward(ctx, target, start_scale, div_flow, weights, coef, *outputs):
        # which predictions want a gradient: asked of autograd, not of the (no-grad) contiguous copies made below
        ctx.need = tuple(ctx.needs_input_grad[5:])
        outputs = [o.contiguous() for o in outputs]
        sums, grads = fn2_capi.multiscale_l1_epe(outputs, target.contiguous(), weights, start_scale, div_flow,
                                                 want_grads=any(ctx.need))
        both = (sums * coef).view(2, -1).sum(dim=1)     # [sum_i w_i mean|out_i - t_i| (losses.py:78), sum_i w_i mean||.||_2 (:77)]
        loss, epe = both[0], both[1]
        if any(ctx.need):
            ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(epe)
        return loss, epe

    @staticmethod
    def backward(ctx, grad_loss, _grad_epe):
        # d loss / d out_i for grad_loss = 1 was written by the forward pass; scale out of place, so that a second backward
        # over the same graph (retain_graph, gradient scalers) sees the saved gradients unchanged
        return (None, None, None, None, None) + tuple(g * grad_loss if n else None for g, n in zip(ctx.saved_tensors, ctx.need))


class MultiScaleL1(nn.Module):
    def __init__(self, startScale=4, numScales=5, l_weight=0.32, div_flow=0.05):
        super().__init__()
        self.startScale, self.numScales, self.div_flow = startScale, numScales, div_flow
        self.loss_weights = [l_weight / 2 ** s for s in range(numScales)]
        self.loss_labels = ["MultiScale-L1", "EPE"]
        self._coef = {}     # (device, shapes) -> w_i / N_i for the L1 sums followed by w_i / (N_i / 2) for the EPE sums

    def forward(self, output, target):
        if torch.is_tensor(output):   # eval(): one full-resolution prediction (reference losses.py:80-83)
            return [torch.abs(output - target).mean(), torch.norm(target - output, p=2, dim=1).mean()]
        assert isinstance(output, (tuple, list)) and len(output) == self.numScales
        key = (target.device, tuple(o.numel() for o in output))
        coef = self._coef.get(key)
        if coef is None:
            n = [max(o.numel(), 1) for o in output]
            vals = [w / k for w, k in zip(self.loss_weights, n)] + [w / (k / 2) for w, k in zip(self.loss_weights, n)]
            coef = self._coef[key] = torch.tensor(vals, dtype=torch.float32, device=target.device)
        loss, epe = MultiScaleL1Function.apply(target, self.startScale, self.div_flow, tuple(self.loss_weights), coef, *output)
        return [loss, epe]
