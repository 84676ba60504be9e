"""ChannelNorm (per-pixel L2 norm over channels): autograd Function + Module over the
``channelnorm_cuda`` extension.

Same public names and signatures as the reference wrapper
(networks/channelnorm_package/channelnorm.py:5-38): ``ChannelNormFunction.apply(input1, norm_deg)``
and ``ChannelNorm(norm_deg=2)``.  ``norm_deg`` is stored and passed on but, as in the reference
kernels, only the L2 norm is implemented.  The backward honours grad_output's strides (the
reference reads the non-contiguous slice autograd hands it as if it were contiguous).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import channelnorm_cuda  # built by flownet2-pytorch_amd/build.py; no fallback on purpose


class ChannelNormFunction(Function):
    """``apply`` = the C++ autograd node ``channelnorm_cuda.apply``; ``forward`` / ``backward`` are the same two calls in Python."""

    @classmethod
    def apply(cls, input1, norm_deg=2):
        assert input1.is_contiguous(), "input1 must be contiguous (reference channelnorm.py:9)"
        return channelnorm_cuda.apply(input1, norm_deg)

    @staticmethod
    def forward(ctx, input1, norm_deg=2):
        assert input1.is_contiguous(), "input1 must be contiguous (reference channelnorm.py:9)"
        output = channelnorm_cuda.forward_alloc(input1, norm_deg)   # (B, 1, H, W), allocated on the C++ side, fully written
        ctx.save_for_backward(input1, output)
        ctx.norm_deg = norm_deg
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, output = ctx.saved_tensors
        return channelnorm_cuda.backward_alloc(input1, output, grad_output, ctx.norm_deg), None


class ChannelNorm(nn.Module):
    def __init__(self, norm_deg=2):
        super().__init__()
        self.norm_deg = norm_deg

    def forward(self, input1):
        return ChannelNormFunction.apply(input1, self.norm_deg)
