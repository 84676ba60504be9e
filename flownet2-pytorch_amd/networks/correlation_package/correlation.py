"""Correlation layer: autograd Function + Module over the ``correlation_cuda`` extension.

Same public names, argument order and defaults as the reference wrapper
(networks/correlation_package/correlation.py:6-60): ``CorrelationFunction.apply(input1, input2,
pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)`` and
``Correlation(pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1)``.
The extension is the gfx950 HIP implementation (csrc/binding/correlation_cuda.cpp); importing this
module fails loudly if it has not been built.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import correlation_cuda  # built by flownet2-pytorch_amd/build.py; no fallback on purpose


class CorrelationFunction(Function):
    """out[n, tj*D+ti, y, x] = mean_c in1[n,c,y',x'] * in2[n,c,y'+tj*s2,x'+ti*s2] (reference
    correlation_cuda_kernel.cu:73-147); backward per :150-334.

    ``apply`` goes straight to the autograd node the extension implements in C++ (``correlation_cuda.apply``: no Python between
    ``apply`` and the launch, no GIL in the backward); ``forward`` / ``backward`` below are the same two calls for code written
    against the reference's static methods."""

    @classmethod
    def apply(cls, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        return correlation_cuda.apply(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)

    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1):
        ctx.save_for_backward(input1, input2)
        ctx.corr_params = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        # correlation_cuda.forward with the reference's three empty tensors (correlation.py:20-22) created and sized on the C++
        # side: one call, device guard included
        return correlation_cuda.forward_alloc(input1, input2, *ctx.corr_params)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        grad_input1, grad_input2 = correlation_cuda.backward_alloc(input1, input2, grad_output, *ctx.corr_params)
        return (grad_input1, grad_input2) + (None,) * 6


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                         self.stride1, self.stride2, self.corr_multiply)

    def extra_repr(self):
        return (f"pad_size={self.pad_size}, kernel_size={self.kernel_size}, max_displacement={self.max_displacement}, "
                f"stride1={self.stride1}, stride2={self.stride2}")


class CorrelationLeakyReLUCatFunction(Function):
    """``cat((redir, leaky_relu(corr(input1, input2), slope)), 1)`` (FlowNetC.py:86-87, :92) as one differentiable op: forward =
    the correlation kernel with the activation and the store into the concat buffer fused into its epilogue; backward = the
    gradient of the buffer's correlation slice read in place, the activation's derivative taken from the sign of the stored
    output (no mask, no saved pre-activation), the correlation backward kernels (correlation_cuda.backward_fused).
    ``apply`` = the C++ autograd node ``correlation_cuda.leakyrelu_cat_apply``; the static methods are the same calls in Python."""

    @classmethod
    def apply(cls, input1, input2, redir, pad_size, kernel_size, max_displacement, stride1, stride2, negative_slope):
        if not negative_slope > 0 and (input1.requires_grad or input2.requires_grad):
            raise ValueError("CorrelationLeakyReLUCatFunction differentiates only for negative_slope > 0")
        return correlation_cuda.leakyrelu_cat_apply(input1, input2, redir, pad_size, kernel_size, max_displacement, stride1, stride2,
                                                    float(negative_slope))

    @staticmethod
    def forward(ctx, input1, input2, redir, pad_size, kernel_size, max_displacement, stride1, stride2, negative_slope):
        if not negative_slope > 0 and (input1.requires_grad or input2.requires_grad):
            raise ValueError("CorrelationLeakyReLUCatFunction differentiates only for negative_slope > 0")
        n_out = ((max_displacement // stride2) * 2 + 1) ** 2
        B, Cr, oH, oW = redir.shape
        buf = redir.new_empty((B, Cr + n_out, oH, oW))
        buf[:, :Cr].copy_(redir)
        with torch.cuda.device_of(input1):
            correlation_cuda.forward_fused(input1, input2, buf, Cr, float(negative_slope), pad_size, kernel_size,
                                           max_displacement, stride1, stride2)
        ctx.save_for_backward(input1, input2, buf)
        ctx.corr_params = (pad_size, kernel_size, max_displacement, stride1, stride2)
        ctx.channel_offset, ctx.negative_slope = Cr, float(negative_slope)
        return buf

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_buf):
        input1, input2, buf = ctx.saved_tensors
        Cr = ctx.channel_offset
        grad_in1 = grad_in2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            with torch.cuda.device_of(input1):
                grad_in1, grad_in2 = input1.new_empty(0), input2.new_empty(0)
                correlation_cuda.backward_fused(input1, input2, buf, grad_buf, Cr, ctx.negative_slope, grad_in1, grad_in2,
                                                *ctx.corr_params)
        grad_redir = grad_buf[:, :Cr] if ctx.needs_input_grad[2] else None
        return (grad_in1, grad_in2, grad_redir) + (None,) * 6


class CorrelationLeakyReLUCat(nn.Module):
    """``torch.cat((redir, leaky_relu(corr(input1, input2), slope)), 1)`` with the activation and the concat
    fused into the correlation epilogue -- the three statements FlowNetC.py:86-87,92 as one kernel pass over the 441-channel
    cost volume instead of three --, and, in training, one pass instead of two in front of the correlation backward
    (``CorrelationLeakyReLUCatFunction``).

        cat = CorrelationLeakyReLUCat(20, 1, 20, 1, 2, negative_slope=0.1)(out_conv3a, out_conv3b, out_conv_redir)
    """

    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, negative_slope=0.1):
        super().__init__()
        # the backward reads the activation's derivative off the sign of the stored output: only an increasing activation
        # (slope > 0) keeps that sign (fn2_correlation_backward_fused rejects anything else) -- say so here, not in backward()
        if not negative_slope > 0:
            raise ValueError(f"CorrelationLeakyReLUCat needs negative_slope > 0 (got {negative_slope}); for ReLU or a negative "
                             "slope compose Correlation, the activation and torch.cat")
        self.corr_params = (pad_size, kernel_size, max_displacement, stride1, stride2)
        self.negative_slope = negative_slope

    def forward(self, input1, input2, redir):
        return CorrelationLeakyReLUCatFunction.apply(input1, input2, redir, *self.corr_params, self.negative_slope)
