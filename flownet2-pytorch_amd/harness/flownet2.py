"""The full FlowNet2 stack (reference models.py:29-186: FlowNetC -> FlowNetS -> FlowNetS, FlowNetSD beside them, FlowNetFusion
on top) for inference, written as layer tables around this repo's layers.

Module names and parameter shapes are the reference's (``flownetc.conv1.0.weight`` ... ``flownetfusion.predict_flow0.bias``,
162 518 834 parameters), so ``FlowNet2_checkpoint`` state dicts load unchanged.  The four warp -> difference -> norm ->
concat groups (models.py:133-138, :145-150, :157-161, :170-174) use ``Resample2d`` / ``ChannelNorm``; in no-grad mode the
first two, which build the 12-channel input of the next sub-network, are the fused one-pass kernel ``WarpDiffNormCat``
(SURVEY.md 8f N2), and FlowNetC's cost volume carries its LeakyReLU + concat epilogue (N1).

``half()``: the convolution stacks run in fp16; Resample2d / ChannelNorm keep fp32 operands (the reference wraps the custom layers
in tofp32 / tofp16, models.py:44-49; its resample kernels are float-only), the cost volume of the no-grad path takes the half
features directly (one f16 MFMA per block product, fused LeakyReLU + concat, half output).
"""
import torch
from torch import nn

from networks.channelnorm_package.channelnorm import ChannelNorm
from networks.correlation_package.correlation import Correlation, CorrelationLeakyReLUCat
from networks.resample2d_package.resample2d import Resample2d, WarpDiffNorm, WarpDiffNormCat


def _conv(cin, cout, k=3, s=1):                    # submodules.conv without batch norm
    return nn.Sequential(nn.Conv2d(cin, cout, k, s, (k - 1) // 2, bias=True), nn.LeakyReLU(0.1, inplace=True))


def _deconv(cin, cout):                            # submodules.deconv
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))


def _iconv(cin, cout):                             # submodules.i_conv without batch norm
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True))


def _flow(cin):                                    # submodules.predict_flow
    return nn.Conv2d(cin, 2, 3, 1, 1, bias=True)


def _init(module):                                 # xavier weights, U(0,1) biases (FlowNetS.py:47-56 and siblings)
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                nn.init.uniform_(m.bias)
            nn.init.xavier_uniform_(m.weight)


class _Refiner(nn.Module):
    """Shared decoder of the FlowNet family: from the coarsest feature map down, predict a flow, upsample it, deconvolve the
    features and concatenate both with the encoder's skip connection.  ``levels`` lists, coarse to fine,
    (level id, deconv in, deconv out, predictor in, inter-conv in/out or None)."""

    def build_refiner(self, levels, last_id, last_pred_in, last_inter, up_bias):
        self._levels, self._last = levels, (last_id, last_inter is not None)
        for i, (lid, din, dout, pin, inter) in enumerate(levels):
            setattr(self, f"deconv{lid - 1}", _deconv(din, dout))
            setattr(self, f"predict_flow{lid}", _flow(pin))
            setattr(self, f"upsampled_flow{lid}_to_{lid - 1}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=up_bias))
            if inter is not None:
                setattr(self, f"inter_conv{lid}", _iconv(*inter))
        setattr(self, f"predict_flow{last_id}", _flow(last_pred_in))
        if last_inter is not None:
            setattr(self, f"inter_conv{last_id}", _iconv(*last_inter))

    def refine(self, feat, skips):
        flows = []
        for (lid, _, _, _, inter), skip in zip(self._levels, skips):
            has_inter = inter is not None and hasattr(self, f"inter_conv{lid}")
            flow = getattr(self, f"predict_flow{lid}")(getattr(self, f"inter_conv{lid}")(feat) if has_inter else feat)
            flows.append(flow)
            feat = torch.cat((skip, getattr(self, f"deconv{lid - 1}")(feat), getattr(self, f"upsampled_flow{lid}_to_{lid - 1}")(flow)), 1)
        lid, has_inter = self._last
        flows.append(getattr(self, f"predict_flow{lid}")(getattr(self, f"inter_conv{lid}")(feat) if has_inter else feat))
        return flows                                  # coarse ... fine


_STD_LEVELS = [(6, 1024, 512, 1024, None), (5, 1026, 256, 1026, None), (4, 770, 128, 770, None), (3, 386, 64, 386, None)]


class FlowNetS(_Refiner):                            # networks/FlowNetS.py:13-93
    def __init__(self, input_channels=12):
        super().__init__()
        for name, cin, cout, k, s in [("conv1", input_channels, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2),
                                      ("conv3_1", 256, 256, 3, 1), ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1),
                                      ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1), ("conv6", 512, 1024, 3, 2),
                                      ("conv6_1", 1024, 1024, 3, 1)]:
            setattr(self, name, _conv(cin, cout, k, s))
        self.build_refiner(_STD_LEVELS, 2, 194, None, up_bias=False)
        _init(self)

    def forward(self, x):
        c2 = self.conv2(self.conv1(x))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flows = self.refine(c6, (c5, c4, c3, c2))
        return tuple(reversed(flows)) if self.training else (flows[-1],)


class FlowNetSD(_Refiner):                           # networks/FlowNetSD.py:10-105: small displacements, inter-convolutions
    def __init__(self):
        super().__init__()
        for name, cin, cout, s in [("conv0", 6, 64, 1), ("conv1", 64, 64, 2), ("conv1_1", 64, 128, 1), ("conv2", 128, 128, 2),
                                   ("conv2_1", 128, 128, 1), ("conv3", 128, 256, 2), ("conv3_1", 256, 256, 1), ("conv4", 256, 512, 2),
                                   ("conv4_1", 512, 512, 1), ("conv5", 512, 512, 2), ("conv5_1", 512, 512, 1), ("conv6", 512, 1024, 2),
                                   ("conv6_1", 1024, 1024, 1)]:
            setattr(self, name, _conv(cin, cout, 3, s))
        # level 6 predicts straight from conv6_1; levels 5..2 from an inter-convolution of the concatenation
        levels = [(6, 1024, 512, 1024, None), (5, 1026, 256, 512, (1026, 512)), (4, 770, 128, 256, (770, 256)), (3, 386, 64, 128, (386, 128))]
        self.build_refiner(levels, 2, 64, (194, 64), up_bias=True)
        _init(self)

    def forward(self, x):
        c1 = self.conv1_1(self.conv1(self.conv0(x)))
        c2 = self.conv2_1(self.conv2(c1))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flows = self.refine(c6, (c5, c4, c3, c2))
        return tuple(reversed(flows)) if self.training else (flows[-1],)


class FlowNetFusion(_Refiner):                       # networks/FlowNetFusion.py:10-80
    def __init__(self):
        super().__init__()
        for name, cin, cout, s in [("conv0", 11, 64, 1), ("conv1", 64, 64, 2), ("conv1_1", 64, 128, 1), ("conv2", 128, 128, 2),
                                   ("conv2_1", 128, 128, 1)]:
            setattr(self, name, _conv(cin, cout, 3, s))
        levels = [(2, 128, 32, 128, None), (1, 162, 16, 32, (162, 32))]
        self.build_refiner(levels, 0, 16, (82, 16), up_bias=True)
        _init(self)

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        return self.refine(c2, (c1, c0))[-1]


class FlowNetCCore(_Refiner):                        # networks/FlowNetC.py:13-126 as FlowNet2's first block (input B x 6 x H x W)
    def __init__(self):
        super().__init__()
        for name, cin, cout, k, s in [("conv1", 3, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2),
                                      ("conv_redir", 256, 32, 1, 1), ("conv3_1", 473, 256, 3, 1), ("conv4", 256, 512, 3, 2),
                                      ("conv4_1", 512, 512, 3, 1), ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1),
                                      ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1)]:
            setattr(self, name, _conv(cin, cout, k, s))
        self.build_refiner(_STD_LEVELS, 2, 194, None, up_bias=True)
        self.corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = nn.LeakyReLU(0.1, inplace=True)
        self.corr_fused = CorrelationLeakyReLUCat(20, 1, 20, 1, 2, negative_slope=0.1)
        _init(self)

    def forward(self, x):
        b = x.shape[0]
        c2 = self.conv2(self.conv1(torch.cat((x[:, 0:3], x[:, 3:]), 0)))      # both towers in one batched pass
        c3 = self.conv3(c2)
        c3a, c3b = c3[:b], c3[b:]
        redir = self.conv_redir(c3a)
        dt = c3a.dtype
        if not torch.is_grad_enabled():
            if dt == torch.float16:      # half tensors are matrix operands as they are (csrc/correlation_f16_fwd.hip): no casts
                merged = self.corr_fused(c3a.contiguous(), c3b.contiguous(), redir)
            else:
                merged = self.corr_fused(c3a.float(), c3b.float(), redir.float()).to(dt)
        else:
            merged = torch.cat((redir, self.corr_activation(self.corr(c3a.float(), c3b.float()).to(dt))), 1)
        c3_1 = self.conv3_1(merged)
        c4 = self.conv4_1(self.conv4(c3_1))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flows = self.refine(c6, (c5, c4, c3_1, c2[:b]))
        return tuple(reversed(flows)) if self.training else (flows[-1],)


class FlowNet2(nn.Module):
    def __init__(self, rgb_max=255.0, div_flow=20.0):
        super().__init__()
        self.rgb_max, self.div_flow = float(rgb_max), float(div_flow)
        self.channelnorm = ChannelNorm()
        self.flownetc = FlowNetCCore()
        self.flownets_1 = FlowNetS()
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.flownetfusion = FlowNetFusion()
        self.resample1, self.resample2, self.resample3, self.resample4 = Resample2d(), Resample2d(), Resample2d(), Resample2d()
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.upsample2 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.upsample3 = nn.Upsample(scale_factor=4, mode="nearest")
        self.upsample4 = nn.Upsample(scale_factor=4, mode="nearest")
        self.warp_cat = WarpDiffNormCat(div_flow=self.div_flow)
        self.warp_err = WarpDiffNorm()
        self.fused_training = True

    def _warp_concat(self, x, flow, resample):
        """models.py:133-138: cat(x, warped second image, flow / div_flow, ||first image - warped||) -- one kernel forward, one
        kernel backward (WarpDiffNormCat is differentiable since round 5); `fused_training = False` composes the unfused
        layers under autograd as rounds 1-4 did."""
        dt = x.dtype
        if not torch.is_grad_enabled() or self.fused_training:
            return self.warp_cat(x.float(), flow.float()).to(dt)
        warped = resample(x[:, 3:].float(), flow.float())
        norm = self.channelnorm(x[:, :3].float() - warped)
        return torch.cat((x.float(), warped, flow.float() / self.div_flow, norm), 1).to(dt)

    def _warp_error(self, x, flow, resample):
        """models.py:157-161 / :170-174: (||flow||, ||first image - second image warped by flow||) -- the second as one kernel
        (WarpDiffNorm, differentiable w.r.t. the flow); `fused_training = False` composes the separate layers."""
        f = flow.float()
        if not torch.is_grad_enabled() or self.fused_training:
            return self.channelnorm(f).to(x.dtype), self.warp_err(x.float(), f).to(x.dtype)
        warped = resample(x[:, 3:].float(), f)
        return self.channelnorm(f).to(x.dtype), self.channelnorm(x[:, :3].float() - warped).to(x.dtype)

    def forward(self, inputs):
        mean = inputs.reshape(inputs.shape[0], inputs.shape[1], -1).mean(dim=-1).view(inputs.shape[0], inputs.shape[1], 1, 1, 1)
        x = (inputs - mean) / self.rgb_max
        x = torch.cat((x[:, :, 0], x[:, :, 1]), 1)                               # B x 6 x H x W
        flow_c = self.upsample1(self.flownetc(x)[0] * self.div_flow)
        flow_s1 = self.upsample2(self.flownets_1(self._warp_concat(x, flow_c, self.resample1))[0] * self.div_flow)
        flow_s2 = self.upsample4(self.flownets_2(self._warp_concat(x, flow_s1, self.resample2))[0] * self.div_flow)
        norm_s2, err_s2 = self._warp_error(x, flow_s2, self.resample4)
        flow_sd = self.upsample3(self.flownets_d(x)[0] / self.div_flow)
        norm_sd, err_sd = self._warp_error(x, flow_sd, self.resample3)
        return self.flownetfusion(torch.cat((x[:, :3], flow_sd, flow_s2, norm_sd, norm_s2, err_sd, err_s2), 1))
