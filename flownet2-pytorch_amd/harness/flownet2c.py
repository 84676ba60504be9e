"""FlowNet2C (reference models.py:187-253 on top of networks/FlowNetC.py:13-126), written as a layer table.

Module names and parameter shapes are the reference's, so a reference checkpoint's ``state_dict`` loads unchanged
(``conv1.0.weight`` ... ``upsampled_flow3_to_2.bias``, 39 175 298 parameters; tests/test_harness.py compares the key set
with the reference's own class where the reference checkout is present).  The cost volume is this repo's
``Correlation`` (HIP kernels, fp32); the LeakyReLU + concat around it are fused into the correlation's epilogue
(``CorrelationLeakyReLUCat``) -- in training too since round 4: the backward reads the activation's derivative
off the sign of the stored output and the gradient from its slice of the concat gradient (``fused_training=False`` keeps the
three separate ops of FlowNetC.py:86-92).
"""
import torch
from torch import nn

from networks.correlation_package.correlation import Correlation, CorrelationLeakyReLUCat

# name, in, out, kernel, stride                      (FlowNetC.py:20-23, :35-41)
_CONVS = [("conv1", 3, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2), ("conv_redir", 256, 32, 1, 1),
          ("conv3_1", 473, 256, 3, 1), ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1), ("conv5", 512, 512, 3, 2),
          ("conv5_1", 512, 512, 3, 1), ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1)]
# refinement level: (deconv name, in, out), (flow predictor name, in), (flow upsampler name)        (FlowNetC.py:43-57)
_DECONVS = [("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128), ("deconv2", 386, 64)]
_PREDICT = [("predict_flow6", 1024), ("predict_flow5", 1026), ("predict_flow4", 770), ("predict_flow3", 386), ("predict_flow2", 194)]
_UPFLOW = ["upsampled_flow6_to_5", "upsampled_flow5_to_4", "upsampled_flow4_to_3", "upsampled_flow3_to_2"]


def _act():
    return nn.LeakyReLU(0.1, inplace=True)


class FlowNet2C(nn.Module):
    def __init__(self, rgb_max=255.0, div_flow=20.0, batch_norm=False, fused_inference=True, fused_training=True):
        super().__init__()
        self.rgb_max, self.div_flow, self.fused_inference = float(rgb_max), float(div_flow), fused_inference
        self.fused_training = fused_training
        for name, cin, cout, k, s in _CONVS:       # submodules.conv: Conv2d (+ BatchNorm2d) + LeakyReLU(0.1), "same" padding
            layers = [nn.Conv2d(cin, cout, k, s, (k - 1) // 2, bias=not batch_norm)]
            if batch_norm:
                layers.append(nn.BatchNorm2d(cout))
            setattr(self, name, nn.Sequential(*layers, _act()))
        for name, cin, cout in _DECONVS:           # submodules.deconv
            setattr(self, name, nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), _act()))
        for name, cin in _PREDICT:                 # submodules.predict_flow
            setattr(self, name, nn.Conv2d(cin, 2, 3, 1, 1, bias=True))
        for name in _UPFLOW:
            setattr(self, name, nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
        self.corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = _act()
        self.corr_fused = CorrelationLeakyReLUCat(20, 1, 20, 1, 2, negative_slope=0.1)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        for m in self.modules():                   # FlowNetC.py:58-67: xavier weights, U(0,1) biases
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)

    def features(self, x1, x2):
        """The two siamese towers up to conv3 (FlowNetC.py:73-82); both images in one batched pass."""
        b = x1.shape[0]
        c1 = self.conv1(torch.cat((x1, x2), 0))
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        return c2[:b], c3[:b], c3[b:]

    def merge(self, c3a, c3b):
        """Cost volume, LeakyReLU, concat with the redirected features (FlowNetC.py:85-92)."""
        redir = self.conv_redir(c3a)
        if self.fused_inference if not torch.is_grad_enabled() else self.fused_training:
            return self.corr_fused(c3a, c3b, redir)
        return torch.cat((redir, self.corr_activation(self.corr(c3a, c3b))), 1)

    def forward(self, inputs):
        # models.py:193-197: per-sample, per-channel mean over both frames; inputs B x 3 x 2 x H x W in [0, rgb_max]
        mean = inputs.reshape(inputs.shape[0], inputs.shape[1], -1).mean(dim=-1).view(inputs.shape[0], inputs.shape[1], 1, 1, 1)
        x = (inputs - mean) / self.rgb_max
        c2a, c3a, c3b = self.features(x[:, :, 0], x[:, :, 1])
        c3_1 = self.conv3_1(self.merge(c3a, c3b))
        c4 = self.conv4_1(self.conv4(c3_1))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        # refinement (FlowNetC.py:101-121): predict, upsample the flow, deconvolve the features, concat with the skip
        feat, flows = c6, []
        for level, skip in enumerate((c5, c4, c3_1, c2a)):
            flow = getattr(self, _PREDICT[level][0])(feat)
            flows.append(flow)
            feat = torch.cat((skip, getattr(self, _DECONVS[level][0])(feat), getattr(self, _UPFLOW[level])(flow)), 1)
        flows.append(self.predict_flow2(feat))
        if self.training:
            return tuple(reversed(flows))            # flow2, flow3, flow4, flow5, flow6
        return self.upsample1(flows[-1] * self.div_flow)
