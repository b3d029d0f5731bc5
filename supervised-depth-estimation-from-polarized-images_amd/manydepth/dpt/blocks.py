"""Decoder blocks of DPT (reference manydepth/dpt/blocks.py:138-172, 255-383) on the HIP kernels: same class names, constructor
arguments and state_dict keys (``resConfUnit{1,2}.conv{1,2}.{weight,bias}``, ``out_conv.{weight,bias}``).

ResidualConvUnit_custom = relu -> conv3x3 -> relu -> conv3x3 -> + x: the first ReLU is one element-wise launch, the second
rides in the first convolution's epilogue, the skip sum is one launch.  FeatureFusionBlock_custom = [x0 + RCU1(x1)] -> RCU2 ->
bilinear x2 with align_corners=True (pd_up2x_ac_fwd) -> 1x1 convolution.  Only the configuration dpt/models.py:15-23 builds is
supported (activation ReLU, no BatchNorm, no deconv / expand); anything else raises."""
import torch
import torch.nn as nn

from polardepth import functional as PF
from polardepth import ops


class Interpolate(nn.Module):
    """blocks.py:138-172; the instances of the reference are (scale_factor=2, mode="bilinear", align_corners=True)."""

    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        if not (scale_factor == 2 and mode == "bilinear" and align_corners):
            raise NotImplementedError("Interpolate: only scale_factor=2, mode='bilinear', align_corners=True is built "
                                      "(the reference's DPT decoder uses nothing else)")
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners

    def forward(self, x):
        return PF.upsample2x_aligned(x)


def _conv3x3(features):
    c = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
    c.weight.data = c.weight.data.contiguous(memory_format=torch.channels_last)
    return c


class ResidualConvUnit_custom(nn.Module):
    """blocks.py:255-315."""

    def __init__(self, features, activation, bn):
        super().__init__()
        if bn or not isinstance(activation, nn.ReLU):
            raise NotImplementedError("ResidualConvUnit_custom: built for activation=nn.ReLU, bn=False (dpt/models.py:15-23)")
        self.bn, self.groups = bn, 1
        self.conv1, self.conv2 = _conv3x3(features), _conv3x3(features)
        self.activation = activation

    def forward(self, x):
        out = PF.conv_bias_act(PF.relu(x), self.conv1, ops.ACT_RELU)         # conv1(relu(x)), then the second ReLU
        return PF.add(PF.conv_bias_act(out, self.conv2), x)


class FeatureFusionBlock_custom(nn.Module):
    """blocks.py:318-383."""

    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True):
        super().__init__()
        if deconv or expand or not align_corners:
            raise NotImplementedError("FeatureFusionBlock_custom: built for deconv=False, expand=False, align_corners=True")
        self.deconv, self.align_corners, self.groups, self.expand = deconv, align_corners, 1, expand
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.out_conv.weight.data = self.out_conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)

    def forward(self, *xs):
        output = xs[0]
        if len(xs) == 2:
            output = PF.add(output, self.resConfUnit1(xs[1]))
        output = PF.upsample2x_aligned(self.resConfUnit2(output))
        return PF.conv_bias_act(output, self.out_conv)
