"""XOLP / normals / joint encoders (reference manydepth/networks/pre_encoders.py) on HIP kernels.

Same classes, constructor signatures and state_dict keys (``Conv1.conv.weight``, ``ResBlock1.conv2.bn.
running_var`` ...).  Each ConvBlock is two kernels forward (implicit-GEMM conv with a BatchNorm
statistics epilogue, then BN-apply + ReLU + pool + dropout [+ residual]) instead of five ATen ops.
Tensors are logically NCHW and physically NHWC (torch.channels_last).
"""
import numpy as np
import torch
import torch.nn as nn

from polardepth import functional as PF
from ..normals_vec import get_normals as _get_normals

XOLP_MEAN, XOLP_STD = 0.08693199701957657, 0.44430732785457433


def _channels_last_(conv):
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class ConvBlock(nn.Module):
    """conv(bias) -> BatchNorm -> ReLU -> [2x2 max-pool] -> Dropout   (pre_encoders.py:8-34)."""

    def __init__(self, in_channels, out_channels, kernel_size, downsampling_mode, padding, dropout_p):
        super().__init__()
        if downsampling_mode == 'avgpool':
            raise NotImplementedError("avgpool down-sampling is not used by the reference networks")
        stride = 2 if downsampling_mode == 'stride2' else 1
        self.conv = _channels_last_(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding))
        self.nonlin = nn.ReLU(inplace=True)
        self.bn = nn.BatchNorm2d(out_channels)
        self.dropout = nn.Dropout(p=dropout_p)
        self.downsampling_mode = downsampling_mode
        self.stride, self.padding = stride, padding
        self.in_affine = None          # (sub, div) folded into the conv gather (ShallowEncoder.normalizeInput)

    def forward(self, x, res=None, skip_in=None, skip_out=None):
        cfg = PF.ChainCfg(stride=self.stride, pad=self.padding, relu_pre=True,
                          pool=self.downsampling_mode == 'maxpool', drop_p=self.dropout.p, relu_post=False,
                          affine=self.in_affine, skip_in=skip_in, skip_out=skip_out)
        return PF.conv_bn_chain(x, self.conv, self.bn, cfg, res=res, training=self.training)


class ResidualBlock(nn.Module):
    """conv2(conv1(x)) + x, no activation after the add (pre_encoders.py:36-46); the add is fused."""

    def __init__(self, channels, kernel_size, padding, dropout):
        super().__init__()
        self.conv1 = ConvBlock(channels, channels, kernel_size, 'none', padding, dropout)
        self.conv2 = ConvBlock(channels, channels, kernel_size, 'none', padding, dropout)

    def forward(self, x):
        # both convolutions read x: the skip gradient rides in the first convolution's data-gradient epilogue
        mail = PF.SkipGrad() if (torch.is_grad_enabled() and x.requires_grad) else None
        return self.conv2(self.conv1(x, skip_in=mail), res=x, skip_out=mail)


class ShallowEncoder(nn.Module):
    def __init__(self, mode, in_channels=2, dropout_rate=0.5):
        super().__init__()
        self.in_channels, self.mode = in_channels, mode
        self.Conv1 = ConvBlock(in_channels, 64, 7, 'stride2', 3, dropout_rate)
        self.ResBlock1 = ResidualBlock(64, 3, 1, dropout_rate)
        self.Conv2 = ConvBlock(64, 64, 5, 'maxpool', 2, dropout_rate)
        self.ResBlock2 = ResidualBlock(64, 3, 1, dropout_rate)
        self.Conv3 = ConvBlock(64, 64, 5, 'maxpool', 2, dropout_rate)
        self.ResBlock3 = ResidualBlock(64, 3, 1, dropout_rate)
        # normalizeInput (:75-83) is applied inside Conv1's gather: in-bounds taps become (x-m)/s,
        # padded taps stay 0 -- identical to normalising first and zero-padding after.
        self.Conv1.in_affine = {'XOLP': (XOLP_MEAN, XOLP_STD), 'RGB': (0.45, 0.225)}.get(mode)

    def forward(self, x):
        out = self.Conv1(x.float())
        out = self.ResBlock1(out)
        out = self.Conv2(out)
        out = self.ResBlock2(out)
        out = self.Conv3(out)
        return self.ResBlock3(out)

    @staticmethod
    def normalizeInput(x, mode):
        if mode == 'XOLP':
            return (x - XOLP_MEAN) / XOLP_STD
        if mode == 'normals':
            return x
        if mode == 'RGB':
            return (x - 0.45) / 0.225


class ShallowNormalsEncoder(ShallowEncoder):
    def __init__(self, in_channels=9, dropout_rate=0.1):
        super().__init__('normals', in_channels, dropout_rate)

    def forward(self, x, normals=None):
        """x: XOLP [B,2,H,W].  ``normals`` may carry the 9 channels already produced by the fused
        polar kernel (K1) from the raw planes; otherwise they are computed from x on the GPU."""
        if normals is None:
            normals = self.get_normals(x)
        return super().forward(normals)

    @staticmethod
    def get_normals(x, n=1.5):
        return _get_normals(x, n)


class JointAttention(nn.Module):
    """Single-head softmax self-attention + residual over the H/8 x W/8 token grid after the modality merge
    (BASELINE config 5 "arch1++_attention"; README.md:53 / presentation slide 38 of the reference describe it, the
    branch itself is not in the checkout, so this block is the build's definition -- DESIGN.md):
        y = x + o(softmax(q(x) k(x)^T / sqrt(C)) v(x)),   q, k, v, o = 1x1 convolutions with bias."""

    def __init__(self, channels=128):
        super().__init__()
        mk = lambda: _channels_last_(nn.Conv2d(channels, channels, 1))
        self.q, self.k, self.v, self.o = mk(), mk(), mk(), mk()

    def forward(self, x):
        a = PF.self_attention(PF.conv_bias(x, self.q), PF.conv_bias(x, self.k), PF.conv_bias(x, self.v))
        return x + PF.conv_bias(a, self.o)


class JointEncoder(nn.Module):
    def __init__(self, dropout_rate=0.0, include_normals=True, include_xolp=True, attention=False):
        super().__init__()
        additional_ch = 64 * (int(include_normals) + int(include_xolp))
        self.attn = JointAttention(128) if attention else None
        self.fc1 = ConvBlock(128 + additional_ch, 256, 1, 'none', 0, dropout_rate)
        self.fc2 = ConvBlock(256, 128, 1, 'none', 0, dropout_rate)
        self.ResBlock1 = ResidualBlock(128, 3, 1, dropout_rate)
        self.ResBlock2 = ResidualBlock(128, 3, 1, dropout_rate)
        self.Conv1 = ConvBlock(128, 256, 5, 'maxpool', 2, dropout_rate)
        self.ResBlock3 = ResidualBlock(256, 3, 1, dropout_rate)
        self.ResBlock4 = ResidualBlock(256, 3, 1, dropout_rate)
        self.Conv2 = ConvBlock(256, 512, 5, 'maxpool', 2, dropout_rate)
        self.ResBlock5 = ResidualBlock(512, 3, 1, dropout_rate)
        self.ResBlock6 = ResidualBlock(512, 3, 1, dropout_rate)

    def forward(self, rgb_feats, xolp_feats=None, normals_feats=None):
        parts = [rgb_feats] + [f for f in (xolp_feats, normals_feats) if f is not None]   # rgb, xolp, normals (:142-151)
        feats = torch.cat(parts, dim=1) if len(parts) > 1 else rgb_feats
        feats = self.fc2(self.fc1(feats))
        if self.attn is not None:
            feats = self.attn(feats)
        feats = self.ResBlock2(self.ResBlock1(feats))
        a = self.ResBlock4(self.ResBlock3(self.Conv1(feats)))
        b = self.ResBlock6(self.ResBlock5(self.Conv2(a)))
        return [a, b]
