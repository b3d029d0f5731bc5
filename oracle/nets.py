"""Oracle (plain PyTorch-CPU fp32 restatement) of the multi-encoder depth network.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Never imported by the product.

Restates, with identical ``state_dict`` keys so that seeded weights can be shared with the
reference modules (fixtures ``tests/golden/g4_*.npz``) and with the HIP path:

* ``manydepth/networks/pre_encoders.py:8-34``   ConvBlock      -> :class:`EncConv`
* ``manydepth/networks/pre_encoders.py:36-46``  ResidualBlock  -> :class:`EncRes`
* ``manydepth/networks/pre_encoders.py:49-83``  ShallowEncoder
* ``manydepth/networks/pre_encoders.py:85-113`` ShallowNormalsEncoder
* ``manydepth/networks/pre_encoders.py:116-164`` JointEncoder
* ``manydepth/networks/resnet_encoder.py:783-822`` ShallowResnetEncoder over torchvision
  ``resnet18`` (torchvision 0.8.2 is not installed: ``ResNet18`` below is restated from the
  published architecture -- BasicBlock [2,2,2,2], 7x7/2 stem, MaxPool(3,2,1), 1x1/2 downsample
  convs, keys ``conv1 bn1 layer{1..4}.{0,1}.{conv1,bn1,conv2,bn2,downsample.{0,1}} fc``;
  *parity unpinned by any reference fixture*, pinned by structure and key names only)
* ``manydepth/layers.py:329-380,446-449``  ConvBlock(ELU) / Conv3x3(reflect) / upsample
* ``manydepth/networks/depth_decoder.py:15-75`` DepthDecoder
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import polar

XOLP_MEAN, XOLP_STD = polar.XOLP_MEAN, polar.XOLP_STD


class EncConv(nn.Module):
    """conv(bias) -> BN -> ReLU -> [pool] -> Dropout   (pre_encoders.py:8-34)."""

    def __init__(self, cin, cout, k, down, pad, p):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=2 if down == 'stride2' else 1, padding=pad)
        self.bn = nn.BatchNorm2d(cout)
        self.dropout = nn.Dropout(p)
        self.down = down

    def forward(self, x):
        y = F.relu(self.bn(self.conv(x)))
        if self.down == 'maxpool':
            y = F.max_pool2d(y, 2)
        elif self.down == 'avgpool':
            y = F.avg_pool2d(y, 2)
        return self.dropout(y)


class EncRes(nn.Module):
    """x + conv2(conv1(x)), no activation after the add (pre_encoders.py:36-46)."""

    def __init__(self, c, k, pad, p):
        super().__init__()
        self.conv1 = EncConv(c, c, k, 'none', pad, p)
        self.conv2 = EncConv(c, c, k, 'none', pad, p)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


class ShallowEncoder(nn.Module):
    def __init__(self, mode, in_channels=2, dropout_rate=0.5):
        super().__init__()
        self.mode, p = mode, dropout_rate
        self.Conv1 = EncConv(in_channels, 64, 7, 'stride2', 3, p)
        self.ResBlock1 = EncRes(64, 3, 1, p)
        self.Conv2 = EncConv(64, 64, 5, 'maxpool', 2, p)
        self.ResBlock2 = EncRes(64, 3, 1, p)
        self.Conv3 = EncConv(64, 64, 5, 'maxpool', 2, p)
        self.ResBlock3 = EncRes(64, 3, 1, p)

    @staticmethod
    def normalizeInput(x, mode):          # pre_encoders.py:75-83
        if mode == 'XOLP':
            return (x - XOLP_MEAN) / XOLP_STD
        if mode == 'RGB':
            return (x - 0.45) / 0.225
        return x

    def forward(self, x):
        x = self.normalizeInput(x, self.mode)
        for name in ("Conv1", "ResBlock1", "Conv2", "ResBlock2", "Conv3", "ResBlock3"):
            x = getattr(self, name)(x)
        return x


class ShallowNormalsEncoder(ShallowEncoder):
    def __init__(self, in_channels=9, dropout_rate=0.1):
        super().__init__('normals', in_channels, dropout_rate)

    @staticmethod
    def get_normals(x, n=1.5):
        return polar.get_normals(x, n)

    def forward(self, x):
        return super().forward(self.get_normals(x).float())


class JointAttention(nn.Module):
    """Attention variant (BASELINE config 5): the reference branch `arch1++_attention` is not in the checkout
    (README.md:53, presentation slide 38 only name it), so this restates the build's own definition --
    **parity unpinned**:  y = x + o(softmax(q k^T / sqrt(C)) v) over the H*W tokens, q/k/v/o = 1x1 convs."""

    def __init__(self, channels=128):
        super().__init__()
        self.q, self.k, self.v, self.o = (nn.Conv2d(channels, channels, 1) for _ in range(4))

    def forward(self, x):
        N, C, H, W = x.shape
        tok = lambda t: t.flatten(2).transpose(1, 2)                 # [N, T, C]
        q, k, v = tok(self.q(x)), tok(self.k(x)), tok(self.v(x))
        p = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1)
        a = (p @ v).transpose(1, 2).reshape(N, C, H, W)
        return x + self.o(a)


class JointEncoder(nn.Module):
    def __init__(self, dropout_rate=0.0, include_normals=True, include_xolp=True, attention=False):
        super().__init__()
        extra = 64 * (int(include_normals) + int(include_xolp))
        self.attn = JointAttention(128) if attention else None
        p = dropout_rate
        self.fc1 = EncConv(128 + extra, 256, 1, 'none', 0, p)
        self.fc2 = EncConv(256, 128, 1, 'none', 0, p)
        self.ResBlock1 = EncRes(128, 3, 1, p)
        self.ResBlock2 = EncRes(128, 3, 1, p)
        self.Conv1 = EncConv(128, 256, 5, 'maxpool', 2, p)
        self.ResBlock3 = EncRes(256, 3, 1, p)
        self.ResBlock4 = EncRes(256, 3, 1, p)
        self.Conv2 = EncConv(256, 512, 5, 'maxpool', 2, p)
        self.ResBlock5 = EncRes(512, 3, 1, p)
        self.ResBlock6 = EncRes(512, 3, 1, p)

    def forward(self, rgb_feats, xolp_feats=None, normals_feats=None):
        parts = [rgb_feats] + [f for f in (xolp_feats, normals_feats) if f is not None]   # order: rgb, xolp, normals
        f = torch.cat(parts, 1) if len(parts) > 1 else rgb_feats
        f = self.fc2(self.fc1(f))
        if self.attn is not None:
            f = self.attn(f)
        f = self.ResBlock2(self.ResBlock1(f))
        a = self.ResBlock4(self.ResBlock3(self.Conv1(f)))
        b = self.ResBlock6(self.ResBlock5(self.Conv2(a)))
        return [a, b]


# ------------------------------------------------------------------ torchvision-compatible ResNet-18
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + idt)


class ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans = [64, 64, 128, 256, 512]
        for i in range(1, 5):
            self.add_module(f"layer{i}", nn.Sequential(BasicBlock(chans[i - 1], chans[i], 1 if i == 1 else 2),
                                                       BasicBlock(chans[i], chans[i], 1)))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():      # torchvision init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


class ShallowResnetEncoder(nn.Module):
    """resnet_encoder.py:783-822: conv1/bn1/relu -> f0; maxpool/layer1 -> f1; layer2 -> f2."""

    def __init__(self, num_layers=18, pretrained=False, num_input_images=1):
        super().__init__()
        assert num_layers == 18 and num_input_images == 1
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet18()

    def forward(self, img):
        e = self.encoder
        x = (img - 0.45) / 0.225
        f0 = e.relu(e.bn1(e.conv1(x)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        return [f0, f1, f2]


# ------------------------------------------------------------------ decoder
class Conv3x3(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1)
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class DecConv(nn.Module):
    """layers.py:329-342 ConvBlock: Conv3x3 + ELU."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv3x3(cin, cout)

    def forward(self, x):
        return F.elu(self.conv(x))


def upsample(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.scales, self.use_skips = list(scales), use_skips
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            convs[("upconv", i, 0)] = DecConv(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if use_skips and i > 0 else 0)
            convs[("upconv", i, 1)] = DecConv(cin, self.num_ch_dec[i])
        for s in self.scales:
            convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], num_output_channels)
        self.convs = convs
        self.decoder = nn.ModuleList(list(convs.values()))     # keys decoder.0..9 upconvs, 10.. dispconvs

    def forward(self, feats):
        out = {}
        x = feats[-1]
        for i in range(4, -1, -1):
            x = upsample(self.convs[("upconv", i, 0)](x))
            if self.use_skips and i > 0:
                x = torch.cat([x, feats[i - 1]], 1)
            x = self.convs[("upconv", i, 1)](x)
            if i in self.scales:
                out[("disp", i)] = torch.sigmoid(self.convs[("dispconv", i)](x))
        return out


class NormalsDecoder(nn.Module):
    """The `arch1++_separate_normals_dec` variant (reference README.md:54: "an additional decoder after the normals encoder.
    The decoder directly predicts normals").  Its source is not in the reference checkout; this build defines it with the
    reference's own decoder blocks (depth_decoder.py:29-67, layers.py:329-380): three times ConvBlock (reflect Conv3x3 + ELU)
    followed by bilinear x2 -- 64 -> 32 -> 16 -> 16 channels, H/8 -> H -- and a Conv3x3 head with three output channels (the raw
    normal; the loss normalises it).  ``decoder.0..2`` = the ConvBlocks in execution order, ``decoder.3`` = the head."""

    def __init__(self, num_ch_in=64, num_ch_dec=(32, 16, 16)):
        super().__init__()
        chans = [int(num_ch_in)] + [int(c) for c in num_ch_dec]
        self.decoder = nn.ModuleList([DecConv(chans[i], chans[i + 1]) for i in range(len(num_ch_dec))] + [Conv3x3(chans[-1], 3)])

    def forward(self, x):
        for blk in list(self.decoder)[:-1]:
            x = upsample(blk(x))
        return self.decoder[-1](x)


def build_models(augment_xolp=True, augment_normals=True, dropout_rate=0.1, scales=range(4), seed=0):
    """The five modules of trainer.py:192-216 with a seeded init."""
    torch.manual_seed(seed)
    m = OrderedDict()
    m["rgb_encoder"] = ShallowResnetEncoder(18, False)
    if augment_normals:
        m["normals_encoder"] = ShallowNormalsEncoder(9, dropout_rate)
    if augment_xolp:
        m["xolp_encoder"] = ShallowEncoder('XOLP', 2, dropout_rate)
    m["joint_encoder"] = JointEncoder(dropout_rate, augment_normals, augment_xolp)
    m["mono_depth"] = DepthDecoder(m["rgb_encoder"].num_ch_enc, scales)
    return m


def forward_models(models, color_aug, xolp):
    """trainer.py:503-513: the supervised single-frame forward."""
    feats = models["rgb_encoder"](color_aug.float())
    xf = models["xolp_encoder"](xolp.float()) if "xolp_encoder" in models else None
    nf = models["normals_encoder"](xolp.float()) if "normals_encoder" in models else None
    feats = list(feats) + models["joint_encoder"](feats[-1], xf, nf)
    return models["mono_depth"](feats)
