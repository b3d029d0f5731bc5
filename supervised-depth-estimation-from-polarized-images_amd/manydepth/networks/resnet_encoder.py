"""ShallowResnetEncoder (reference manydepth/networks/resnet_encoder.py:783-822) on HIP kernels.

The module tree reproduces torchvision's ``resnet18`` names (conv1, bn1, layer1..4.{0,1}.{conv1,bn1,
conv2,bn2,downsample.{0,1}}, fc) so that ``rgb_encoder.pth`` checkpoints round-trip, including the
layer3 / layer4 / fc parameters the reference constructs but never runs (:819-820).
``pretrained=True`` would download ImageNet weights in the reference (options.py:261-265); there is
no network here: weights are loaded from $PD_RESNET18_WEIGHTS if set, else scratch with a warning.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from polardepth import functional as PF


def _cl(conv):
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = _cl(nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _cl(nn.Conv2d(planes, planes, 3, 1, 1, bias=False))
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(_cl(nn.Conv2d(inplanes, planes, 1, stride, bias=False)),
                                            nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):
        tr = self.training
        # identity blocks: conv1 and the skip read the same x -- the skip gradient rides in conv1's data-gradient epilogue
        mail = PF.SkipGrad() if (self.downsample is None and torch.is_grad_enabled() and x.requires_grad) else None
        y = PF.conv_bn_chain(x, self.conv1, self.bn1, PF.ChainCfg(stride=self.stride, pad=1, relu_pre=True, skip_in=mail),
                             training=tr)
        idt = x
        if self.downsample is not None:
            idt = PF.conv_bn_chain(x, self.downsample[0], self.downsample[1],
                                   PF.ChainCfg(stride=self.stride, pad=0, relu_pre=False), training=tr)
        # bn2 -> + identity -> relu
        return PF.conv_bn_chain(y, self.conv2, self.bn2,
                                PF.ChainCfg(stride=1, pad=1, relu_pre=False, relu_post=True, skip_out=mail),
                                res=idt, training=tr)


class ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = _cl(nn.Conv2d(3, 64, 7, 2, 3, bias=False))
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans = [64, 64, 128, 256, 512]
        for i in range(1, 5):
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(chans[i - 1], chans[i], 1 if i == 1 else 2),
                                                     BasicBlock(chans[i], chans[i], 1)))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


class ShallowResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1, **kwargs):
        super().__init__()
        if num_layers != 18:
            raise ValueError("{} is not a supported number of resnet layers (the hot path uses 18)".format(num_layers))
        if num_input_images != 1:
            raise NotImplementedError("multi-image input belongs to the self-supervised path (out of scope)")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet18()
        if pretrained:
            path = os.environ.get("PD_RESNET18_WEIGHTS")
            if path and os.path.exists(path):
                self.encoder.load_state_dict(torch.load(path, map_location="cpu"))
            else:
                warnings.warn("weights_init=pretrained: no local resnet18 weights ($PD_RESNET18_WEIGHTS) and no "
                              "network access -- training the RGB encoder from scratch")

    def forward(self, input_image):
        e = self.encoder
        tr = self.training
        self.features = []
        # (x - 0.45) / 0.225 is folded into the stem's gather (resnet_encoder.py:812)
        cfg = PF.ChainCfg(stride=2, pad=3, relu_pre=True, affine=(0.45, 0.225))
        f0 = PF.conv_bn_chain(input_image.float(), e.conv1, e.bn1, cfg, training=tr)
        self.features.append(f0)
        # f0 feeds the max-pool and, as a skip connection, the decoder: the decoder deposits its gradient in this mailbox and
        # the max-pool's backward kernel adds it (functional.SkipGrad), instead of autograd's pass over three 335 MB tensors
        mail = PF.SkipGrad() if (PF.USE_SKIP_FUSION and torch.is_grad_enabled() and f0.requires_grad) else None
        if mail is not None:
            f0._pd_skip_mail = mail
        x = PF.maxpool3s2(f0, mail)
        for blk in e.layer1:
            x = blk(x)
        self.features.append(x)
        for blk in e.layer2:
            x = blk(x)
        self.features.append(x)
        return self.features
