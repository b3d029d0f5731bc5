"""DepthDecoder (reference manydepth/networks/depth_decoder.py:15-75) on HIP kernels.

Per level: upconv(i,0) [reflect conv + ELU, one kernel] -> bilinear x2 + skip concat [one kernel]
-> upconv(i,1) -> dispconv + sigmoid [one kernel].  ModuleList order and keys (``decoder.0..9``
upconvs, ``decoder.10..13`` dispconvs) follow depth_decoder.py:29-53.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from manydepth.layers import ConvBlock, Conv3x3, Conv5x5
from polardepth import functional as PF
from polardepth import ops


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, uncertainty=False):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = 'nearest'      # attribute kept; like the reference it is unused (bilinear is applied)
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                cin += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(cin, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.uncertainty = bool(uncertainty)
        if self.uncertainty:            # depth_decoder.py:46-50: two 5x5 heads per scale, ModuleList entries 14.. in this order
            for s in self.scales:
                self.convs[("unc_conv", s)] = Conv5x5(self.num_ch_dec[s], self.num_output_channels)
                self.convs[("unc_conv_color", s)] = Conv5x5(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        self.outputs = {}
        x = input_features[-1]
        # ActGrad mailboxes: the ELU derivative of every ConvBlock is applied by the kernel that produces its output
        # gradient (upsample gradient / disparity head), and the head sums the two gradients of upconv(i,1)'s output
        # (with the uncertainty heads upconv(i,1)'s output has four consumers: plain autograd routes, off the supervised path)
        fuse = PF.USE_ACT_FUSION and torch.is_grad_enabled() and not self.uncertainty
        pend = None
        for i in range(4, -1, -1):
            m0 = PF.ActGrad() if fuse else None
            x = self.convs[("upconv", i, 0)](x, act_mail=m0, dx_mail=pend)
            skip = input_features[i - 1] if self.use_skips and i > 0 else None
            x = PF.upcat(x, skip, act_mail=m0)
            m1 = PF.ActGrad(expect_deposit=i > 0) if fuse and i in self.scales else None
            x = self.convs[("upconv", i, 1)](x, act_mail=m1)
            if i in self.scales:
                self.outputs[("disp", i)] = self.convs[("dispconv", i)](x, act=ops.ACT_SIGMOID, head_mail=m1)
                if self.uncertainty:     # depth_decoder.py:71-73: sigmoid(Conv5x5(x)), the sigmoid in the conv epilogue
                    self.outputs[("uncertainty", i)] = self.convs[("unc_conv", i)](x, act=ops.ACT_SIGMOID)
                    self.outputs[("uncertainty_color", i)] = self.convs[("unc_conv_color", i)](x, act=ops.ACT_SIGMOID)
            pend = m1
        if fuse and len(self.outputs) > 1:
            # every head an ancestor of every output: a backward pass from a subset of the scales still runs all heads,
            # which collect the gradients deposited for them (functional.JoinHeadsFn)
            keys = list(self.outputs)
            for k, d in zip(keys, PF.join_heads([self.outputs[k] for k in keys])):
                self.outputs[k] = d
        return self.outputs
