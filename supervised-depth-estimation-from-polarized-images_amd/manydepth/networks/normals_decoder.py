"""NormalsDecoder: the `arch1++_separate_normals_dec` variant of the reference (README.md:54 -- "an additional decoder after the
normals encoder.  The decoder directly predicts normals.  These normals are compared with normals calculated from ground truth
to drive the supervised learning").

The branch's source is not part of the reference checkout; this build defines the variant with the reference's own decoder
blocks (depth_decoder.py:29-67, layers.py:329-380): three times ConvBlock (reflect Conv3x3 + ELU, one kernel) followed by the
bilinear x2 kernel -- 64 -> 32 -> 16 -> 16 channels, H/8 -> H -- and a Conv3x3 head with three output channels (the raw normal;
the loss normalises it).  ``decoder.0..2`` are the ConvBlocks in execution order, ``decoder.3`` the head.  Parity: against
oracle/nets.py:NormalsDecoder (same definition on PyTorch-CPU), unpinned by the reference like the attention variant.
"""
import torch.nn as nn

from manydepth.layers import ConvBlock, Conv3x3
from polardepth import functional as PF
from polardepth import ops


class NormalsDecoder(nn.Module):
    def __init__(self, num_ch_in=64, num_ch_dec=(32, 16, 16)):
        super().__init__()
        chans = [int(num_ch_in)] + [int(c) for c in num_ch_dec]
        self.decoder = nn.ModuleList([ConvBlock(chans[i], chans[i + 1]) for i in range(len(num_ch_dec))] +
                                     [Conv3x3(chans[-1], 3)])

    def forward(self, x):
        for blk in list(self.decoder)[:-1]:
            x = PF.upcat(blk(x), None)            # bilinear x2 (no skip connection)
        return self.decoder[-1](x, act=ops.ACT_NONE)
