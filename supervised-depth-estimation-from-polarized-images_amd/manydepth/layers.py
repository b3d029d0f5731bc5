"""manydepth.layers façade: the hot-path subset of reference manydepth/layers.py on HIP kernels.

``ConvBlock`` / ``Conv3x3`` (:329-380) run as one reflect-padded implicit-GEMM conv with the ELU in
its epilogue; ``upsample`` (:446-449) is the bilinear x2 kernel.  The small loss helpers keep
their reference formulas as torch expressions for API compatibility (the training loop itself uses
the fused K5 loss kernels, see polardepth.functional.multiscale_loss).
"""
import numpy as np
import torch
import torch.nn as nn

from polardepth import functional as PF
from polardepth import ops


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:62-71."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


class Conv3x3(nn.Module):
    """ReflectionPad2d(1) + Conv2d(3) (layers.py:364-380)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        if not use_refl:
            raise NotImplementedError("zero-padded Conv3x3 is not used on the hot path")
        self.pad = nn.ReflectionPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, act=ops.ACT_NONE, **mails):
        return PF.reflect_conv_act(x, self.conv, act, **mails)


class Conv5x5(nn.Module):
    """ReflectionPad2d(2) (or ZeroPad2d(2)) + Conv2d(5) (layers.py:345-362; the decoder's uncertainty heads,
    depth_decoder.py:49-50)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(2) if use_refl else nn.ZeroPad2d(2)
        self.use_refl = bool(use_refl)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 5)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, act=ops.ACT_NONE):
        return PF.padded_conv(x, self.conv, 2, reflect=self.use_refl, act=act)


class ConvBlock(nn.Module):
    """Conv3x3 + ELU (layers.py:329-342), one kernel."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x, **mails):
        return self.conv(x, act=ops.ACT_ELU, **mails)


def upsample(x):
    """F.interpolate(scale_factor=2, bilinear, align_corners=False) (layers.py:446-449)."""
    return PF.upcat(x, None)


def get_smooth_loss(disp, img):
    """layers.py:452-465 (edge-aware first-order smoothness of `disp` as given) on the K5 kernels, differentiable in disp."""
    return PF.smooth_loss(disp, img)


class SSIM(nn.Module):
    """layers.py:468-499 as fused HIP kernels, forward and backward (pd_ssim_fwd / pd_ssim_bwd); constructed by the trainer,
    inactive under --depth_supervision_only."""

    def __init__(self):
        super().__init__()
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2       # (the kernels carry the same constants)

    def forward(self, x, y):
        return ops.ssim(x, y)


def compute_depth_errors(gt, pred):
    """layers.py:539-557 on the device (pd_depth_metrics over all elements handed in, which the reference's callers have
    masked already): (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) as 0-dim tensors."""
    m = ops.depth_metrics(gt.reshape(1, -1), pred.reshape(1, -1), 0.0, float("inf"))[0]
    return tuple(m[i] for i in range(7))


def compute_depth_errors_numpy(gt, pred):
    """layers.py:559-577: the same seven numbers for NumPy arrays (evaluation.py:215-288), as Python floats."""
    if not torch.cuda.is_available():
        raise RuntimeError("compute_depth_errors_numpy runs on the MI355X; there is no CPU fallback")
    vals = compute_depth_errors(torch.from_numpy(np.ascontiguousarray(gt, dtype=np.float32)).cuda(),
                                torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)).cuda())
    return tuple(float(v) for v in vals)
