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

    def forward(self, x):
        return PF.padded_conv(x, self.conv, 2, reflect=self.use_refl)


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
    """layers.py:452-465."""
    grad_disp_x = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    grad_disp_y = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    grad_img_x = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    grad_img_y = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (grad_disp_x * torch.exp(-grad_img_x)).mean() + (grad_disp_y * torch.exp(-grad_img_y)).mean()


class SSIM(nn.Module):
    """layers.py:468-499; constructed by the trainer, inactive under --depth_supervision_only."""

    def __init__(self):
        super().__init__()
        self.pool = nn.AvgPool2d(3, 1)
        self.refl = nn.ReflectionPad2d(1)
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        if x.is_cuda:
            return ops.ssim(x, y)             # fused HIP kernels, forward and backward (pd_ssim_fwd / pd_ssim_bwd)
        x, y = self.refl(x), self.refl(y)
        mu_x, mu_y = self.pool(x), self.pool(y)
        sigma_x = self.pool(x ** 2) - mu_x ** 2
        sigma_y = self.pool(y ** 2) - mu_y ** 2
        sigma_xy = self.pool(x * y) - mu_x * mu_y
        n = (2 * mu_x * mu_y + self.C1) * (2 * sigma_xy + self.C2)
        d = (mu_x ** 2 + mu_y ** 2 + self.C1) * (sigma_x + sigma_y + self.C2)
        return torch.clamp((1 - n / d) / 2, 0, 1)


def compute_depth_errors(gt, pred):
    """layers.py:539-557."""
    thresh = torch.max((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def compute_depth_errors_numpy(gt, pred):
    """layers.py:559-577."""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
