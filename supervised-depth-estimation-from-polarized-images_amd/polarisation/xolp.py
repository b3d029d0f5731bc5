"""polarisation.xolp façade: ``Iun_and_xolp`` served by the fused HIP kernel K1.

Reference: polarisation/xolp.py:8-34.  Differences, by design (SURVEY.md §7 hard part 1):
the canonical closed form replaces the LAPACK least-squares solve (identical up to lstsq noise;
the AoLP branch x2 == 0, x1 < 0 is +pi/2), inputs must hold integers in 0..255, and DoLP / AoLP
are the fp32-rounded values the network consumes (returned as float64 arrays like the reference).
"""
import numpy as np
import torch

from polardepth import polar as _polar

_ANGLES = np.array([0, 45, 90, 135]) * np.pi / 180


def Iun_and_xolp(images, angles=None):
    """images [H,W,4] (0/45/90/135 deg) -> (Iun, rho, phi), each [H,W] float64."""
    images = np.asarray(images)
    if angles is not None and not np.allclose(np.asarray(angles, dtype=np.float64), _ANGLES):
        raise NotImplementedError("the HIP kernel implements the fixed 0/45/90/135 degree polarizer set")
    if images.ndim != 3 or images.shape[2] != 4:
        raise ValueError(f"images must be [H,W,4], got {images.shape}")
    u8 = images.astype(np.uint8)
    if not np.array_equal(u8, images):
        raise ValueError("Iun_and_xolp (HIP) expects integer intensities in 0..255")
    H, W, _ = u8.shape
    pad = (-H * W) % 4
    pol = torch.from_numpy(np.ascontiguousarray(np.moveaxis(u8, -1, 0)))[None]
    if pad:   # kernel works on multiples of 4 pixels: pad the flattened planes
        flat = torch.zeros((1, 4, 1, H * W + pad), dtype=torch.uint8)
        flat[0, :, 0, :H * W] = pol.reshape(4, -1)
        pol = flat
    out = _polar.polar_forward(pol.cuda(), want=("xolp",))["xolp"][0].double().cpu().numpy()
    rho = out[0].reshape(-1)[:H * W].reshape(H, W)
    phi = out[1].reshape(-1)[:H * W].reshape(H, W)
    Iun = u8.astype(np.float64).sum(2) / 4.0      # (Imax + Imin) / 2 == x0
    return Iun, rho, phi
