"""manydepth.normals_vec façade (reference manydepth/normals_vec.py:11-60) on the HIP kernel.

``rho_diffuse`` / ``rho_spec`` / ``calc_normals`` keep their signatures; the per-pixel table
interpolation and trigonometry run in ``pd_polar_normals_from_xolp`` instead of the reference's
GPU -> CPU(scipy) -> GPU round trip.  The thetas themselves are not materialised by the kernel;
callers that need them get them by inverting the N3 = cos(theta) channel is NOT offered -- use
``get_normals`` (pre_encoders.ShallowNormalsEncoder.get_normals) for the fused result.
"""
import torch

from polardepth import polar as _polar


def get_normals(xolp, n=1.5):
    """[B,2,H,W] fp32 (DoLP, AoLP) -> [B,9,H,W] fp32: cat(N_diff, N_spec1, N_spec2)."""
    return _polar.normals_from_xolp(xolp, n)


def _normals_of(rho, phi, n):
    x = torch.stack((rho, phi), 1)
    return get_normals(x, n)


def calc_normals(phi, theta):
    """N = (cos(phi) sin(theta), sin(phi) sin(theta), cos(theta)); elementwise, on the tensors' device."""
    N1 = (torch.cos(phi) * torch.sin(theta)).unsqueeze(1)
    N2 = (torch.sin(phi) * torch.sin(theta)).unsqueeze(1)
    N3 = torch.cos(theta).unsqueeze(1)
    return torch.cat((N1, N2, N3), 1)


def rho_diffuse(rho, n):
    """theta_diffuse from the fused kernel: N_diff with phi = 0 is (sin(theta), 0, cos(theta))."""
    nd = _normals_of(rho, torch.zeros_like(rho), n)[:, 0:3].double()
    return torch.atan2(nd[:, 0], nd[:, 2])


def rho_spec(rho, n):
    """(theta_spec1, theta_spec2) modulo 2*pi, recovered the same way (phi' = pi/2 -> N = (~0, sin, cos))."""
    ns = _normals_of(rho, torch.zeros_like(rho), n).double()
    return torch.atan2(ns[:, 4], ns[:, 5]), torch.atan2(ns[:, 7], ns[:, 8])
