"""manydepth.normals_vec facade (reference manydepth/normals_vec.py:11-60) on the HIP kernels.

``rho_diffuse`` / ``rho_spec`` / ``calc_normals`` keep their signatures and results: the thetas are the reference's fp64
``scipy.interpolate.interp1d(..., fill_value="extrapolate")`` values (``pd_polar_theta`` evaluates scipy's
``slope * (rho - x_lo) + y_lo`` in fp64, so DoLP beyond a table extrapolates to the same +41 / -135 rad), returned as
fp64 CPU tensors of rho's shape exactly like ``torch.from_numpy(f(rho))`` there.  The training path does not call these
(``ShallowNormalsEncoder.get_normals`` is served by the fused kernel K1); ``get_normals`` below is that fused result.
"""
import torch

from polardepth import polar as _polar


def get_normals(xolp, n=1.5):
    """[B,2,H,W] fp32 (DoLP, AoLP) -> [B,9,H,W] fp32: cat(N_diff, N_spec1, N_spec2)  (pre_encoders.py:99-113)."""
    return _polar.normals_from_xolp(xolp, n)


def _device(rho):
    if not torch.cuda.is_available():
        raise RuntimeError("manydepth.normals_vec runs on the MI355X; there is no CPU fallback")
    return rho if rho.is_cuda else rho.cuda()


def rho_diffuse(rho, n):
    """normals_vec.py:11-22: theta_diffuse(rho), fp64 CPU tensor."""
    return _polar.theta_from_rho(_device(rho), n, want=("d",))["d"].cpu()


def rho_spec(rho, n):
    """normals_vec.py:25-50: (theta_spec1, theta_spec2), fp64 CPU tensors (branches below / above the DoLP maximum)."""
    out = _polar.theta_from_rho(_device(rho), n, want=("s1", "s2"))
    return out["s1"].cpu(), out["s2"].cpu()


def calc_normals(phi, theta):
    """normals_vec.py:53-60: the unit normal of azimuth phi and zenith theta, [B,3,H,W] on the GPU (pd_polar_calc_normals);
    the dtype follows torch's promotion there (fp32 phi with fp64 theta gives fp64)."""
    return _polar.calc_normals(_device(phi), _device(theta))
