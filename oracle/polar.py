"""Oracle (CPU restatement) of the per-pixel polarization preprocessing.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Never imported by the product.

Reference functions restated here (paths inside the reference checkout):

* ``polarisation/xolp.py:8-34``  ``Iun_and_xolp``  (dup ``polarisation/xolp_and_normals.py:13-39``)
* ``manydepth/datasets/indoor_dataset.py:430-442``  ``IndoorDataset.get_xolp`` (plane order, stacking)
* ``ppp_code/physical_normals_channels.py:15-36``  ``PolarisationImage_channel`` (Stokes variant)
* ``manydepth/normals_vec.py:11-22 / 25-50 / 53-60``  ``rho_diffuse`` / ``rho_spec`` / ``calc_normals``
  (numpy dups ``polarisation/xolp_and_normals.py:41-98``, ``ppp_code/physical_normals_channels.py:39-83``)
* ``manydepth/networks/pre_encoders.py:76-81, 99-113``  ``normalizeInput`` / ``get_normals``
* scipy ``interp1d(kind='linear', fill_value='extrapolate')`` == ``_call_linear``
  (scipy 1.15.3 ``interpolate/_interpolate.py``): searchsorted-left, clip to [1, n-1],
  ``slope*(x_new-x_lo)+y_lo`` with x sorted ascending (stable mergesort).

Canonical closed form (SURVEY.md §7 hard part 1): the reference solves a 4x3
least-squares system per pixel whose exact solution is x0 = (I0+I45+I90+I135)/4,
x1 = (I0-I90)/2, x2 = (I45-I135)/2.  LAPACK returns that solution with +-1e-14
noise which decides the atan2 branch when x2 == 0 and x1 < 0; the build defines
the closed form evaluated with the reference's op order in fp64 as canonical.
``iun_and_xolp_lstsq`` keeps the literal lstsq formulation so that agreement with
it can be *reported* (tests/test_oracle_polar.py).
"""
import numpy as np
import torch

POL_ANGLES_DEG = (0.0, 45.0, 90.0, 135.0)
XOLP_MEAN = 0.08693199701957657   # pre_encoders.py:79
XOLP_STD = 0.44430732785457433    # pre_encoders.py:79
N_THETA = 1000                    # normals_vec.py:13,27


# --------------------------------------------------------------------------- A1
def iun_and_xolp_lstsq(images, angles=None):
    """Literal restatement of polarisation/xolp.py:8-34 (LAPACK lstsq)."""
    if angles is None:
        angles = np.array(POL_ANGLES_DEG) * np.pi / 180
    images = np.asarray(images, dtype=np.float64)
    I = images.reshape((images.shape[0] * images.shape[1], 4))
    A = np.zeros((4, 3))
    A[:, 0] = 1
    A[:, 1] = np.cos(2 * angles)
    A[:, 2] = np.sin(2 * angles)
    x = np.linalg.lstsq(A, I.T, rcond=None)[0].T
    return _xolp_from_x(x[:, 0], x[:, 1], x[:, 2], images.shape[:2])


def _xolp_from_x(x0, x1, x2, shape):
    """xolp.py:22-33 given the three fit coefficients (fp64)."""
    r = np.sqrt(x1 ** 2 + x2 ** 2)
    Imax = x0 + r
    Imin = x0 - r
    Iun = (Imax + Imin) / 2
    with np.errstate(divide='ignore', invalid='ignore'):
        rho = np.true_divide(Imax - Imin, Imax + Imin)
        rho[rho == np.inf] = 0
        rho = np.nan_to_num(rho)
    phi = 0.5 * np.arctan2(x2, x1)
    return Iun.reshape(shape), rho.reshape(shape), phi.reshape(shape)


def iun_and_xolp(images):
    """Canonical closed form of ``Iun_and_xolp`` for the fixed 0/45/90/135 angles.

    images: [H, W, 4] (uint8 or float holding integers), plane order 0/45/90/135.
    Returns (Iun, rho, phi) fp64 [H, W], op order of xolp.py:22-30.
    """
    I = np.asarray(images, dtype=np.float64)
    x0 = (I[..., 0] + I[..., 1] + I[..., 2] + I[..., 3]) / 4.0
    x1 = (I[..., 0] - I[..., 2]) / 2.0
    x2 = (I[..., 1] - I[..., 3]) / 2.0
    # x2 == -0.0 cannot occur for integer inputs; +0.0 keeps atan2(+0, x1<0) = +pi.
    return _xolp_from_x(x0.ravel(), x1.ravel(), x2.ravel() + 0.0, I.shape[:2])


def xolp_planes(pol_u8):
    """[B,4,H,W] uint8 planes (0/45/90/135) -> fp32 [B,2,H,W] (ch0 DoLP, ch1 AoLP).

    Mirrors indoor_dataset.py:430-442 (stack -> Iun_and_xolp -> stack(rho, phi)
    -> ToTensor) followed by ``.float()`` at trainer.py:508,510.  Also returns the
    integer by-products d1 = I0-I90, d2 = I45-I135 (int16) used by the exact tests.
    """
    pol = np.asarray(pol_u8)
    assert pol.dtype == np.uint8 and pol.ndim == 4 and pol.shape[1] == 4
    B, _, H, W = pol.shape
    out = np.empty((B, 2, H, W), np.float32)
    for b in range(B):
        _, rho, phi = iun_and_xolp(np.moveaxis(pol[b], 0, -1))
        out[b, 0] = rho.astype(np.float32)
        out[b, 1] = phi.astype(np.float32)
    p = pol.astype(np.int16)
    return out, p[:, 0] - p[:, 2], p[:, 1] - p[:, 3]


def standardize_xolp(x):
    """pre_encoders.py:78-79 on an fp32 tensor (torch CPU op order)."""
    x = torch.as_tensor(x)
    return (x - XOLP_MEAN) / XOLP_STD


# --------------------------------------------------------------------------- A2
def stokes_channel(images, mask):
    """ppp_code/physical_normals_channels.py:15-36 -> (rho, phi, Iun), zero outside mask."""
    images = np.asarray(images, dtype=np.float64)
    mask = np.asarray(mask, dtype=bool)
    I = images.reshape((-1, 4))
    s0 = I[:, 0] + I[:, 2]
    s1 = I[:, 0] - I[:, 2]
    s2 = I[:, 1] - I[:, 3]
    with np.errstate(divide='ignore', invalid='ignore'):
        rho = np.divide(np.sqrt(s1 ** 2 + s2 ** 2), s0)
    phi = 0.5 * np.arctan2(s2, s1)
    shp = images.shape[:2]
    rho2 = np.zeros(shp); phi2 = np.zeros(shp); iun2 = np.zeros(shp)
    rho2[mask] = rho.reshape(shp)[mask]
    phi2[mask] = phi.reshape(shp)[mask]
    iun2[mask] = (s0 / 2).reshape(shp)[mask]
    return rho2, phi2, iun2


# ----------------------------------------------------------------------- A3, A4
def theta_tables(n=1.5):
    """The three (x ascending, y) tables behind rho_diffuse / rho_spec.

    normals_vec.py:13-19 (diffuse), :27-47 (specular, split at argmax).
    Returns dict name -> (x, y) with x sorted ascending exactly like
    scipy.interp1d(assume_sorted=False) does (stable argsort), plus 'imax'.
    """
    th = np.linspace(0, np.pi / 2, N_THETA)
    rho_d = ((n - 1 / n) ** 2 * np.sin(th) ** 2) / (
        2 + 2 * n ** 2 - (n + 1 / n) ** 2 * np.sin(th) ** 2
        + 4 * np.cos(th) * np.sqrt(n ** 2 - np.sin(th) ** 2))
    rho_s = (2 * np.sin(th) ** 2 * np.cos(th) * np.sqrt(n ** 2 - np.sin(th) ** 2)) / (
        n ** 2 - np.sin(th) ** 2 - n ** 2 * np.sin(th) ** 2 + 2 * np.sin(th) ** 4)
    imax = int(np.argmax(rho_s))

    def _sorted(x, y):
        ind = np.argsort(x, kind="mergesort")
        return x[ind].copy(), y[ind].copy()

    return {
        "diffuse": _sorted(rho_d, th),
        "spec1": _sorted(rho_s[:imax], th[:imax]),
        "spec2": _sorted(rho_s[imax:], th[imax:]),
        "imax": imax, "rho_d": rho_d, "rho_s": rho_s, "theta": th,
    }


def interp_linear_extrap(x, y, x_new):
    """scipy interp1d._call_linear with fill_value='extrapolate'.

    Returns (y_new fp64, idx int64) with idx the clipped searchsorted index (the
    "bin index" by-product that the GPU path must reproduce exactly).
    """
    x_new = np.asarray(x_new)
    shp = x_new.shape
    xn = x_new.ravel()
    idx = np.searchsorted(x, xn).clip(1, len(x) - 1).astype(np.int64)
    lo = idx - 1
    x_lo, x_hi, y_lo, y_hi = x[lo], x[idx], y[lo], y[idx]
    slope = (y_hi - y_lo) / (x_hi - x_lo)
    with np.errstate(invalid='ignore', over='ignore'):
        y_new = slope * (xn - x_lo) + y_lo
    return y_new.reshape(shp), idx.reshape(shp)


def rho_diffuse(rho, n=1.5, return_idx=False):
    """normals_vec.py:11-22. rho: numpy fp32/fp64 array -> theta fp64."""
    x, y = theta_tables(n)["diffuse"]
    th, idx = interp_linear_extrap(x, y, rho)
    return (th, idx) if return_idx else th


def rho_spec(rho, n=1.5, return_idx=False):
    """normals_vec.py:25-50 -> (theta1, theta2) fp64."""
    t = theta_tables(n)
    th1, i1 = interp_linear_extrap(*t["spec1"], rho)
    th2, i2 = interp_linear_extrap(*t["spec2"], rho)
    return (th1, th2, i1, i2) if return_idx else (th1, th2)


# --------------------------------------------------------------------------- A5
def calc_normals(phi, theta):
    """normals_vec.py:53-60 with torch CPU dtype semantics.

    phi: fp32 (or fp64) tensor [B,H,W]; theta: fp64 tensor -> [B,3,H,W] promoted.
    """
    phi = torch.as_tensor(phi)
    theta = torch.as_tensor(theta)
    N1 = (torch.cos(phi) * torch.sin(theta)).unsqueeze(dim=1)
    N2 = (torch.sin(phi) * torch.sin(theta)).unsqueeze(dim=1)
    N3 = torch.cos(theta).unsqueeze(dim=1)
    return torch.cat((N1, N2, N3), dim=1)


def calc_normals_numpy(phi, theta):
    """polarisation/xolp_and_normals.py:85-98 (all-fp64 numpy twin, [H,W,3])."""
    N = np.zeros(phi.shape + (3,))
    N[..., 0] = np.cos(phi) * np.sin(theta)
    N[..., 1] = np.sin(phi) * np.sin(theta)
    N[..., 2] = np.cos(theta)
    return N


# --------------------------------------------------------------------------- A6
def get_normals(xolp, n=1.5):
    """pre_encoders.py:99-113 -> fp64 [B,9,H,W] (caller applies .float(), :95).

    xolp: fp32 tensor [B,2,H,W] (the reference interpolates ``xolp.float()``).
    """
    x = torch.as_tensor(xolp)
    rho = x[:, 0, :, :]
    phi = x[:, 1, :, :]
    rho_np = rho.cpu().numpy()
    theta_diff = torch.from_numpy(rho_diffuse(rho_np, n))
    t1, t2 = rho_spec(rho_np, n)
    N_diff = calc_normals(phi, theta_diff)
    N_spec1 = calc_normals(phi + np.pi / 2, torch.from_numpy(t1))
    N_spec2 = calc_normals(phi + np.pi / 2, torch.from_numpy(t2))
    return torch.cat((N_diff, N_spec1, N_spec2), dim=1)


def polar_forward(pol_u8, n=1.5):
    """Whole K1 contract on CPU: planes -> (xolp fp32, xolp_std fp32, normals fp32, ints).

    ints = dict(d1, d2, idx_d, idx_s1, idx_s2) -- the exactly reproducible by-products.
    """
    xolp, d1, d2 = xolp_planes(pol_u8)
    xt = torch.from_numpy(xolp)
    normals = get_normals(xt, n).float()
    _, idx_d = rho_diffuse(xolp[:, 0], n, return_idx=True)
    _, _, i1, i2 = rho_spec(xolp[:, 0], n, return_idx=True)
    ints = {"d1": d1, "d2": d2, "idx_d": idx_d, "idx_s1": i1, "idx_s2": i2}
    return xt, standardize_xolp(xt), normals, ints
