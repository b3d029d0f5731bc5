"""Host wrapper of K1, the fused Stokes/DoLP/AoLP/normals kernel (csrc/polar.hip).

Mirrors polarisation/xolp.py:8-34 (Iun_and_xolp), indoor_dataset.py:430-442 (get_xolp),
normals_vec.py:11-60 and pre_encoders.py:78-79,99-113 of the reference.
"""
import ctypes
import functools

import numpy as np
import torch

from ._lib import lib, check, ptr, stream_ptr

MODE_LS, MODE_STOKES = 0, 1
N_THETA = 1000


def theta_tables_numpy(n=1.5):
    """The three theta(rho) tables exactly as the reference builds them (NumPy, fp64).

    normals_vec.py:13-19 (diffuse) and :27-47 (specular split at argmax); each table is
    returned x-ascending the way scipy.interp1d(assume_sorted=False) stores it.
    """
    th = np.linspace(0, np.pi / 2, N_THETA)
    s = np.sin(th)
    rho_d = ((n - 1 / n) ** 2 * s ** 2) / (
        2 + 2 * n ** 2 - (n + 1 / n) ** 2 * s ** 2 + 4 * np.cos(th) * np.sqrt(n ** 2 - s ** 2))
    rho_s = (2 * s ** 2 * np.cos(th) * np.sqrt(n ** 2 - s ** 2)) / (
        n ** 2 - s ** 2 - n ** 2 * s ** 2 + 2 * s ** 4)
    imax = int(np.argmax(rho_s))

    def asc(x, y):
        ind = np.argsort(x, kind="mergesort")
        return np.ascontiguousarray(x[ind]), np.ascontiguousarray(y[ind])

    return asc(rho_d, th), asc(rho_s[:imax], th[:imax]), asc(rho_s[imax:], th[imax:])


def pack_tables(tables):
    """[(x,y)]*3 fp64 -> packed host blob (uint8 tensor) via pd_polar_tables_pack."""
    (xd, yd), (x1, y1), (x2, y2) = tables
    nbytes = lib.pd_polar_tables_bytes(len(xd), len(x1), len(x2))
    if nbytes == 0:
        raise ValueError("theta tables need at least two nodes each")
    blob = torch.empty(nbytes, dtype=torch.uint8)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    check(lib.pd_polar_tables_pack(dp(xd), dp(yd), len(xd), dp(x1), dp(y1), len(x1), dp(x2), dp(y2), len(x2),
                                   ctypes.c_void_p(blob.data_ptr()), nbytes), "pd_polar_tables_pack")
    return blob


def build_tables_libm(n=1.5):
    """Packed blob computed entirely inside the library (libm); for C-only consumers."""
    nbytes = lib.pd_polar_tables_bytes(N_THETA, N_THETA, N_THETA)
    blob = torch.empty(nbytes, dtype=torch.uint8)
    used = ctypes.c_size_t(0)
    check(lib.pd_polar_tables_build(float(n), ctypes.c_void_p(blob.data_ptr()), nbytes, ctypes.byref(used)),
          "pd_polar_tables_build")
    return blob[:used.value].clone()


@functools.lru_cache(maxsize=8)
def _device_tables(n, device_index):
    blob = pack_tables(theta_tables_numpy(n))
    return blob.to(torch.device("cuda", device_index))


def polar_forward(pol, n=1.5, mode=MODE_LS, mask=None, want=("xolp",), tables=None, out_width=None, out=None,
                  precise=False, ieee_rho=False, nt_loads=None):
    """Run K1 on ``pol`` [B,4,H,W] uint8 (planes 0/45/90/135 deg) on the GPU.

    want: any of "xolp", "xolp_std", "normals", "ints".  Returns a dict of fp32 NCHW tensors
    ([B,2,H,W], [B,2,H,W], [B,9,H,W]) and the int32 [B,5,H,W] by-products.  out_width > W makes every
    output [.., H, out_width] with the extra right columns zero (612 -> 640 padding for the network).
    precise=True selects PD_POLAR_PRECISE_NORMALS (fp64 theta trig; the default fp32 path is within ~3e-7 of it);
    ieee_rho=True selects PD_POLAR_IEEE_RHO (the literal fp64 sqrt/div sequence for every pixel: same bits, slower).
    nt_loads=True / False forces the nontemporal hint on / off the plane loads (PD_POLAR_NT_LOADS / PD_POLAR_PLAIN_LOADS:
    measurement; None = the library's size rule).
    """
    if not (isinstance(pol, torch.Tensor) and pol.is_cuda):
        raise RuntimeError("polar_forward needs a CUDA(HIP) uint8 tensor; there is no CPU fallback")
    if pol.dtype != torch.uint8 or pol.dim() != 4 or pol.shape[1] != 4:
        raise ValueError(f"pol must be uint8 [B,4,H,W], got {pol.dtype} {tuple(pol.shape)}")
    pol = pol.contiguous()
    B, _, H, W = pol.shape
    if tables is None:
        tables = _device_tables(float(n), pol.device.index)
    out = dict(out) if out is not None else {}      # pre-allocated outputs may be passed in
    Wout = W if out_width is None else int(out_width)
    mk = lambda c, dt=torch.float32: torch.empty((B, c, H, Wout), dtype=dt, device=pol.device)
    for key, ch, dt in (("xolp", 2, torch.float32), ("xolp_std", 2, torch.float32), ("normals", 9, torch.float32),
                        ("ints", 5, torch.int32)):
        if key in want and key not in out:
            out[key] = mk(ch, dt)
    if mask is not None:
        mask = mask.to(torch.uint8).contiguous()
    with torch.cuda.device(pol.device):
        check(lib.pd_polar_fwd(ptr(pol), ptr(mask), ptr(out.get("xolp")), ptr(out.get("xolp_std")),
                               ptr(out.get("normals")), ptr(out.get("ints")), ptr(tables), tables.numel(),
                               B, H, W, Wout, mode, int(bool(precise)) | (2 if ieee_rho else 0) |
                               (0 if nt_loads is None else (4 if nt_loads else 8)), stream_ptr()), "pd_polar_fwd")
    return out


def split_mosaic(mosaic):
    """Raw sensor frame [..., 2h, 2w] with the four polarizer images as QUADRANTS -> planes [..., 4, h, w] in K1's
    order 0/45/90/135 degrees.  Quadrant layout of the reference's offline splitter (polarisation/
    pol_split_and_save.py:16-25 with the angle labels of indoor_dataset.py:435-438): top-left = pol00 (0 deg),
    top-right = pol01 (45), bottom-left = pol10 (90), bottom-right = pol11 (135).  Works on host or device tensors
    (one strided copy); with it the loader can hand over the un-split frame."""
    H2, W2 = mosaic.shape[-2], mosaic.shape[-1]
    if H2 % 2 or W2 % 2:
        raise ValueError(f"split_mosaic: the quadrant frame must have even sides, got {H2}x{W2}")
    h, w = H2 // 2, W2 // 2
    tl, tr = mosaic[..., :h, :w], mosaic[..., :h, w:]
    bl, br = mosaic[..., h:, :w], mosaic[..., h:, w:]
    return torch.stack((tl, tr, bl, br), dim=-3).contiguous()


def normals_from_xolp(xolp, n=1.5, precise=False):
    """ShallowNormalsEncoder.get_normals on the GPU: fp32 [B,2,H,W] (DoLP, AoLP) -> fp32 [B,9,H,W]."""
    if not (isinstance(xolp, torch.Tensor) and xolp.is_cuda):
        raise RuntimeError("normals_from_xolp needs a CUDA(HIP) tensor; there is no CPU fallback")
    if xolp.dim() != 4 or xolp.shape[1] != 2:
        raise ValueError(f"xolp must be [B,2,H,W], got {tuple(xolp.shape)}")
    xolp = xolp.float().contiguous()
    B, _, H, W = xolp.shape
    tables = _device_tables(float(n), xolp.device.index)
    out = torch.empty((B, 9, H, W), dtype=torch.float32, device=xolp.device)
    with torch.cuda.device(xolp.device):
        check(lib.pd_polar_normals_from_xolp(ptr(xolp), ptr(out), ptr(tables), tables.numel(), B, H, W,
                                             int(bool(precise)), stream_ptr()), "pd_polar_normals_from_xolp")
    return out


def theta_from_rho(rho, n=1.5, want=("d", "s1", "s2"), want_bins=False):
    """rho_diffuse / rho_spec of the reference on the GPU (normals_vec.py:11-50): fp32 rho of any shape -> dict of fp64
    tensors theta_d / theta_s1 / theta_s2 of that shape (scipy interp1d 'extrapolate' semantics, evaluated in fp64 in
    scipy's operation order) and, optionally, "bins": int32 [3, *shape] searchsorted indices."""
    if not (isinstance(rho, torch.Tensor) and rho.is_cuda):
        raise RuntimeError("theta_from_rho needs a CUDA(HIP) tensor; there is no CPU fallback")
    r = rho.float().contiguous()
    tables = _device_tables(float(n), r.device.index)
    out = {k: torch.empty(r.shape, dtype=torch.float64, device=r.device) for k in want}
    bins = torch.empty((3,) + tuple(r.shape), dtype=torch.int32, device=r.device) if want_bins else None
    with torch.cuda.device(r.device):
        check(lib.pd_polar_theta(ptr(r), ptr(out.get("d")), ptr(out.get("s1")), ptr(out.get("s2")), ptr(bins), ptr(tables),
                                 tables.numel(), r.numel(), stream_ptr()), "pd_polar_theta")
    if want_bins:
        out["bins"] = bins
    return out


def calc_normals(phi, theta):
    """normals_vec.py:53-60 on the GPU (pd_polar_calc_normals): phi, theta of one shape [B, ...] -> [B,3,...]; fp64 when
    either operand is fp64 (a factor of an fp32 operand is evaluated in fp32 and promoted, as torch does there)."""
    if not (phi.is_cuda and theta.is_cuda):
        raise RuntimeError("calc_normals needs CUDA(HIP) tensors; there is no CPU fallback")
    if phi.shape != theta.shape:
        raise ValueError(f"calc_normals: phi {tuple(phi.shape)} and theta {tuple(theta.shape)} must have one shape")
    conv = lambda t: t.contiguous() if t.dtype in (torch.float32, torch.float64) else t.float().contiguous()
    phi, theta = conv(phi), conv(theta)
    B = phi.shape[0] if phi.dim() > 0 else 1
    P = phi.numel() // max(B, 1)
    f64 = phi.dtype == torch.float64 or theta.dtype == torch.float64
    out = torch.empty((B, 3) + tuple(phi.shape[1:]), dtype=torch.float64 if f64 else torch.float32, device=phi.device)
    with torch.cuda.device(phi.device):
        check(lib.pd_polar_calc_normals(ptr(phi), ptr(theta), ptr(out), B, P, int(phi.dtype == torch.float64),
                                        int(theta.dtype == torch.float64), stream_ptr()), "pd_polar_calc_normals")
    return out
