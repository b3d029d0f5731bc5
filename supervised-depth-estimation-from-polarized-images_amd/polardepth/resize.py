"""Pillow's 8-bit LANCZOS resize (``Image.resize(size, Image.ANTIALIAS)``, indoor_dataset.py:335-349) on the GPU.

The coefficient tables follow Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c)
literally, in double precision with libm's sin (``math.sin``), so the device result is bit-identical to PIL."""
import functools
import math

import numpy as np
import torch

from ._lib import lib, check, ptr, stream_ptr

PRECISION_BITS = 32 - 8 - 2
_LANCZOS_SUPPORT = 3.0


def _lanczos(x):
    def sinc(v):
        if v == 0.0:
            return 1.0
        v *= math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


@functools.lru_cache(maxsize=64)
def lanczos_coeffs(in_size, out_size):
    """(coeffs int32 [out_size, ksize], bounds int32 [out_size, 2]) of one resampling pass."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = _LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


_DEV_TABLES = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, device.index)
    t = _DEV_TABLES.get(key)
    if t is None:
        kk, b = lanczos_coeffs(in_size, out_size)
        t = (torch.from_numpy(kk).to(device), torch.from_numpy(b).to(device), kk.shape[1])
        _DEV_TABLES[key] = t
    return t


def resize_lanczos_u8(x, size):
    """x: uint8 CUDA tensor [..., Hs, Ws]; size = (Hd, Wd).  Horizontal pass, then vertical pass (Pillow's order)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.uint8):
        raise RuntimeError("resize_lanczos_u8 needs a CUDA(HIP) uint8 tensor; there is no CPU fallback (PIL is the CPU path)")
    Hd, Wd = int(size[0]), int(size[1])
    lead, (Hs, Ws) = x.shape[:-2], x.shape[-2:]
    P = int(np.prod(lead)) if lead else 1
    cur = x.contiguous().view(P, Hs, Ws)
    with torch.cuda.device(x.device):
        if Wd != Ws:
            kk, b, ks = _tables(Ws, Wd, x.device)
            out = torch.empty((P, Hs, Wd), dtype=torch.uint8, device=x.device)
            check(lib.pd_resize_u8_pass(ptr(cur), ptr(out), ptr(kk), ptr(b), ks, P, Hs, Ws, Wd, 0, stream_ptr()),
                  "pd_resize_u8_pass")
            cur = out
        if Hd != Hs:
            kk, b, ks = _tables(Hs, Hd, x.device)
            out = torch.empty((P, Hd, cur.shape[2]), dtype=torch.uint8, device=x.device)
            check(lib.pd_resize_u8_pass(ptr(cur), ptr(out), ptr(kk), ptr(b), ks, P, Hs, cur.shape[2], Hd, 1, stream_ptr()),
                  "pd_resize_u8_pass")
            cur = out
    return cur.view(*lead, Hd, Wd)
