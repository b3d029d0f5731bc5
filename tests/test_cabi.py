"""The C-ABI library loads (no GPU needed) and exports every symbol include/polardepth.h declares;
host-only entry points (table packing) are exercised on the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from polardepth import _lib, polar as pdpolar
from oracle import polar as opolar

HEADER = os.path.join(ROOT, "include", "polardepth.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert "pd_polar_fwd" in syms and len(syms) >= 6
    h = ctypes.CDLL(_lib.lib.path)
    for s in syms:
        assert hasattr(h, s), f"{s} declared in polardepth.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "python binding and header disagree"


def test_table_blob_numpy_vs_oracle_tables():
    mine = pdpolar.theta_tables_numpy(1.5)
    t = opolar.theta_tables(1.5)
    for k, (x, y) in zip(("diffuse", "spec1", "spec2"), mine):
        np.testing.assert_array_equal(x, t[k][0])
        np.testing.assert_array_equal(y, t[k][1])


def test_table_blob_layout_and_lut():
    blob = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy().tobytes()
    hdr = np.frombuffer(blob[:32], dtype=np.uint32)
    assert hdr[0] == 0x50444c54 and tuple(hdr[1:4]) == (1000, 625, 375)
    off_lut, off_lds, lds_bytes, total = (int(v) for v in hdr[4:8])
    assert total == len(blob) and lds_bytes <= 64 * 1024
    lut = np.frombuffer(blob[off_lut:off_lut + 511 * 511 * 4], dtype=np.float32).reshape(511, 511)
    d = np.arange(-255, 256, dtype=np.float64)
    ref = (0.5 * np.arctan2(d[:, None] / 2, d[None, :] / 2)).astype(np.float32)   # xolp.py:30
    np.testing.assert_array_equal(lut, ref)
    assert lut[255, 0] == np.float32(np.pi / 2)            # canonical branch: x2 == 0, x1 < 0 -> +pi/2
    # keys are floor32 of the fp64 nodes: searchsorted on them is exact for fp32 queries
    nk = 2000
    keys = np.frombuffer(blob[off_lds:off_lds + nk * 4], dtype=np.float32)
    x_all = np.concatenate([opolar.theta_tables(1.5)[k][0] for k in ("diffuse", "spec1", "spec2")])
    assert np.all(keys.astype(np.float64) <= x_all)
    assert np.all(np.nextafter(keys, np.float32(np.inf)).astype(np.float64) > x_all)
    bins = np.frombuffer(blob[off_lds + nk * 4:off_lds + nk * 4 + nk * 24], dtype=np.float64).reshape(3, nk)
    x, y = opolar.theta_tables(1.5)["diffuse"]
    np.testing.assert_array_equal(bins[0, 1:1000], x[:-1])
    np.testing.assert_array_equal(bins[1, 1:1000], y[:-1])
    np.testing.assert_array_equal(bins[2, 1:1000], (y[1:] - y[:-1]) / (x[1:] - x[:-1]))


def test_libm_tables_close_to_numpy_tables():
    a = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy()
    b = pdpolar.build_tables_libm(1.5).numpy()
    assert a.shape == b.shape
    off = 64 + (511 * 511 * 4 + 15) // 16 * 16
    np.testing.assert_array_equal(a[:off], b[:off])         # header + LUT identical
    fa = np.frombuffer(a[off + 8000:].tobytes(), dtype=np.float64)
    fb = np.frombuffer(b[off + 8000:].tobytes(), dtype=np.float64)
    # x_lo / y_lo within a few ulp (libm vs NumPy sin/cos); slopes amplify that near the flat ends
    np.testing.assert_allclose(fa[:4000], fb[:4000], rtol=1e-13, atol=1e-300)


def test_bad_arguments_are_rejected():
    x = np.array([0.0, 1.0]); y = np.array([0.0, 1.0])
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    buf = np.zeros(16, np.uint8)
    rc = _lib.lib.pd_polar_tables_pack(dp(x), dp(y), 2, dp(x), dp(y), 2, dp(x), dp(y), 2,
                                       ctypes.c_void_p(buf.ctypes.data), buf.size)
    assert rc == -22 and b"too small" in _lib.lib.pd_last_error()
    xd = np.array([1.0, 0.0])
    big = np.zeros(_lib.lib.pd_polar_tables_bytes(2, 2, 2), np.uint8)
    rc = _lib.lib.pd_polar_tables_pack(dp(xd), dp(y), 2, dp(x), dp(y), 2, dp(x), dp(y), 2,
                                       ctypes.c_void_p(big.ctypes.data), big.size)
    assert rc == -22 and b"ascending" in _lib.lib.pd_last_error()
    # product path refuses CPU tensors loudly
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pdpolar.polar_forward(torch.zeros(1, 4, 4, 4, dtype=torch.uint8))
