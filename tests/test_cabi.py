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


BUCKETS = 4096


def _blob_offsets(nk=2000, buckets=BUCKETS):
    off_bstart = ((nk + 3) // 4 * 4 + 4) * 4        # keys padded with +inf: four are read at once
    off_rank = off_bstart + (buckets + 4) * 2 + 8
    off_bins = (off_rank + (nk + 1) * 8 + 15) // 16 * 16
    return off_bstart, off_rank, off_bins


def test_table_blob_layout_and_lut():
    blob = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy().tobytes()
    hdr = np.frombuffer(blob[:32], dtype=np.uint32)
    assert hdr[0] == 0x50444c34 and tuple(hdr[1:4]) == (1000, 625, 375)
    off_lut, off_lds, lds_bytes, total = (int(v) for v in hdr[4:8])
    assert total == len(blob) and lds_bytes <= 160 * 1024
    lut = np.frombuffer(blob[off_lut:off_lut + 511 * 511 * 4], dtype=np.float32).reshape(511, 511)
    d = np.arange(-255, 256, dtype=np.float64)
    ref = (0.5 * np.arctan2(d[:, None] / 2, d[None, :] / 2)).astype(np.float32)   # xolp.py:30
    np.testing.assert_array_equal(lut, ref)
    assert lut[255, 0] == np.float32(np.pi / 2)            # canonical branch: x2 == 0, x1 < 0 -> +pi/2
    # merged keys are floor32 of the fp64 nodes, sorted; rank[p] = per-table counts among the first p keys,
    # so rank[searchsorted(merged, v)] reproduces the three per-table searchsorted results for fp32 v
    nk = 2000
    img = blob[off_lds:off_lds + lds_bytes]
    off_bstart, off_rank, off_bins = _blob_offsets(nk)
    mkeys = np.frombuffer(img[:nk * 4], dtype=np.float32)
    assert np.all(np.diff(mkeys) >= 0)
    rank = np.frombuffer(img[off_rank:off_rank + (nk + 1) * 8], dtype=np.uint16).reshape(nk + 1, 4)
    assert tuple(rank[0][:3]) == (0, 0, 0) and tuple(rank[nk][:3]) == (1000, 625, 375)
    tabs = [opolar.theta_tables(1.5)[k][0] for k in ("diffuse", "spec1", "spec2")]
    rng = np.random.default_rng(0)
    q = np.concatenate([rng.random(5000).astype(np.float32) * 1.2, mkeys[::7], np.float32([0, 1, 2, 0.38461538])])
    pos = np.searchsorted(mkeys, q, side="left")
    for t in range(3):
        np.testing.assert_array_equal(rank[pos, t], np.searchsorted(tabs[t], q.astype(np.float64), side="left"))
    # sqrt(rho) bucket index: the true position always lies in [bstart[b-1], bstart[b+2]] for the device's b
    nb = BUCKETS + 4
    bstart = np.frombuffer(img[off_bstart:off_bstart + nb * 2], dtype=np.uint16).astype(int)
    bscale = np.frombuffer(img[off_bstart + nb * 2:off_bstart + nb * 2 + 4], dtype=np.float32)[0]
    assert np.all(np.diff(bstart) >= 0) and bstart[0] == 0 and bstart[BUCKETS + 1] == nk
    b = np.minimum(np.sqrt(np.maximum(q, 0)).astype(np.float32) * bscale, BUCKETS).astype(int)
    assert np.all(bstart[np.maximum(b - 1, 0)] <= pos) and np.all(pos <= bstart[b + 2])
    span = bstart[3:] - bstart[:-3]
    assert span.max() <= 32 and (span > 4).mean() < 0.01   # <= 4 candidate keys (one read each) almost everywhere
    assert np.all(np.isinf(np.frombuffer(img[nk * 4:off_bstart], dtype=np.float32)))   # +inf key padding
    bins = np.frombuffer(img[off_bins:off_bins + nk * 32], dtype=np.float64).reshape(nk, 4)
    x, y = opolar.theta_tables(1.5)["diffuse"]
    np.testing.assert_array_equal(bins[1:1000, 0], x[:-1])
    np.testing.assert_array_equal(bins[1:1000, 1], (y[1:] - y[:-1]) / (x[1:] - x[:-1]))
    np.testing.assert_allclose(bins[1:1000, 2], np.sin(y[:-1]), rtol=0, atol=2e-16)
    np.testing.assert_allclose(bins[1:1000, 3], np.cos(y[:-1]), rtol=0, atol=2e-16)


def test_libm_tables_close_to_numpy_tables():
    a = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy()
    b = pdpolar.build_tables_libm(1.5).numpy()
    assert a.shape == b.shape
    off = 64 + (511 * 511 * 4 + 15) // 16 * 16
    np.testing.assert_array_equal(a[:off], b[:off])         # header + LUT identical
    _, _, off_bins = _blob_offsets()
    fa = np.frombuffer(a[off + off_bins:off + off_bins + 2000 * 32].tobytes(), dtype=np.float64).reshape(2000, 4)
    fb = np.frombuffer(b[off + off_bins:off + off_bins + 2000 * 32].tobytes(), dtype=np.float64).reshape(2000, 4)
    # x_lo and sin/cos(y_lo) within a few ulp (libm vs NumPy sin/cos); slopes amplify that near the flat ends
    np.testing.assert_allclose(fa[:, [0, 2, 3]], fb[:, [0, 2, 3]], rtol=1e-12, atol=1e-15)


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


def test_shape_validation_of_the_fused_kernels_needs_no_gpu():
    """Argument checks of the entry points added for the decoder heads and the attention block return PD_EINVAL (-22)
    with a message before anything touches the device."""
    L = _lib.lib
    p = ctypes.c_void_p(16)      # a non-null, 16-byte aligned dummy: never dereferenced on these paths
    assert L.pd_attn_fwd(p, p, p, p, p, 1, 96, 64, 0.1, None) == -22 and b"head dimension" in L.pd_last_error()
    assert L.pd_attn_fwd(p, p, p, p, p, 1, 100, 128, 0.1, None) == -22 and b"multiple of 32" in L.pd_last_error()
    assert L.pd_attn_bwd(p, p, p, p, p, p, p, p, p, p, 1, 100, 128, 0.1, None) == -22
    assert L.pd_attn_fwd(None, None, None, None, None, 0, 96, 128, 0.1, None) == 0          # empty batch
    assert L.pd_disphead_fwd(p, p, p, p, 1, 8, 8, 24, None) == -22 and b"C=24" in L.pd_last_error()
    assert L.pd_disphead_fwd(p, p, p, p, 1, 1, 8, 16, None) == -22 and b"reflection" in L.pd_last_error()
    assert L.pd_disphead_bwd_weight(p, p, p, p, p, p, 16, 1, 8, 8, 16, 1, None) == -22 and b"workspace" in L.pd_last_error()
    assert L.pd_disphead_workspace(16) == 1024 * (9 * 16 + 1) * 4
    assert L.pd_conv2d(p, p, None, None, p, None, 1, 8, 8, 16, 1024, 128, 16, 1, 8, 8, 16, 3, 3, 1, 1, 0, 0, 0, 0.0, 1.0,
                       8, None) == -22 and b"row stride" in L.pd_last_error()
