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
SQRT_MAGIC = 0x1fbd1df5
LUT = 511 * 511


def _header(blob):
    h = np.frombuffer(blob[:64], dtype=np.uint32)
    names = ("magic", "n_d", "n_s1", "n_s2", "off_lut", "off_lut4", "off_img_fast", "img_fast_bytes",
             "off_img_precise", "img_precise_bytes", "common_bytes", "total_bytes")
    d = {k: int(v) for k, v in zip(names, h[:12])}
    d["bscale"] = np.frombuffer(blob[48:52], dtype=np.float32)[0]
    d["n_buckets"] = int(h[13])
    d["off_ylo"] = int(h[14])
    return d


def _bucket_of(rho, bscale):
    """The device's bucket function (csrc/polar.hip:bucket_of): integer shift/add, one fp32 multiply, truncation."""
    u = np.asarray(rho, np.float32).view(np.uint32).copy()
    u[u >= 0x80000000] = 0
    t = ((u >> 1) + np.uint32(SQRT_MAGIC)).view(np.float32)
    v = t * np.float32(bscale)
    return np.minimum(v, np.float32(BUCKETS - 1)).astype(np.int64)


def test_table_blob_layout_and_lut():
    blob = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy().tobytes()
    h = _header(blob)
    assert h["magic"] == 0x50444c36 and (h["n_d"], h["n_s1"], h["n_s2"]) == (1000, 625, 375)
    assert h["total_bytes"] == len(blob) and h["n_buckets"] == BUCKETS
    assert h["img_fast_bytes"] % 1024 == 0 and h["img_precise_bytes"] % 1024 == 0 and h["img_precise_bytes"] <= 160 * 1024
    lut = np.frombuffer(blob[h["off_lut"]:h["off_lut"] + LUT * 4], dtype=np.float32).reshape(511, 511)
    d = np.arange(-255, 256, dtype=np.float64)
    ref = (0.5 * np.arctan2(d[:, None] / 2, d[None, :] / 2)).astype(np.float32)   # xolp.py:30
    np.testing.assert_array_equal(lut, ref)
    assert lut[255, 0] == np.float32(np.pi / 2)            # canonical branch: x2 == 0, x1 < 0 -> +pi/2
    # float4 LUT of the fast normals: phi, cos(fl32 phi), sin(fl32 phi) correctly rounded, 0
    lut4 = np.frombuffer(blob[h["off_lut4"]:h["off_lut4"] + LUT * 16], dtype=np.float32).reshape(511, 511, 4)
    np.testing.assert_array_equal(lut4[..., 0], ref)
    np.testing.assert_array_equal(lut4[..., 1], np.cos(ref.astype(np.float64)).astype(np.float32))
    np.testing.assert_array_equal(lut4[..., 2], np.sin(ref.astype(np.float64)).astype(np.float32))
    assert not lut4[..., 3].any()
    # y_lo of every bin (entry i = y[i-1]) behind the two LDS images: the theta outputs of pd_polar_theta
    assert h["off_ylo"] == h["off_img_precise"] + h["img_precise_bytes"] and h["off_ylo"] + 2000 * 8 == len(blob)
    ylo = np.frombuffer(blob[h["off_ylo"]:h["off_ylo"] + 2000 * 8], dtype=np.float64)
    o = 0
    for k in ("diffuse", "spec1", "spec2"):
        y = opolar.theta_tables(1.5)[k][1]
        np.testing.assert_array_equal(ylo[o + 1:o + len(y)], y[:-1])
        o += len(y)

    # one-read search: idx_T = base_T + (key_T < rho) reproduces clip(searchsorted(x_T, rho, 'left'), 1, n_T - 1)
    # for every fp32 rho whose bucket is not flagged; flagged buckets are few and take exact binary searches
    nk = 2000
    img = blob[h["off_img_fast"]:h["off_img_fast"] + h["img_fast_bytes"]]
    assert h["common_bytes"] == BUCKETS * 16 + nk * 4
    assert blob[h["off_img_precise"]:h["off_img_precise"] + h["common_bytes"]] == img[:h["common_bytes"]]
    rec = np.frombuffer(img[:BUCKETS * 16], dtype=np.uint32).reshape(BUCKETS, 4)
    keys_in = rec[:, :3].copy().view(np.float32)
    keys = np.frombuffer(img[BUCKETS * 16:BUCKETS * 16 + nk * 4], dtype=np.float32)
    names = ("diffuse", "spec1", "spec2")
    tabs = [opolar.theta_tables(1.5)[k][0] for k in names]
    ns = [len(t) for t in tabs]
    o = 0
    for t, n in zip(tabs, ns):       # keys = floor32 of the fp64 nodes
        k32 = t.astype(np.float32)
        k32 = np.where(k32.astype(np.float64) > t, np.nextafter(k32, np.float32(-np.inf)), k32)
        np.testing.assert_array_equal(keys[o:o + n], k32)
        o += n
    rng = np.random.default_rng(0)
    q = np.concatenate([rng.random(20000).astype(np.float32) * 1.2, keys[::3], np.nextafter(keys[::5], np.float32(2)),
                        np.float32([0, 1, 2, 0.38461538, 1e-7, 3e-6, 100.0, np.inf, -1.0, -0.0])])
    b = _bucket_of(q, h["bscale"])
    multi = (rec[b, 3] >> 30) != 0
    assert multi.mean() < 0.02 and ((rec[:, 3] >> 30) != 0).sum() < 24
    for t in range(3):
        want = np.searchsorted(tabs[t], q.astype(np.float64), side="left").clip(1, ns[t] - 1)
        got = ((rec[b, 3] >> (10 * t)) & 1023).astype(np.int64) + (keys_in[b, t] < q)
        np.testing.assert_array_equal(got[~multi], want[~multi])
    # the flagged region: rho within 5e-3 of the specular maximum plus slivers near zero
    flagged = np.flatnonzero((rec[:, 3] >> 30) != 0)
    grid = np.linspace(0, 1.05, 400001).astype(np.float32)
    gb = _bucket_of(grid, h["bscale"])
    bad = grid[np.isin(gb, flagged)]
    assert np.all((bad > 0.995) | (bad < 0.02))
    # bins: fast float4 (x_lo32, slope32, c_hi, c_lo|j) and precise fp64 (x_lo, slope, sin y_lo, cos y_lo)
    x, y = opolar.theta_tables(1.5)["diffuse"]
    pimg = blob[h["off_img_precise"]:h["off_img_precise"] + h["img_precise_bytes"]]
    bins = np.frombuffer(pimg[h["common_bytes"]:h["common_bytes"] + nk * 32], dtype=np.float64).reshape(nk, 4)
    slope = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
    np.testing.assert_array_equal(bins[1:1000, 0], x[:-1])
    np.testing.assert_array_equal(bins[1:1000, 1], slope)
    np.testing.assert_allclose(bins[1:1000, 2], np.sin(y[:-1]), rtol=0, atol=2e-16)
    np.testing.assert_allclose(bins[1:1000, 3], np.cos(y[:-1]), rtol=0, atol=2e-16)
    fb = np.frombuffer(img[h["common_bytes"]:h["common_bytes"] + nk * 16], dtype=np.float32).reshape(nk, 4)[1:1000]
    j = (fb[:, 3].copy().view(np.uint32) & 1).astype(np.float64)
    np.testing.assert_array_equal(j, (0.5 * (y[:-1] + y[1:]) > np.pi / 4).astype(np.float64))
    # theta at the bin's upper node from the fast record == y_hi within 2e-7
    rho_hi = keys[1:1000]
    r = fb[:, 1].astype(np.float64) * (rho_hi - fb[:, 0]).astype(np.float64) + fb[:, 2] + fb[:, 3]
    np.testing.assert_allclose(r + j * np.pi / 2, slope * (rho_hi.astype(np.float64) - x[:-1]) + y[:-1], rtol=0, atol=2e-7)


def test_libm_tables_close_to_numpy_tables():
    a = pdpolar.pack_tables(pdpolar.theta_tables_numpy(1.5)).numpy()
    b = pdpolar.build_tables_libm(1.5).numpy()
    assert a.shape == b.shape
    h = _header(a.tobytes())
    off = h["off_img_fast"]
    np.testing.assert_array_equal(a[:off], b[:off])         # header + LUTs identical
    s = h["off_img_precise"] + h["common_bytes"]
    fa = np.frombuffer(a[s:s + 2000 * 32].tobytes(), dtype=np.float64).reshape(2000, 4)
    fb = np.frombuffer(b[s:s + 2000 * 32].tobytes(), dtype=np.float64).reshape(2000, 4)
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
                       8, 0, None) == -22 and b"row stride" in L.pd_last_error()


def test_which_shapes_take_the_bf16_split_kernels():
    """pd_conv2d_uses_x3 / pd_conv2d_wgrad_uses_x3 (host logic, no GPU): the routing rule the profiler labels, bench.py's
    roofline object and the production-size tests rely on.  Arguments: M, Cout, C, KH, KW, stride, pad, mode, act, scale."""
    lib = _lib.lib
    AUTO, FP32, X3, REGS, GEN, IM2COL, ROWWG = 0, 1, 2, 4, 8, 16, 32     # PD_CONV_* flags (include/polardepth.h), the last argument of every query
    M16 = 16 * 256 * 320
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 5, 5, 1, 2, 0, 0, 0, 0, 0, AUTO) == 2            # encoder 5x5 at batch 16: 256-row tiles
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 3, 3, 1, 1, 2, 0, 0, 0, 0, AUTO) == 2            # its stride-1 data gradient
    assert lib.pd_conv2d_uses_x3(256 * 320, 64, 64, 3, 3, 1, 1, 0, 0, 0, 0, 0, AUTO) == 1      # 320 tiles of 256 rows, 640 of 128
    assert lib.pd_conv2d_uses_x3(16 * 16 * 20, 512, 512, 3, 3, 1, 1, 0, 0, 0, 0, 0, AUTO) == 1  # 320 tiles of 128 rows
    assert lib.pd_conv2d_uses_x3(16 * 16 * 20, 256, 512, 3, 3, 1, 1, 0, 0, 0, 0, 0, AUTO) == 0  # 160 tiles of 128 rows: fp32 kernel
    assert lib.pd_conv2d_uses_x3(M16, 32, 96, 3, 3, 1, 1, 0, 0, 0, 0, 0, AUTO) == 0            # 32 output channels
    assert lib.pd_conv2d_uses_x3(M16, 64, 36, 4, 4, 1, 2, 0, 0, 0, 0, 0, AUTO) == 2            # space-to-depth stem: 36 = 2 channel groups + 4
    assert lib.pd_conv2d_uses_x3(M16, 64, 4, 4, 4, 1, 2, 0, 0, 0, 0, 0, AUTO) == 0             # a group of 16 with 4 valid channels: no
    assert lib.pd_conv2d_uses_x3(M16, 64, 128, 3, 3, 1, 1, 1, 2, 0, 0, 0, AUTO) == 2           # decoder ConvBlock: 3x3 reflection padding + ELU
    assert lib.pd_conv2d_uses_x3(M16, 64, 128, 5, 5, 1, 2, 1, 0, 0, 0, 0, AUTO) == 0           # 5x5 reflection padding
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 1, 0, 0, 0, AUTO) == 0            # ReLU in the epilogue
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 3, 0, 0, 0, AUTO) == 0            # sigmoid in the epilogue
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 0, 1, 0, 0, AUTO) == 0            # folded BatchNorm scale
    assert lib.pd_conv2d_uses_x3(4 * M16, 64, 128, 3, 3, 2, 1, 2, 0, 0, 0, 0, AUTO) == 0       # stride-2 data gradient (parity launches)
    assert lib.pd_conv2d_uses_x3(M16 // 4, 128, 64, 3, 3, 2, 1, 0, 0, 0, 0, 0, AUTO) == 2      # stride-2 forward
    # M, Cout, C, KH, KW, stride, pad, mode, H, W, Ho, Wo
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, AUTO) == 3          # rolling-row kernel: 3 filter rows per workgroup, 160 columns x 3 slices
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, ROWWG) == 2         # ... or one filter row per workgroup (halo-tile kernel, transposed LDS reads)
    assert lib.pd_conv2d_wgrad_uses_x3(M16 // 8, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, AUTO) == 2     # two images: 20 columns, slices too short to roll
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 5, 5, 1, 2, 0, 256, 320, 256, 320, AUTO) == 2          # 5x5: 25 accumulator tiles do not fit
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, IM2COL) == 1        # ... unless the caller asks for the gather kernel
    assert lib.pd_conv2d_wgrad_uses_x3(16 * 64 * 80, 128, 128, 3, 3, 1, 1, 0, 64, 80, 64, 80, AUTO) == 2   # Wo = 80: 4 x 16 tiles
    assert lib.pd_conv2d_wgrad_uses_x3(16 * 32 * 40, 256, 256, 3, 3, 1, 1, 0, 32, 40, 32, 40, AUTO) == 2   # 8 x 8 tiles
    assert lib.pd_conv2d_wgrad_uses_x3(16 * 16 * 20, 512, 512, 3, 3, 1, 1, 0, 16, 20, 16, 20, AUTO) == 2   # Wo = 20: 8 x 4 tiles
    assert lib.pd_conv2d_wgrad_uses_x3(16 * 16 * 20, 512, 512, 5, 5, 1, 2, 0, 16, 20, 16, 20, AUTO) == 1   # ... 3x3 only: the gather kernel
    assert lib.pd_conv2d_wgrad_uses_x3(16 * 18 * 22, 512, 512, 3, 3, 1, 1, 0, 18, 22, 18, 22, AUTO) == 1   # Wo = 22: no tile shape
    assert lib.pd_conv2d_wgrad_uses_x3(M16 // 4, 64, 128, 3, 3, 1, 1, 1, 128, 160, 128, 160, AUTO) == 3   # decoder: ReflectionPad2d(1) + Conv3x3 (mirrored rows / columns), 80 columns x 3 slices
    assert lib.pd_conv2d_wgrad_uses_x3(M16 // 16, 128, 256, 3, 3, 1, 1, 1, 64, 80, 64, 80, AUTO) == 2    # ... 256 -> 128 @64x80: 80 columns > 64 slices: one filter row per workgroup
    assert lib.pd_conv2d_wgrad_uses_x3(M16 // 4, 64, 96, 3, 3, 1, 1, 1, 128, 160, 128, 160, AUTO) == 1    # ... 96 input channels: the gather kernel
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 5, 5, 1, 2, 1, 256, 320, 256, 320, AUTO) == 0       # reflect 5x5: general kernel
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 32, 96, 3, 3, 1, 1, 0, 256, 320, 256, 320, AUTO) == 2       # 32 output channels: the halo kernel's 32-channel workgroups
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 32, 96, 3, 3, 1, 1, 1, 256, 320, 256, 320, AUTO) == 2       # ... the decoder's 96 -> 32 (reflection padding)
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 32, 96, 3, 3, 1, 1, 0, 256, 320, 256, 320, IM2COL) == 0     # ... otherwise the fp32 kernel's 32-wide tile
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 32, 96, 5, 5, 1, 2, 0, 256, 320, 256, 320, AUTO) == 0       # (5x5 with 32 channels: not instantiated)
    assert lib.pd_conv2d_wgrad_workspace(M16, 32, 864, AUTO) >= 2 * (768 // 6) * (32 * 864 + 32) * 4   # two partial rows per slice
    # with the output grid known, the halo-tile kernel takes the 3x3 / 5x5 stride-1 layers with whole 8 x 32 tiles
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 5, 5, 1, 2, 0, 0, 0, 256, 320, AUTO) == 3
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 3, 3, 1, 1, 2, 0, 0, 256, 320, AUTO) == 3            # stride-1 data gradient
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 5, 5, 1, 2, 0, 0, 0, 256, 320, IM2COL) == 2          # ... unless the caller asks for the gather kernel
    assert lib.pd_conv2d_uses_x3(16 * 64 * 80, 128, 128, 3, 3, 1, 1, 0, 0, 0, 64, 80, AUTO) == 3   # Wo = 80: 16 x 16 tiles
    assert lib.pd_conv2d_uses_x3(16 * 32 * 40, 512, 256, 5, 5, 1, 2, 0, 0, 0, 32, 40, AUTO) == 3   # 32 x 40: 32 x 8 tiles, 640 workgroups
    assert lib.pd_conv2d_uses_x3(16 * 32 * 40, 256, 256, 3, 3, 1, 1, 0, 0, 0, 32, 40, AUTO) == 1   # ... 80 x 4 = 320 workgroups: the gather kernel's 128-row tiles
    assert lib.pd_conv2d_uses_x3(16 * 32 * 40, 256, 512, 5, 5, 1, 2, 2, 0, 0, 32, 40, AUTO) == 3   # ... 5x5 there: 640 halo workgroups of 32 columns
    assert lib.pd_conv2d_uses_x3(16 * 64 * 80, 64, 64, 3, 3, 1, 1, 0, 0, 0, 64, 80, AUTO) == 3     # 64 -> 64 @64x80: 320 -> 640 workgroups of 32 columns
    assert lib.pd_conv2d_uses_x3(16 * 16 * 20, 512, 512, 3, 3, 1, 1, 0, 0, 0, 16, 20, AUTO) == 1   # 16 x 20: no tile shape divides it
    assert lib.pd_conv2d_uses_x3(M16, 64, 36, 4, 4, 1, 2, 0, 0, 0, 256, 320, AUTO) == 3            # 4x4 space-to-depth stem: the halo kernel's row-window form
    assert lib.pd_conv2d_uses_x3(M16, 64, 36, 4, 4, 1, 2, 0, 0, 0, 256, 320, IM2COL) == 2          # ... or the gather kernel with a partly empty channel group
    assert lib.pd_conv2d_uses_x3(M16, 64, 36, 4, 4, 1, 2, 0, 0, 0, 250, 320, AUTO) == 2            # ... as does a grid no 8 x 32 tile divides
    assert lib.pd_conv2d_uses_x3(M16, 32, 96, 3, 3, 1, 1, 1, 2, 0, 256, 320, AUTO) == 3            # 32 output channels: 32-column workgroups (decoder 96 -> 32)
    assert lib.pd_conv2d_uses_x3(M16, 96, 32, 3, 3, 1, 1, 2, 0, 0, 256, 320, AUTO) == 3            # ... and its data gradient: three 32-column tiles
    assert lib.pd_conv2d_uses_x3(M16 // 4, 32, 64, 3, 3, 1, 1, 1, 2, 0, 128, 160, AUTO) == 3       # 64 -> 32 @128x160
    assert lib.pd_conv2d_uses_x3(M16, 32, 96, 5, 5, 1, 2, 0, 0, 0, 256, 320, AUTO) == 0            # (5x5 with 32 columns: not instantiated)
    assert lib.pd_conv2d_uses_x3(M16, 64, 128, 3, 3, 1, 1, 1, 2, 0, 256, 320, AUTO) == 3           # reflection padding: mirrored halo
    # the caller's flags decide the arithmetic -- no environment variable is read by the library
    assert lib.pd_conv2d_uses_x3(M16, 64, 64, 5, 5, 1, 2, 0, 0, 0, 0, 0, FP32) == 0
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, FP32) == 0
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, REGS) == 0     # the in-register split is the uni kernel's
    assert lib.pd_conv2d_wgrad_uses_x3(M16, 64, 64, 3, 3, 1, 1, 0, 256, 320, 256, 320, GEN) == 0
    assert lib.pd_conv2d_uses_x3(16 * 16 * 20, 256, 512, 3, 3, 1, 1, 0, 0, 0, 0, 0, X3) == 1                # forced: 160 tiles of 128 rows
    assert lib.pd_conv2d_uses_x3(M16, 32, 96, 3, 3, 1, 1, 0, 0, 0, 0, 0, X3) == 0                           # ... but never a shape the kernel cannot run
    # workspace of the weight gradient follows the same flags (the two families plan different slice counts)
    assert lib.pd_conv2d_wgrad_workspace(M16, 64, 576, AUTO) != lib.pd_conv2d_wgrad_workspace(M16, 64, 576, FP32)


def test_conv_flags_are_validated():
    """Unknown bits, or fp32-MFMA and bf16-split requested together: PD_EINVAL before anything is launched (null tensors
    would be the next complaint)."""
    lib = _lib.lib
    for bad in (64, 1 | 2, 0x80000000):
        rc = lib.pd_conv2d(8, 8, None, None, 8, None, 1, 4, 4, 4, 64, 16, 4, 1, 4, 4, 4, 1, 1, 1, 0, 0, 0, 0, 0.0, 1.0, 4, bad, None)
        assert rc == -22 and b"flags" in lib.pd_last_error(), bad
        rc = lib.pd_conv2d_wgrad(8, 8, 8, None, 8, 1 << 20, 1, 4, 4, 4, 64, 16, 4, 1, 4, 4, 4, 1, 1, 1, 0, 0, 0, 0.0, 1.0, 4, 0, bad, None)
        assert rc == -22 and b"flags" in lib.pd_last_error(), bad


def test_library_reads_no_environment_variable():
    """no global mutable state -- the kernel family is an argument.  The shared object must not even import
    getenv (the Python layer owns the PD_* knobs and reads them once, at import)."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.lib.path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms, [l for l in syms.splitlines() if "getenv" in l]


def test_toolchain_is_the_pinned_one():
    """csrc/TOOLCHAIN.txt pins the hipcc the inline-asm LDS-DMA paths of csrc/conv.hip were validated on (an asm statement
    that overwrites M0 relies on the backend treating M0 as reserved; one statement carries no "memory" clobber by design)."""
    import shutil
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc on this machine")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert "AMD clang version" in g.toolchain_check()


def test_routing_queries_over_a_shape_sweep():
    """pd_conv2d_uses_x3 / pd_conv2d_wgrad_uses_x3 / pd_conv2d_wgrad_workspace are pure host functions of the shape: over a sweep
    of layer shapes (the network's and perturbations: odd planes, channel counts off the kernels' grids, both filter sizes,
    every flags word) they return one of their documented codes, the workspace always covers at least one partial tile, and
    the rolling-row / halo weight-gradient kernels are only ever chosen together with a workspace that holds their slices."""
    import itertools
    lib = _lib.lib
    planes = [(256, 320), (128, 160), (64, 80), (32, 40), (16, 20), (34, 42), (150, 150), (8, 12), (512, 640)]
    chans = [(64, 64), (128, 128), (256, 512), (512, 512), (96, 32), (32, 96), (64, 32), (36, 64), (12, 64), (8, 64), (16, 16), (48, 64), (100, 64)]
    for (H, W), (C, Co), k, N, mode, fl in itertools.product(planes, chans, (1, 3, 4, 5), (1, 16), (0, 1, 2), (0, 1, 2, 4, 8, 16, 32)):
        if mode == 1 and k != 3:
            continue
        M = N * H * W
        pad = k // 2
        a = lib.pd_conv2d_uses_x3(M, Co, C, k, k, 1, pad, mode, 0, 0, H, W, fl)
        assert a in (0, 1, 2, 3), (H, W, C, Co, k, N, mode, fl, a)
        if fl & 1:
            assert a == 0
        if mode != 2:
            b = lib.pd_conv2d_wgrad_uses_x3(M, Co, C, k, k, 1, pad, mode, H, W, H, W, fl)
            ws = lib.pd_conv2d_wgrad_workspace(M, Co, k * k * C, fl)
            per = 4 * (Co * k * k * C + Co)
            assert b in (0, 1, 2, 3) and ws >= per, (H, W, C, Co, k, N, mode, fl, b, ws)
            if fl & 1:
                assert b == 0
            if b == 3:      # rolling rows: 3x3, whole 64-channel blocks, at least 384 workgroups' worth of slices in the workspace
                assert k == 3 and C % 64 == 0 and Co % 64 == 0 and not (fl & 32)
                assert (ws // per) * (C // 64) * (Co // 64) >= 384
