"""Low-level tensor ops on the HIP library (no autograd here; see functional.py).

Activations are fp32, logically NCHW, physically NHWC (torch.channels_last); weights are
[Cout,Cin,kh,kw] logically and [Cout][kh][kw][Cin] physically (channels_last), i.e. exactly the
operand layout of the implicit-GEMM kernels -- no repacking on the forward path.
"""
import os

import torch

from ._lib import lib, check, ptr, stream_ptr

MODE_ZERO, MODE_REFLECT, MODE_TRANSPOSED = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3
CL = torch.channels_last
PROFILE = None      # bench.py sets this to a list: (kernel label, algorithmic flops, start event, end event, shape)

# Kernel family / arithmetic of the convolutions: the `flags` word of pd_conv2d* / pd_conv2d_wgrad (include/polardepth.h).
# The library reads no environment variable; this module owns the knobs and reads them ONCE, at import:
#   PD_CONV_X3=0    forward / data gradient on the fp32 MFMA only      PD_WGRAD_X3C=0   weight gradients likewise
#   PD_CONV_HALO=0  bf16-split forward / data gradient through conv_igemm_x3_kernel also where the halo-tile kernel fits
#   PD_WGRAD_ROLL=0 3x3 weight gradients with one filter row per workgroup also where the rolling-row kernel fits
# Tests and bench.py switch families in-process by assigning CONV_FLAGS / WGRAD_FLAGS (or `with conv_flags(...)`).
CONV_AUTO, CONV_FP32_MFMA, CONV_BF16X3, CONV_WGRAD_SPLIT_IN_REGS, CONV_GENERAL_KERNELS, CONV_X3_IM2COL = 0, 1, 2, 4, 8, 16
CONV_WGRAD_ROW_WORKGROUPS = 32        # one filter row per workgroup also where the rolling-row weight-gradient kernel fits (A/B, tests)
CONV_FLAGS = (CONV_FP32_MFMA if os.environ.get("PD_CONV_X3", "1") == "0" else       # pd_conv2d, _add, _rect
              CONV_X3_IM2COL if os.environ.get("PD_CONV_HALO", "1") == "0" else CONV_AUTO)   # PD_CONV_HALO=0: the per-tap gather kernel everywhere
WGRAD_FLAGS = (CONV_FP32_MFMA if os.environ.get("PD_WGRAD_X3C", "1") == "0" else     # pd_conv2d_wgrad
               CONV_X3_IM2COL if os.environ.get("PD_CONV_HALO", "1") == "0" else
               CONV_WGRAD_ROW_WORKGROUPS if os.environ.get("PD_WGRAD_ROLL", "1") == "0" else CONV_AUTO)   # PD_WGRAD_ROLL=0: one filter row per workgroup


class conv_flags:
    """Context manager: run the enclosed convolutions with the given flags (None = leave that entry point's word alone)."""

    def __init__(self, conv=None, wgrad=None):
        self.new = (conv, wgrad)

    def __enter__(self):
        global CONV_FLAGS, WGRAD_FLAGS
        self.old = (CONV_FLAGS, WGRAD_FLAGS)
        if self.new[0] is not None:
            CONV_FLAGS = int(self.new[0])
        if self.new[1] is not None:
            WGRAD_FLAGS = int(self.new[1])
        return self

    def __exit__(self, *exc):
        global CONV_FLAGS, WGRAD_FLAGS
        CONV_FLAGS, WGRAD_FLAGS = self.old
        return False


def _profiled(label, flops, fn, shape=None):
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((label, flops, e0, e1, shape))
    return r


def _igemm_label(M, Co, vec, kind, C=0, KH=1, KW=1, stride=1, pad=0, mode=0, act=ACT_NONE, out_scale=False, out_hw=(0, 0)):
    """Profiler label = the kernel family pd_conv2d launches for this call (same rule as launch_conv in conv.hip)."""
    rb = lib.pd_conv2d_uses_x3(M, Co, C, KH, KW, stride, pad, mode, act, int(out_scale), out_hw[0], out_hw[1], CONV_FLAGS) if vec else 0
    if rb == 3:
        return "conv_halo_x3_kernel<8x32,64>"
    if rb:
        return f"conv_igemm_x3_kernel<{128 * rb},64>"
    bm = lib.pd_conv2d_tile_m(M, Co)
    bn = 64 if Co > 32 else (32 if Co > 16 else 16)
    uni = (vec and not (CONV_FLAGS & CONV_GENERAL_KERNELS) and bn >= 32 and C % 32 == 0 and C > 0 and KH * KW <= 31 and pad < KH and pad < KW and
           (mode in (MODE_ZERO, MODE_REFLECT) or (mode == MODE_TRANSPOSED and stride == 1)))
    if uni:
        return f"conv_igemm_uni_kernel<{bm},{bn}>"
    return f"conv_igemm_kernel<{bm},{bn},{'vec' if vec else 'scalar'}>"


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("polardepth ops need CUDA(HIP) tensors; there is no CPU fallback")


def empty_nhwc(n, c, h, w, device):
    return torch.empty((n, c, h, w), dtype=torch.float32, device=device, memory_format=CL)


def is_nhwc(t):
    n, c, h, w = t.shape
    return t.stride() == (h * w * c, 1, w * c, c) or t.is_contiguous(memory_format=CL)


def as_nhwc(t):
    return t if is_nhwc(t) else t.contiguous(memory_format=CL)


def weight_cl(w):
    """Weight in [Cout][kh][kw][Cin] storage."""
    return w if w.is_contiguous(memory_format=CL) else w.contiguous(memory_format=CL)


def conv_out_size(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, mode=MODE_ZERO, act=ACT_NONE, want_stats=False,
               affine=None, out=None, out_hw=None, out_scale=None, alg_k=None):
    """y = act(conv(x, w) [* out_scale] + bias).  x: [N,C,H,W] any strides; returns channels_last [N,Co,Ho,Wo].

    out_scale: optional fp32 [Co] multiplier of the accumulator (an inference-mode BatchNorm folded into the epilogue).

    want_stats -> also returns the per-workgroup BatchNorm partials [rows, Co, 2] (sum, sum of squares
    of the pre-activation output).  affine=(sub, div) applies (x-sub)/div to in-bounds input taps.
    out: optional pre-allocated NHWC tensor whose channel slice [:, c0:c0+Co] receives the result
    (pass the sliced view; its channel stride must be 1).
    alg_k: algorithmic contraction length for the profiler's FLOP count when it differs from C*KH*KW (the
    space-to-depth stems execute K = 64*C for an algorithmic 7x7 filter: 49*C).
    """
    _require_cuda(x, w, bias, out_scale)
    global FWD_EPOCH
    FWD_EPOCH += 1
    N, C, H, W = x.shape
    Co, Ci, KH, KW = w.shape
    assert Ci == C, f"channel mismatch {Ci} vs {C}"
    w = weight_cl(w)
    Ho, Wo = conv_out_size(H, KH, stride, pad), conv_out_size(W, KW, stride, pad)
    if out_hw is not None:          # leading part of the output grid only (asymmetric padding)
        assert out_hw[0] <= Ho and out_hw[1] <= Wo
        Ho, Wo = out_hw
    if out is None:
        out = empty_nhwc(N, Co, Ho, Wo, x.device)
    assert out.shape == (N, Co, Ho, Wo) and out.stride(1) == 1
    ldy = out.stride(3)
    assert out.stride(2) == Wo * ldy and out.stride(0) == Ho * Wo * ldy
    stats = None
    if want_stats:
        rows = lib.pd_conv2d_stats_rows(N * Ho * Wo, Co)
        stats = torch.empty((rows, Co, 2), dtype=torch.float32, device=x.device)
    sub, div = (affine if affine is not None else (0.0, 1.0))
    sN, sC, sH, sW = x.stride()
    vec = C % 4 == 0 and sC == 1 and affine is None
    if (USE_CONV16 and mode == MODE_REFLECT and Co == 16 and C in (16, 32) and KH == 3 and KW == 3 and stride == 1 and pad == 1
            and vec and not want_stats and out_scale is None and act in (ACT_NONE, ACT_ELU) and out_hw is None
            and H >= 2 and W >= 2 and sN % 4 == 0 and sH % 4 == 0 and sW % 4 == 0):
        # 16-channel decoder tail: halo-tile kernel (every input pixel staged once instead of once per tap)
        _profiled("conv16_halo_kernel", 2.0 * N * Ho * Wo * Co * C * 9,
                  lambda: check(lib.pd_conv16(ptr(x), ptr(w), ptr(bias), ptr(out), N, H, W, C, sN, sH, sW, Ho, Wo, Co, ldy,
                                              0, act, stream_ptr()), "pd_conv16"),
                  shape=("fwd", N, C, H, W, Co, KH, stride, mode))
        return out
    _profiled(_igemm_label(N * Ho * Wo, Co, vec, "fwd", C, KH, KW, stride, pad, mode, act, out_scale is not None, (Ho, Wo)), 2.0 * N * Ho * Wo * Co * (alg_k if alg_k is not None else C * KH * KW),
              lambda: check(lib.pd_conv2d(ptr(x), ptr(w), ptr(bias), ptr(out_scale), ptr(out), ptr(stats), N, H, W, C, sN, sH, sW, sC,
                                          Ho, Wo, Co, KH, KW, stride, pad, mode, act, int(affine is not None), sub,
                                          div, ldy, CONV_FLAGS, stream_ptr()), "pd_conv2d"),
              shape=("fwd", N, C, H, W, Co, KH, stride, mode))
    return (out, stats) if want_stats else out


# objects with .transposed(w) -> tensor | None: a parameter store hands out the data-gradient operands of ALL its
# convolution weights from one batched launch per backward pass (engine.ParamStore registers itself here).  The copies
# are valid until the next forward convolution: FWD_EPOCH counts conv2d_fwd calls, and weights cannot change between a
# forward pass and its backward pass without invalidating the gradients anyway -- no reliance on version counters
# (writes through ``p.data`` or raw pointers do not bump any).
USE_S2_PHASES = os.environ.get("PD_S2_PHASES", "1") != "0"   # stride-2 data gradient as four parity-class sub-convolutions
USE_CONV16 = os.environ.get("PD_CONV16", "1") != "0"     # halo-tile kernel for the 16-channel decoder tail
WT_PROVIDERS = []
FWD_EPOCH = 0


def weight_transposed(w):
    """[Co,Ci,kh,kw] (channels_last) -> operand of the data-gradient GEMM: [Ci][kh][kw][Co] storage,
    returned as a logical [Ci,Co,kh,kw] channels_last tensor."""
    Co, Ci, KH, KW = w.shape
    w = weight_cl(w)
    for prov in WT_PROVIDERS:
        wt = prov.transposed(w)
        if wt is not None:
            return wt
    wt = torch.empty((Ci, Co, KH, KW), dtype=torch.float32, device=w.device, memory_format=CL)
    check(lib.pd_weight_transpose(ptr(w), ptr(wt), Co, KH * KW, Ci, stream_ptr()), "pd_weight_transpose")
    return wt


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0, wt=None, addend=None):
    """dX of a zero-padded convolution: transposed convolution of dy (NHWC) with w.

    addend: optional tensor of dX's shape (channel stride 1) that is added in the kernel's epilogue -- the gradient of a
    residual block's skip connection, which autograd would otherwise add in a separate pass."""
    _require_cuda(dy, w)
    dy = as_nhwc(dy)
    N, Co, Hy, Wy = dy.shape
    _, Ci, KH, KW = w.shape
    H, W = in_hw
    if wt is None:
        wt = weight_transposed(w)
    dx = empty_nhwc(N, Ci, H, W, dy.device)
    sN, sC, sH, sW = dy.stride()
    if (USE_S2_PHASES and not (CONV_FLAGS & CONV_GENERAL_KERNELS) and addend is None and stride == 2 and KH == 3 and KW == 3 and pad == 1 and H == 2 * Hy and W == 2 * Wy
            and Co % 32 == 0 and Ci % 4 == 0 and Ci > 16 and sC == 1 and sN % 4 == 0 and sH % 4 == 0 and sW % 4 == 0):
        # stride-2 data gradient by output parity: four stride-1 2x2 sub-filter launches + one interleave
        def _phases():
            wsub = torch.empty(9 * Ci * Co, dtype=torch.float32, device=dy.device)     # class filters 1x1, 1x2, 2x1, 2x2
            check(lib.pd_dgrad_s2_filters(ptr(wt), ptr(wsub), Ci, Co, stream_ptr()), "pd_dgrad_s2_filters")
            sub = torch.empty((4, N, Hy, Wy, Ci), dtype=torch.float32, device=dy.device)
            for c, off in enumerate((0, 1, 3, 5)):
                ph, pw = c >> 1, c & 1
                check(lib.pd_conv2d_rect(ptr(dy), wsub.data_ptr() + 4 * off * Ci * Co, ptr(sub[c]), N, Hy, Wy, Co, sN, sH, sW,
                                         sC, Hy, Wy, Ci, 1 + ph, 1 + pw, ph, pw, MODE_TRANSPOSED, Ci, CONV_FLAGS, stream_ptr()),
                      "pd_conv2d_rect(dgrad s2 phase)")
            check(lib.pd_interleave4(ptr(sub), ptr(dx), N, Hy, Wy, Ci, stream_ptr()), "pd_interleave4")
        _profiled("conv_dgrad_s2_phases", 2.0 * N * Hy * Wy * Co * Ci * KH * KW, _phases,
                  shape=("dgrad", N, Ci, H, W, Co, KH, stride, MODE_TRANSPOSED))
        return dx
    c16_mode = 1 if (pad == 0 and H == Hy + 2 and W == Wy + 2) else (2 if (pad == 1 and H == Hy and W == Wy) else 0)
    if (USE_CONV16 and addend is None and Co == 16 and Ci in (16, 32) and KH == 3 and KW == 3 and stride == 1 and c16_mode
            and sC == 1 and sN % 4 == 0 and sH % 4 == 0 and sW % 4 == 0 and Hy >= 2 and Wy >= 2):
        # data gradient of a 16-channel 3x3 conv (on the padded grid, or pad 1 on the same grid): halo-tile kernel
        _profiled("conv16_halo_kernel", 2.0 * N * Hy * Wy * Co * Ci * 9,
                  lambda: check(lib.pd_conv16(ptr(dy), ptr(wt), None, ptr(dx), N, Hy, Wy, Co, sN, sH, sW, H, W, Ci, Ci, c16_mode,
                                              0, stream_ptr()), "pd_conv16(dgrad)"),
                  shape=("dgrad", N, Ci, H, W, Co, KH, stride, MODE_TRANSPOSED))
        return dx
    if addend is not None:
        assert addend.shape == dx.shape and addend.stride(1) == 1 and addend.is_cuda
        ld_add = addend.stride(3)
        assert addend.stride(2) == W * ld_add and addend.stride(0) == H * W * ld_add
        _profiled(_igemm_label(N * H * W, Ci, True, "dgrad", Co, KH, KW, stride, pad, MODE_TRANSPOSED, out_hw=(H, W)),
                  2.0 * N * Hy * Wy * Co * Ci * KH * KW,
                  lambda: check(lib.pd_conv2d_add(ptr(dy), ptr(wt), ptr(addend), ld_add, ptr(dx), N, Hy, Wy, Co, sN, sH, sW,
                                                  sC, H, W, Ci, KH, KW, stride, pad, MODE_TRANSPOSED, Ci, CONV_FLAGS, stream_ptr()),
                                "pd_conv2d_add(dgrad)"),
                  shape=("dgrad", N, Ci, H, W, Co, KH, stride, MODE_TRANSPOSED))
        return dx
    # algorithmic flops of the data gradient = those of the forward conv it differentiates
    _profiled(_igemm_label(N * H * W, Ci, True, "dgrad", Co, KH, KW, stride, pad, MODE_TRANSPOSED, out_hw=(H, W)), 2.0 * N * Hy * Wy * Co * Ci * KH * KW,
              lambda: check(lib.pd_conv2d(ptr(dy), ptr(wt), None, None, ptr(dx), None, N, Hy, Wy, Co, sN, sH, sW, sC,
                                          H, W, Ci, KH, KW, stride, pad, MODE_TRANSPOSED, ACT_NONE, 0, 0.0, 1.0, Ci,
                                          CONV_FLAGS, stream_ptr()), "pd_conv2d(dgrad)"),
              shape=("dgrad", N, Ci, H, W, Co, KH, stride, MODE_TRANSPOSED))
    return dx


_ws_cache = {}


def _workspace(nbytes, device):
    """Scratch of the split reductions, one buffer per (device, stream): weight-gradient kernels run on a side
    stream next to main-stream users of the same scratch (gemm_tn in the materialised-score attention backward)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, mode=MODE_ZERO, affine=None, dw=None, dbias=None,
                 want_bias=False, accumulate=False, alg_k=None):
    """dW (channels_last [Co,Ci,kh,kw]) and optionally dbias of conv(x, w) given dy (NHWC)."""
    _require_cuda(x, dy)
    dy = as_nhwc(dy)
    N, C, H, W = x.shape
    Co, Ci, KH, KW = w_shape
    assert Ci == C
    Ho, Wo = dy.shape[2], dy.shape[3]
    if dw is None:
        dw = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device, memory_format=CL)
        accumulate = False
    assert dw.is_contiguous(memory_format=CL) or dw.numel() == dw.shape[0] * dw.shape[1]
    if want_bias and dbias is None:
        dbias = torch.empty(Co, dtype=torch.float32, device=x.device)
    M, K = N * Ho * Wo, KH * KW * C
    sN, sC, sH, sW = x.stride()
    if (USE_CONV16 and mode == MODE_REFLECT and Co == 16 and C in (16, 32) and KH == 3 and KW == 3 and stride == 1 and pad == 1
            and affine is None and sC == 1 and sN % 4 == 0 and sH % 4 == 0 and sW % 4 == 0 and H >= 2 and W >= 2
            and dy.stride(3) % 4 == 0):
        ws = _workspace(lib.pd_conv16_wgrad_workspace(C), x.device)
        _profiled("conv16_wgrad_kernel", 2.0 * M * Co * K,
                  lambda: check(lib.pd_conv16_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), ws.numel(), N, H, W, C,
                                                    sN, sH, sW, dy.stride(3), int(accumulate), stream_ptr()),
                                "pd_conv16_wgrad"),
                  shape=("wgrad", N, C, H, W, Co, KH, stride, mode))
        return (dw, dbias) if (want_bias or dbias is not None) else dw
    nbytes = lib.pd_conv2d_wgrad_workspace(M, Co, K, WGRAD_FLAGS)
    ws = _workspace(nbytes, x.device)
    sub, div = (affine if affine is not None else (0.0, 1.0))
    sN, sC, sH, sW = x.stride()
    x3c = (lib.pd_conv2d_wgrad_uses_x3(M, Co, C, KH, KW, stride, pad, mode, H, W, Ho, Wo, WGRAD_FLAGS)
           if (affine is None and sC == 1 and sN % 4 == 0 and sH % 4 == 0 and sW % 4 == 0 and dy.stride(3) % 4 == 0) else 0)
    _profiled("conv_wgrad_roll_x3_kernel" if x3c == 3 else "conv_wgrad_halo_x3_kernel" if x3c == 2 else "conv_wgrad_x3c_kernel" if x3c else "conv_wgrad_kernel",
              2.0 * M * Co * (alg_k if alg_k is not None else K),
              lambda: check(lib.pd_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), ws.numel(), N, H, W, C,
                                                sN, sH, sW, sC, Ho, Wo, Co, KH, KW, stride, pad, mode,
                                                int(affine is not None), sub, div, dy.stride(3), int(accumulate),
                                                WGRAD_FLAGS, stream_ptr()), "pd_conv2d_wgrad"),
              shape=("wgrad", N, C, H, W, Co, KH, stride, mode))
    return (dw, dbias) if (want_bias or dbias is not None) else dw


class _SSIMFn(torch.autograd.Function):
    """layers.SSIM (mode 0: the [N,C,H,W] loss map) and compute_reprojection_loss (mode 1: [N,1,H,W]) with their gradients
    w.r.t. both images (pd_ssim_fwd / pd_ssim_bwd)."""

    @staticmethod
    def forward(ctx, x, y, mode, no_ssim):
        x, y = x.float().contiguous(), y.float().contiguous()
        N, C, H, W = x.shape
        out = torch.empty((N, C if mode == 0 else 1, H, W), dtype=torch.float32, device=x.device)
        check(lib.pd_ssim_fwd(ptr(x), ptr(y), ptr(out), N, C, H, W, mode, int(no_ssim), stream_ptr()), "pd_ssim_fwd")
        ctx.args = (mode, int(no_ssim))
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        mode, no_ssim = ctx.args
        N, C, H, W = x.shape
        gout = gout.float().contiguous()
        ws = torch.empty(5 * x.numel(), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        check(lib.pd_ssim_bwd(ptr(x), ptr(y), ptr(gout), ptr(ws), ptr(gx), ptr(gy), N, C, H, W, mode, no_ssim, stream_ptr()),
              "pd_ssim_bwd")
        return gx, gy, None, None


def ssim(x, y):
    """layers.SSIM on the GPU: planar NCHW fp32 -> SSIM loss map of the same shape (differentiable in both images)."""
    _require_cuda(x, y)
    return _SSIMFn.apply(x, y, 0, False)


def reprojection_loss(pred, target, no_ssim=False):
    """trainer.py:1069-1081: [N,1,H,W] = 0.85 * mean_c SSIM(pred, target) + 0.15 * mean_c |target - pred| (differentiable)."""
    _require_cuda(pred, target)
    return _SSIMFn.apply(pred, target, 1, bool(no_ssim))


def depth_metrics(gt, pred, min_depth, max_depth, mask=None, mask_value=0):
    """Per-image depth metrics on the device: [N,8] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, count."""
    _require_cuda(gt, pred, mask)
    gt, pred = gt.float().contiguous(), pred.float().contiguous()
    N = gt.shape[0]
    P = gt.numel() // N
    if mask is not None:
        mask = mask.to(torch.int32).contiguous()
    ws = torch.empty(N * 64 * 8, dtype=torch.float32, device=gt.device)
    out = torch.empty((N, 8), dtype=torch.float32, device=gt.device)
    check(lib.pd_depth_metrics(ptr(gt), ptr(pred), ptr(mask), int(mask_value), ptr(ws), ptr(out), N, P,
                               float(min_depth), float(max_depth), stream_ptr()), "pd_depth_metrics")
    return out


# ------------------------------------------------------------------ 7x7/s2 stems as 4x4/s1 over space-to-depth
def s2d_input(x, affine=None):
    """[N,C,H,W] (any strides) -> channels_last [N,4C,H/2,W/2] with q = (dy*2+dx)*C + c; optional (x-sub)/div."""
    _require_cuda(x)
    N, C, H, W = x.shape
    out = empty_nhwc(N, 4 * C, H // 2, W // 2, x.device)
    sub, div = affine if affine is not None else (0.0, 1.0)
    sN, sC, sH, sW = x.stride()
    check(lib.pd_stem_s2d_input(ptr(x), ptr(out), N, C, H, W, sN, sC, sH, sW, int(affine is not None), sub, div,
                                stream_ptr()), "pd_stem_s2d_input")
    return out


def s2d_weight(w):
    """[Co,C,7,7] (channels_last) -> channels_last [Co,4C,4,4]."""
    Co, C, KH, KW = w.shape
    assert KH == 7 and KW == 7
    w = weight_cl(w)
    w2 = torch.empty((Co, 4 * C, 4, 4), dtype=torch.float32, device=w.device, memory_format=CL)
    check(lib.pd_stem_s2d_weight(ptr(w), ptr(w2), Co, C, stream_ptr()), "pd_stem_s2d_weight")
    return w2


def s2d_weight_grad(dw2, dw, accumulate=True):
    """dW2 [Co,4C,4,4] (channels_last) -> (+)= dW [Co,C,7,7] (channels_last)."""
    Co, C = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous(memory_format=CL) or C == 1
    check(lib.pd_stem_s2d_weight_grad(ptr(dw2), ptr(dw), Co, C, int(accumulate), stream_ptr()),
          "pd_stem_s2d_weight_grad")
    return dw


# ------------------------------------------------------------------ dense GEMMs on the conv kernels (attention)
def gemm_nt(a, b, out=None):
    """C[M,N] = A[M,K] @ B[N,K]^T (fp32 MFMA): pd_conv2d with a 1x1 filter, A as a [1,M,1,K] NHWC image."""
    _require_cuda(a, b)
    M, K = a.shape
    Nn, Kb = b.shape
    assert K == Kb and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=a.device)
    _profiled(_igemm_label(M, Nn, K % 4 == 0, "gemm", K), 2.0 * M * Nn * K,
              lambda: check(lib.pd_conv2d(ptr(a), ptr(b), None, None, ptr(out), None, 1, M, 1, K, M * K, K, K, 1,
                                          M, 1, Nn, 1, 1, 1, 0, MODE_ZERO, ACT_NONE, 0, 0.0, 1.0, out.stride(0),
                                          CONV_FLAGS, stream_ptr()), "pd_conv2d(gemm_nt)"))
    return out


def gemm_tn(dy, x):
    """C[N,K] = dY[M,N]^T @ X[M,K] (contraction over rows): pd_conv2d_wgrad with a 1x1 filter."""
    _require_cuda(dy, x)
    M, Nn = dy.shape
    Mx, K = x.shape
    assert M == Mx and dy.is_contiguous() and x.is_contiguous()
    out = torch.empty((Nn, K), dtype=torch.float32, device=x.device)
    nbytes = lib.pd_conv2d_wgrad_workspace(M, Nn, K, WGRAD_FLAGS)
    ws = _workspace(nbytes, x.device)
    _profiled("conv_wgrad_kernel", 2.0 * M * Nn * K,
              lambda: check(lib.pd_conv2d_wgrad(ptr(x), ptr(dy), ptr(out), None, ptr(ws), ws.numel(), 1, M, 1, K,
                                                M * K, K, K, 1, M, 1, Nn, 1, 1, 1, 0, MODE_ZERO, 0, 0.0, 1.0, Nn, 0,
                                                WGRAD_FLAGS, stream_ptr()), "pd_conv2d_wgrad(gemm_tn)"))
    return out


def transpose2d(a, out=None):
    """[R,C] -> [C,R] (pd_weight_transpose)."""
    R, C = a.shape
    if out is None:
        out = torch.empty((C, R), dtype=torch.float32, device=a.device)
    assert out.shape == (C, R) and out.is_contiguous() and a.is_contiguous()
    check(lib.pd_weight_transpose(ptr(a), ptr(out), R, 1, C, stream_ptr()), "pd_weight_transpose")
    return out


def softmax_rows_(x, scale):
    R, L = x.shape
    check(lib.pd_softmax_rows_fwd(ptr(x), R, L, float(scale), stream_ptr()), "pd_softmax_rows_fwd")
    return x


def softmax_rows_bwd_(p, dp, scale):
    R, L = p.shape
    check(lib.pd_softmax_rows_bwd(ptr(p), ptr(dp), R, L, float(scale), stream_ptr()), "pd_softmax_rows_bwd")
    return dp
