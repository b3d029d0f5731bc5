"""K2 (pd_conv2d / pd_conv2d_wgrad) vs a plain PyTorch fp32 CPU reference of the same op.

Tolerance: fp32 MFMA accumulates like an fmaf chain in a fixed k order (exact fp32), the CPU
reference in another order; |err| <= ~1e-6 * sum|a*b|.  Checked as rtol 2e-5 on the output scale."""
import pytest
import torch
import torch.nn.functional as F

from polardepth import ops

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=2e-5):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


CASES = [
    # N, C, H, W, Co, k, s, p, mode
    (2, 64, 16, 20, 64, 3, 1, 1, 0),      # ResidualBlock conv
    (2, 64, 16, 20, 64, 5, 1, 2, 0),      # Conv2 5x5
    (1, 128, 9, 11, 256, 5, 1, 2, 0),     # joint Conv1, odd sizes, M tail
    (2, 192, 8, 10, 256, 1, 1, 0, 0),     # fc1 1x1
    (2, 64, 16, 20, 128, 3, 2, 1, 0),     # resnet layer2 3x3 s2
    (2, 64, 16, 20, 128, 1, 2, 0, 0),     # resnet downsample 1x1 s2
    (2, 2, 32, 40, 64, 7, 2, 3, 0),       # XOLP stem (scalar gather)
    (1, 9, 32, 40, 64, 7, 2, 3, 0),       # normals stem
    (2, 96, 16, 20, 32, 3, 1, 1, 1),      # decoder upconv(1,1), reflect
    (2, 16, 32, 40, 16, 3, 1, 1, 1),      # decoder tail Cin=16, Cout=16
    (2, 32, 16, 20, 1, 3, 1, 1, 1),       # dispconv Cout=1
    (3, 512, 4, 5, 512, 3, 1, 1, 0),      # deep layer, tiny M
]


def _ref_conv(x, w, b, s, p, mode):
    if mode == 1:
        x = F.pad(x, (p, p, p, p), mode="reflect")
        p = 0
    return F.conv2d(x, w, b, stride=s, padding=p)


@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad(case):
    N, C, H, W, Co, k, s, p, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = _ref_conv(xr, wr, br, s, p, mode)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)

    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    y, stats = ops.conv2d_fwd(xd, wd, b.cuda(), stride=s, pad=p, mode=mode, want_stats=True)
    assert y.is_contiguous(memory_format=torch.channels_last)
    _close(y.cpu(), ref.detach())
    # BatchNorm partials: column sums / sums of squares of the output
    tot = stats.double().sum(0).cpu()
    rf = ref.detach().double()
    _close(tot[:, 0], rf.sum((0, 2, 3)), 1e-5)
    _close(tot[:, 1], (rf ** 2).sum((0, 2, 3)), 1e-5)

    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dw, db = ops.conv2d_wgrad(xd, dyd, w.shape, stride=s, pad=p, mode=mode, want_bias=True)
    _close(dw.cpu(), wr.grad)
    _close(db.cpu(), br.grad)
    if mode == 0:
        dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p)
        _close(dx.cpu(), xr.grad)


UNI_CASES = [
    # N, C, H, W, Co, k, s, p -- shapes the uniform-tap forward / data-gradient kernels and the scalar-pixel
    # weight-gradient kernel take (C % 32 == 0, zero padding, no bias): full tiles (transposed lean epilogue), pixel
    # tails, tiles spanning images, stride 2, half-empty k tiles (K = 576, 288), several Cout tiles, pad 0 and 1x1
    (2, 64, 32, 40, 64, 3, 1, 1),
    (1, 64, 20, 28, 64, 5, 1, 2),
    (2, 128, 16, 20, 128, 3, 1, 1),
    (2, 64, 32, 40, 128, 3, 2, 1),
    (3, 32, 14, 14, 64, 3, 1, 1),
    (2, 96, 16, 20, 64, 3, 1, 1),
    (1, 256, 8, 16, 256, 1, 1, 0),
    (2, 64, 18, 22, 64, 3, 1, 0),
    (5, 64, 6, 16, 64, 3, 1, 1),
    (1, 64, 64, 48, 32, 3, 1, 1),
]


@pytest.mark.parametrize("case", UNI_CASES)
def test_uniform_tap_and_scalar_pixel_kernels(case):
    N, C, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    y, stats = ops.conv2d_fwd(xd, wd, None, stride=s, pad=p, want_stats=True)
    _close(y.cpu(), ref.detach())
    tot = stats.double().sum(0).cpu()
    rf = ref.detach().double()
    _close(tot[:, 0], rf.sum((0, 2, 3)), 1e-5)
    _close(tot[:, 1], (rf ** 2).sum((0, 2, 3)), 1e-5)
    _close(ops.conv2d_fwd(xd, wd, None, stride=s, pad=p).cpu(), ref.detach())       # without the statistics
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dw = ops.conv2d_wgrad(xd, dyd, w.shape, stride=s, pad=p)
    _close(dw.cpu(), wr.grad)
    dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p)
    _close(dx.cpu(), xr.grad)


@pytest.mark.parametrize("case", [(2, 128, 16, 20, 64), (1, 96, 12, 14, 32), (2, 64, 8, 8, 64), (1, 256, 5, 4, 128),
                                  (3, 32, 2, 3, 64), (1, 64, 64, 48, 64), (2, 96, 13, 14, 64), (1, 512, 3, 16, 256)])
def test_reflection_padded_conv_on_the_uniform_tap_kernel(case):
    """ReflectionPad2d(1) + Conv3x3 with C % 32 == 0 (the decoder's upconvs): border rows take their reflected offsets
    per tap; tiny images where both borders fold onto neighbouring pixels; with and without bias / ELU."""
    N, C, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wr, br)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    ref = ref.detach() - b[None, :, None, None]
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    _close(ops.conv2d_fwd(xd, wd, None, stride=1, pad=1, mode=1).cpu(), ref)
    _close(ops.conv2d_fwd(xd, wd, b.cuda(), stride=1, pad=1, mode=1, act=ops.ACT_ELU).cpu(), F.elu(ref + b[None, :, None, None]), 3e-5)
    # weight / bias gradient (scalar-pixel kernel with reflect corrections when Cout >= 33 and W is even and >= 14)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dw, db = ops.conv2d_wgrad(xd, dyd, w.shape, stride=1, pad=1, mode=1, want_bias=True)
    _close(dw.cpu(), wr.grad)
    _close(db.cpu(), br.grad)
    _close(ops.conv2d_wgrad(xd, dyd, w.shape, stride=1, pad=1, mode=1).cpu(), wr.grad)
    # data gradient through the autograd node: zero-padding (pad 1) data gradient + the folded border strips
    # (pd_reflect_dgrad_border), and the older padded-grid + pd_reflect_fold route, against torch
    from polardepth import functional as PF
    xr = x.clone().requires_grad_(True)
    (F.elu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), w, b)) * dy).sum().backward()
    conv = torch.nn.Conv2d(C, Co, 3).cuda()
    conv.weight.data = wd.clone(); conv.bias.data = b.cuda()
    for border in (True, False):
        PF.USE_REFLECT_BORDER = border
        xc = xd.clone().requires_grad_(True)
        (PF.reflect_conv_act(xc, conv, ops.ACT_ELU) * dyd).sum().backward()
        PF.sync_wgrad_stream()
        _close(xc.grad.cpu(), xr.grad, 3e-5)
    PF.USE_REFLECT_BORDER = True


@pytest.mark.parametrize("case", [(2, 64, 32, 40, 64, 3, 1, 1), (1, 64, 20, 28, 64, 5, 1, 2), (2, 16, 9, 11, 24, 3, 1, 1),
                                  (2, 64, 16, 20, 128, 3, 2, 1)])
def test_data_gradient_with_addend_in_the_epilogue(case):
    """pd_conv2d_add: dX = dgrad(dY, W) + addend in one kernel (full tiles: transposed epilogue; tails and the
    general kernel: element-wise), for a separate addend and for a strided view of a wider buffer."""
    N, C, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    ref = F.conv2d(x, w, None, stride=s, padding=p)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    add = torch.randn(N, C, H, W, generator=g)
    expect = x.grad + add
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    addd = add.cuda().contiguous(memory_format=torch.channels_last)
    dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p, addend=addd)
    _close(dx.cpu(), expect)
    assert torch.equal(addd.cpu(), add)                       # the addend itself is left alone
    wide = torch.zeros(N, C + 8, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    wide[:, 4:4 + C] = addd
    dx2 = ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p, addend=wide[:, 4:4 + C])
    _close(dx2.cpu(), expect)


@pytest.mark.parametrize("case", [(2, 64, 32, 40, 128), (1, 32, 6, 10, 64), (2, 64, 16, 20, 32), (3, 32, 4, 4, 32),
                                  (1, 128, 34, 18, 64)])
def test_stride2_data_gradient_by_output_parity(case):
    """3x3 / stride 2 / pad 1 data gradient as four exact sub-filter launches (pd_dgrad_s2_filters + pd_conv2d_rect:
    1x1, 1x2, 2x1, 2x2 taps with their own row / column padding) + pd_interleave4, against autograd on the CPU and
    against the masked transposed gather of the same library."""
    N, C, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    assert ops.USE_S2_PHASES
    ops.PROFILE = []
    dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=2, pad=1)
    labels = [p[0] for p in ops.PROFILE]
    ops.PROFILE = None
    assert labels == ["conv_dgrad_s2_phases"], labels
    _close(dx.cpu(), x.grad)
    ops.USE_S2_PHASES = False
    try:
        dx2 = ops.conv2d_dgrad(dyd, wd, (H, W), stride=2, pad=1)
    finally:
        ops.USE_S2_PHASES = True
    _close(dx2.cpu(), x.grad)


@pytest.mark.parametrize("case", [(2, 16, 12, 16, 32), (1, 8, 6, 10, 64), (2, 4, 8, 8, 32)])
def test_stride2_data_gradient_with_few_input_channels_takes_the_general_kernel(case):
    """Ci <= 16: the 16-wide tile has no uniform-tap kernel for the rectangular sub-filters, so the parity split must not
    be chosen (it used to raise PD_EINVAL from pd_conv2d_rect); the masked transposed gather serves these shapes."""
    N, C, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case) + 9)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    ops.PROFILE = []
    dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=2, pad=1)
    labels = [p[0] for p in ops.PROFILE]
    ops.PROFILE = None
    assert labels and labels != ["conv_dgrad_s2_phases"], labels
    _close(dx.cpu(), x.grad)


@pytest.mark.parametrize("geom", [(1, 2, 0, 1), (2, 1, 1, 0), (2, 3, 0, 2), (3, 1, 1, 0), (1, 1, 0, 0)])
def test_rectangular_filter_with_separate_padding(geom):
    """pd_conv2d_rect, mode 0: KH x KW filter with its own row / column zero padding (uniform-tap kernel) against
    F.conv2d(padding=(ph, pw)); mode 2 is covered by the parity-split stride-2 data gradient."""
    from polardepth._lib import lib, check, ptr
    KH, KW, ph, pw = geom
    N, C, H, W, Co = 2, 64, 12, 20, 64
    g = torch.Generator().manual_seed(KH * 10 + KW)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, KH, KW, generator=g) / (C * KH * KW) ** 0.5
    ref = F.conv2d(x, w, None, stride=1, padding=(ph, pw))
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    if KH * KW == 1:
        wd = wd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty(N, Co, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last)
    sN, sC, sH, sW = xd.stride()
    check(lib.pd_conv2d_rect(ptr(xd), ptr(wd), ptr(y), N, H, W, C, sN, sH, sW, sC, Ho, Wo, Co, KH, KW, ph, pw, 0, Co, 0, None),
          "pd_conv2d_rect")
    torch.cuda.synchronize()
    _close(y.cpu(), ref)


@pytest.mark.parametrize("case", [(2, 16, 32, 40), (1, 32, 16, 20), (2, 16, 9, 35), (1, 32, 13, 7), (3, 16, 2, 2),
                                  (1, 16, 64, 96), (1, 32, 3, 70)])
def test_sixteen_channel_tail_halo_kernel(case):
    """pd_conv16: ReflectionPad2d(1) + Conv3x3 with 16 output channels (+ bias, ELU) from a halo tile in LDS, and its data
    gradient on the padded grid (through ReflectConvActFn: pd_conv16 mode 1 + pd_reflect_fold), against autograd
    through a PyTorch fp32 CPU reference; tiles that overhang the image, images smaller than a tile, 2x2."""
    from polardepth import functional as PF
    N, C, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(N, C, H, W, generator=g)
    conv = torch.nn.Conv2d(C, 16, 3)
    conv.weight.data = torch.randn(16, C, 3, 3, generator=g) / (C * 9) ** 0.5
    conv.bias.data = torch.randn(16, generator=g) * 0.3
    gy = torch.randn(N, 16, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.elu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), conv.weight, conv.bias))
    (yr * gy).sum().backward()
    ref = (yr.detach(), xr.grad, conv.weight.grad.clone(), conv.bias.grad.clone())
    conv.weight.grad = conv.bias.grad = None
    assert ops.USE_CONV16
    conv = conv.cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ops.PROFILE = []
    y = PF.reflect_conv_act(xc, conv, ops.ACT_ELU)
    (y * gy.cuda()).sum().backward()
    PF.sync_wgrad_stream()
    labels = [p[0] for p in ops.PROFILE]
    ops.PROFILE = None
    assert labels.count("conv16_halo_kernel") == 2, labels       # forward and data gradient took the halo kernel
    assert labels.count("conv16_wgrad_kernel") == 1, labels      # ... and the weight / bias gradient its sibling
    _close(y.detach().cpu(), ref[0], 3e-5)
    _close(xc.grad.cpu(), ref[1], 3e-5)
    _close(conv.weight.grad.cpu(), ref[2], 3e-5)
    _close(conv.bias.grad.cpu(), ref[3], 3e-5)
    # without the activation, into a channel slice of a wider buffer (row stride != 16)
    buf = torch.zeros(N, 48, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    ops.conv2d_fwd(xc.detach(), conv.weight.data, conv.bias.data, stride=1, pad=1, mode=1, out=buf[:, 16:32])
    _close(buf[:, 16:32].cpu(), F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), conv.weight.data.cpu(), conv.bias.data.cpu()), 3e-5)
    assert buf[:, :16].abs().max().item() == 0 and buf[:, 32:].abs().max().item() == 0


def test_conv_nchw_input_with_affine_and_activations():
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 24, 32, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    ref = F.conv2d((x - 0.45) / 0.225, w, None, stride=2, padding=3)
    xd = x.cuda()                                   # NCHW strides, scalar gather path
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(xd, wd, None, stride=2, pad=3, affine=(0.45, 0.225))
    _close(y.cpu(), ref)
    dy = torch.randn(ref.shape, generator=g)
    wr = w.clone().requires_grad_(True)
    F.conv2d((x - 0.45) / 0.225, wr, None, stride=2, padding=3).backward(dy)
    dw = ops.conv2d_wgrad(xd, dy.cuda(), w.shape, stride=2, pad=3, affine=(0.45, 0.225))
    _close(dw.cpu(), wr.grad)
    for act, fn in ((ops.ACT_RELU, F.relu), (ops.ACT_ELU, F.elu), (ops.ACT_SIGMOID, torch.sigmoid)):
        ya = ops.conv2d_fwd(xd, wd, None, stride=2, pad=3, affine=(0.45, 0.225), act=act)
        _close(ya.cpu(), fn(ref), 3e-5)


def test_conv_writes_into_concat_slice_and_accumulates_wgrad():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 8, 12, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    buf = torch.zeros(2, 192, 8, 12, device="cuda").contiguous(memory_format=torch.channels_last)
    ops.conv2d_fwd(xd, wd, None, stride=1, pad=1, out=buf[:, 128:192])
    ref = F.conv2d(x, w, None, padding=1)
    _close(buf[:, 128:192].cpu(), ref)
    assert buf[:, :128].abs().max().item() == 0
    dy = torch.randn(ref.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    dw = ops.conv2d_wgrad(xd, dy, w.shape, stride=1, pad=1)
    dw2 = ops.conv2d_wgrad(xd, dy, w.shape, stride=1, pad=1, dw=dw.clone(), accumulate=True)
    _close(dw2.cpu(), 2 * dw.cpu(), 1e-6)


@pytest.mark.parametrize("C", [2, 3, 9])
def test_stem_as_space_to_depth_conv(C):
    """7x7/s2/p3 stem == 4x4/s1/p2 conv over the space-to-depth input with regrouped weights (fwd + wgrad)."""
    g = torch.Generator().manual_seed(C)
    x = torch.rand(2, C, 32, 48, generator=g)
    w = torch.randn(64, C, 7, 7, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d((x - 0.45) / 0.225, wr, b, stride=2, padding=3)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.cuda()
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    x2 = ops.s2d_input(xd, (0.45, 0.225))
    w2 = ops.s2d_weight(wd)
    assert x2.shape == (2, 4 * C, 16, 24) and w2.shape == (64, 4 * C, 4, 4)
    y = ops.conv2d_fwd(x2, w2, b.cuda(), stride=1, pad=2, out_hw=(16, 24))
    _close(y.cpu(), ref.detach())
    dw2 = ops.conv2d_wgrad(x2, dy.cuda().contiguous(memory_format=torch.channels_last), (64, 4 * C, 4, 4), stride=1, pad=2)
    dw = torch.zeros_like(wd)
    ops.s2d_weight_grad(dw2, dw, accumulate=True)
    _close(dw.cpu(), wr.grad)


@pytest.mark.parametrize("case", [(2, 3, 8, 140), (1, 9, 4, 128), (3, 2, 6, 20), (1, 3, 2, 260)])
def test_space_to_depth_input_layouts(case):
    """pd_stem_s2d_input: the LDS-tiled kernel (row-contiguous planes; full and partial 64-pixel tiles) and the
    element-wise kernel (any strides) against the permutation written in PyTorch -- bit-exact, with and without the
    input normalisation."""
    N, C, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.rand(N, C, H, W, generator=g)

    def expect(t):
        return t.reshape(N, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 4 * C, H // 2, W // 2)

    for xd in (x.cuda(), x.cuda().contiguous(memory_format=torch.channels_last)):       # tile kernel / strided kernel
        assert torch.equal(ops.s2d_input(xd).cpu(), expect(x))
        assert torch.equal(ops.s2d_input(xd, (0.45, 0.225)).cpu(), expect((x - 0.45) / 0.225))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", [(2, 16, 7, 9), (1, 32, 2, 2), (3, 64, 3, 5), (1, 128, 6, 4), (2, 16, 33, 21), (2, 16, 19, 131),
                                  (1, 32, 64, 80)])
def test_disparity_head_direct_kernels(case, fused, monkeypatch):
    """pd_disphead_{fwd,bwd_data,bwd_weight} (sigmoid(Conv3x3) with one output channel, reflection padding and its
    gradient fold built in) vs autograd through a PyTorch fp32 CPU reference, including the smallest legal images
    (H or W of 2 and 3, where both borders fold onto the same pixel) and ragged pixel counts."""
    from polardepth import functional as PF
    monkeypatch.setattr(PF, "USE_DISPHEAD_FUSED", fused)     # pd_disphead_bwd, or pd_disphead_bwd_data + _bwd_weight
    N, C, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    conv = torch.nn.Conv2d(C, 1, 3)
    conv.weight.data = (torch.randn(1, C, 3, 3, generator=g) * 0.2)
    conv.bias.data = torch.randn(1, generator=g) * 0.1
    gy = torch.randn(N, 1, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.sigmoid(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), conv.weight, conv.bias))
    (yr * gy).sum().backward()
    ref = (yr.detach(), xr.grad, conv.weight.grad.clone(), conv.bias.grad.clone())
    conv.weight.grad = conv.bias.grad = None

    conv = conv.cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert PF.USE_DISP_HEADS
    y = PF.reflect_conv_act(xc, conv, ops.ACT_SIGMOID)
    assert y.grad_fn.__class__.__name__.startswith("DispHeadFn")
    (y * gy.cuda()).sum().backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()
    _close(y.cpu(), ref[0], 2e-6)
    _close(xc.grad.cpu(), ref[1])
    _close(conv.weight.grad.cpu(), ref[2])
    _close(conv.bias.grad.cpu(), ref[3])


X3_CASES = [
    # N, C, H, W, Co, k, s, p -- shapes of the bf16x3 kernel (>= 512 tiles of 256 x 64 or >= 320 of 128 x 64, C % 4 == 0,
    # Co % 64 == 0): the 64-channel 3x3 / 5x5 layers of the shallow encoders, a stride-2 layer, 16 | C only, two column
    # tiles, a tile spanning two images, a small-M layer with 64-row statistics tiles (M < 65536)
    (2, 64, 256, 320, 64, 3, 1, 1),
    (2, 64, 256, 320, 64, 5, 1, 2),
    (1, 64, 512, 640, 128, 3, 2, 1),
    (4, 48, 256, 160, 64, 3, 1, 1),
    (1, 32, 256, 320, 128, 3, 1, 1),
    (32, 64, 72, 60, 64, 3, 1, 1),
    (4, 64, 64, 80, 512, 3, 1, 1),
    (1, 64, 256, 320, 64, 3, 1, 1),        # 320 tiles of 256 rows: the 128-row tiles (one row block per wave), bias
    (4, 64, 64, 80, 256, 3, 1, 1),         # ... with 64-row statistics tiles (M = 20480)
    (2, 36, 256, 320, 64, 3, 1, 1),        # 36 channels (the space-to-depth normals stem): a partly empty third channel group
    (2, 8, 256, 320, 64, 3, 1, 1),         # 8 channels (XOLP stem): one half-empty group
    (2, 96, 256, 320, 32, 3, 1, 1),        # 32 output channels (decoder 96 -> 32): the halo kernel's 32-column workgroups;
                                           # its data gradient has 96 output columns = three of them
    (8, 64, 128, 160, 32, 3, 1, 1),        # ... 64 -> 32 @128x160
    (16, 128, 16, 20, 128, 3, 1, 1),       # a 16x20 plane (the 512-channel layers): weight gradient on 8 x 4 pixel tiles
    (8, 64, 32, 40, 512, 5, 1, 2),         # 5x5 on a 32x40 plane, 320 tiles of 64 columns: 640 halo workgroups of 32 columns (32 x 8 tiles)
]


@pytest.mark.parametrize("case", X3_CASES)
def test_bf16x3_kernel_keeps_fp32_accuracy(case, monkeypatch):
    """fp32 products as six bf16 MFMAs of a three-way split (conv_igemm_x3_kernel): compared with an fp64 convolution, next
    to the fp32-MFMA kernel on the same input.  Bar: the error of the split kernel is at most 1.5x the fp32 kernel's (both
    are accumulation-order noise of ~1e-6 of the output scale at K = 576 .. 1600), and both pass the suite's tolerance."""
    N, C, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    x[:, :, : H // 2] *= 37.0                                   # two scales in one tensor
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if sum(case) % 2 == 0 else None         # (the encoders' convolutions carry a bias)
    ref = F.conv2d(x.double(), w.double(), b.double() if b is not None else None, stride=s, padding=p)
    dy = torch.randn(ref.shape, generator=g)
    add = torch.randn(N, C, H, W, generator=g)
    ref_dx = None
    bd = b.cuda() if b is not None else None
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    addd = add.cuda().contiguous(memory_format=torch.channels_last)
    if s == 1:
        ref_dx = F.conv_transpose2d(dy.double(), w.double(), None, stride=1, padding=p) + add.double()
    scale = ref.abs().max().item()
    errs, stats, dxe = {}, {}, {}
    dwe = {}
    ref_dw = torch.autograd.grad(F.conv2d(x.double(), wv := w.double().requires_grad_(True), None, stride=s, padding=p), wv, dy.double())[0]
    dbe = {}
    ref_db = dy.double().sum((0, 2, 3))
    for knob in ("1", "0"):
        fl = ops.CONV_AUTO if knob == "1" else ops.CONV_FP32_MFMA      # the `flags` word of pd_conv2d* / pd_conv2d_wgrad
        monkeypatch.setattr(ops, "CONV_FLAGS", fl)
        monkeypatch.setattr(ops, "WGRAD_FLAGS", fl)        # weight gradient: every element split once (conv_wgrad_x3c_kernel)
        dw, db = ops.conv2d_wgrad(xd, dyd, w.shape, stride=s, pad=p, want_bias=True)
        dwe[knob] = (dw.cpu().double() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
        dbe[knob] = (db.cpu().double() - ref_db).abs().max().item() / ref_db.abs().max().item()
        if knob == "1":                                   # ... and the in-register split of the uniform-tap kernel
            monkeypatch.setattr(ops, "WGRAD_FLAGS", ops.CONV_WGRAD_SPLIT_IN_REGS)
            dw = ops.conv2d_wgrad(xd, dyd, w.shape, stride=s, pad=p)
            dwe["r"] = (dw.cpu().double() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
            monkeypatch.setattr(ops, "WGRAD_FLAGS", fl)
        y, st = ops.conv2d_fwd(xd, wd, bd, stride=s, pad=p, want_stats=True)
        errs[knob] = (y.cpu().double() - ref).abs().max().item() / scale
        stats[knob] = st.double().sum(0).cpu()
        if s == 1:
            dx = ops.conv2d_dgrad(dyd, wd, (H, W), stride=1, pad=p, addend=addd)
            dxe[knob] = (dx.cpu().double() - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    assert errs["1"] <= 5e-6 and errs["0"] <= 5e-6, errs
    assert errs["1"] <= 1.5 * errs["0"] + 1e-8, errs
    # mean of the output vs the reference's, in units of the output's rms (the bf16 MFMA aligns its addends by truncation:
    # a shift of ~ -1e-7 rms at K = 1600 that a sum over M outputs multiplies by M; see test_prodsize_gpu.py)
    M = ref.numel() // Co
    assert (stats["1"][:, 0] - ref.sum((0, 2, 3))).abs().max().item() / M <= 2.5e-7 * ref.pow(2).mean().sqrt().item()
    _close(stats["1"][:, 1], (ref ** 2).sum((0, 2, 3)), 1e-5)
    if s == 1:
        assert dxe["1"] <= 5e-6 and dxe["1"] <= 1.5 * dxe["0"] + 1e-8, dxe
    # the weight gradient sums 65536+ products per element: slices of <= 4096 pixels in fp32, partial tiles added in order.
    # The bf16 MFMA truncates where it aligns its addends (a bias that grows with instructions x |accumulator|); the split
    # kernel therefore restarts its MFMA accumulators every 8 chunks and sums them in fp32 (round to nearest) -- with that the
    # bar is the forward / data-gradient one: 1.5x the fp32 kernel's error (round 3, without the restart: 3x)
    assert dwe["1"] <= 2e-5 and dwe["1"] <= 1.5 * dwe["0"] + 2e-7 and dwe["r"] <= 1.5 * dwe["0"] + 1e-7, dwe
    assert dbe["1"] <= 2e-5 and dbe["0"] <= 2e-5, dbe


@pytest.mark.parametrize("C", [36, 12, 8])
def test_row_window_form_of_the_space_to_depth_stems(C, monkeypatch):
    """The 4x4 / pad (2, 1) convolution over the space-to-depth stem input (C = 36 / 12 / 8) on the halo kernel's row-window
    form: a filter row of 4 C contiguous floats read as whole 16-channel groups, windows that leave the image row masked in
    the first / last tile of a row.  vs an fp64 convolution, next to the fp32-MFMA kernel and the gather kernel (which
    leaves the last channel group partly empty); BatchNorm partial sums included."""
    N, H, W, Co = 2, 256, 320, 64
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, H, W, generator=g)
    x[:, :, :, :3] *= 11.0                                      # (the masked columns carry weight)
    x[:, :, :, -3:] *= 7.0
    w = torch.randn(Co, C, 4, 4, generator=g) / (C * 16) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=1, padding=2)[:, :, :H, :W]
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    errs, sums = {}, {}
    for name, fl, code in (("halo", ops.CONV_AUTO, 3), ("gather", ops.CONV_X3_IM2COL, 2), ("fp32", ops.CONV_FP32_MFMA, 0)):
        monkeypatch.setattr(ops, "CONV_FLAGS", fl)
        assert ops.lib.pd_conv2d_uses_x3(N * H * W, Co, C, 4, 4, 1, 2, 0, 0, 0, H, W, fl) == code
        y, st = ops.conv2d_fwd(xd, wd, b.cuda(), stride=1, pad=2, out_hw=(H, W), want_stats=True)
        errs[name] = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        sums[name] = st.double().sum(0).cpu()
    assert errs["halo"] <= 5e-6 and errs["halo"] <= 1.5 * errs["fp32"] + 1e-8, errs
    _close(sums["halo"][:, 1], (ref ** 2).sum((0, 2, 3)), 1e-5)
    M = N * H * W
    assert (sums["halo"][:, 0] - ref.sum((0, 2, 3))).abs().max().item() / M <= 2.5e-7 * ref.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("case", [(8, 128, 128, 160, 64), (16, 256, 64, 80, 128), (16, 512, 32, 40, 256), (4, 48, 128, 160, 64),
                                  (2, 96, 256, 320, 32), (8, 64, 128, 160, 32)])
def test_bf16x3_kernel_reflection_padding_bias_elu(case, monkeypatch):
    """The decoder's ConvBlock (ReflectionPad2d(1) + Conv3x3 + bias + ELU) on the bf16-split kernel (256- and 128-row tiles, a
    partly empty channel group) vs an fp64 reference, next to the fp32-MFMA kernel."""
    N, C, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    ref = F.elu(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double()))
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    errs = {}
    for knob in ("1", "0"):
        monkeypatch.setattr(ops, "CONV_FLAGS", ops.CONV_AUTO if knob == "1" else ops.CONV_FP32_MFMA)
        assert bool(ops.lib.pd_conv2d_uses_x3(N * H * W, Co, C, 3, 3, 1, 1, ops.MODE_REFLECT, ops.ACT_ELU, 0, H, W, ops.CONV_FLAGS)) == (knob == "1")
        y = ops.conv2d_fwd(xd, wd, b.cuda(), stride=1, pad=1, mode=ops.MODE_REFLECT, act=ops.ACT_ELU)
        errs[knob] = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert errs["1"] <= 5e-6 and errs["1"] <= 1.5 * errs["0"] + 1e-7, errs
    # ... and the weight / bias gradient of the same layer (mirrored strips in the halo kernel; 32 output channels: its
    # 32-channel workgroups with two partial slices per tile)
    dy = torch.randn(ref.shape, generator=g)
    wv = w.double().requires_grad_(True)
    ref_dw = torch.autograd.grad(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), wv), wv, dy.double())[0]
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dwe = {}
    for knob in ("1", "0"):
        monkeypatch.setattr(ops, "WGRAD_FLAGS", ops.CONV_AUTO if knob == "1" else ops.CONV_FP32_MFMA)
        dw, db = ops.conv2d_wgrad(xd, dyd, w.shape, stride=1, pad=1, mode=ops.MODE_REFLECT, want_bias=True)
        dwe[knob] = (dw.cpu().double() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
        _close(db.cpu(), dy.double().sum((0, 2, 3)).float(), 2e-5)
    assert dwe["1"] <= 2e-5 and dwe["1"] <= 1.5 * dwe["0"] + 2e-7, dwe


@pytest.mark.parametrize("case", [(4, 64, 256, 320, 64, ops.MODE_ZERO), (8, 128, 128, 160, 64, ops.MODE_REFLECT), (32, 64, 128, 48, 128, ops.MODE_ZERO)])
def test_rolling_row_weight_gradient(case, monkeypatch):
    """conv_wgrad_roll_x3_kernel: all three filter rows of a 3x3 weight gradient in one workgroup, the input row groups rolling
    through a ring in LDS (1 x 32 tiles; 2 x 16 tiles for Wo = 48; zero and reflection padding; bias): vs an fp64 reference,
    next to the one-filter-row-per-workgroup kernel and the fp32-MFMA kernel on the same input."""
    N, C, H, W, Co, mode = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    x[:, :, :3] *= 5.0                                       # (first / last rows and columns carry weight: the padding matters)
    x[:, :, :, -2:] *= 3.0
    dy = torch.randn(N, Co, H, W, generator=g)
    wv = torch.zeros(Co, C, 3, 3, dtype=torch.float64, requires_grad=True)
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if mode == ops.MODE_REFLECT else "constant")
    ref_dw = torch.autograd.grad(F.conv2d(xp, wv), wv, dy.double())[0]
    ref_db = dy.double().sum((0, 2, 3))
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    err = {}
    for name, fl, code in (("roll", ops.CONV_AUTO, 3), ("rows", ops.CONV_WGRAD_ROW_WORKGROUPS, 2), ("fp32", ops.CONV_FP32_MFMA, 0)):
        monkeypatch.setattr(ops, "WGRAD_FLAGS", fl)
        assert ops.lib.pd_conv2d_wgrad_uses_x3(N * H * W, Co, C, 3, 3, 1, 1, mode, H, W, H, W, fl) == code
        dw, db = ops.conv2d_wgrad(xd, dyd, (Co, C, 3, 3), stride=1, pad=1, mode=mode, want_bias=True)
        err[name] = (dw.cpu().double() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
        assert (db.cpu().double() - ref_db).abs().max().item() / ref_db.abs().max().item() <= 2e-5
    assert err["roll"] <= 2e-5 and err["roll"] <= 1.5 * err["fp32"] + 2e-7 and err["rows"] <= 1.5 * err["fp32"] + 2e-7, err


def test_padded_grid_data_gradient_with_a_partial_last_tile(monkeypatch):
    """The decoder's deep levels take the data gradient of ReflectionPad2d(1) + Conv3x3 on the PADDED grid (34 x 42 for a 32 x 40
    plane): M = 16 * 34 * 42 = 22848 = 178 tiles of 128 rows + 64 rows.  The bf16-split gather kernel takes it with a partial
    last tile (row-tested stores); vs an fp64 transposed convolution, next to the fp32-MFMA kernel; the rows behind the
    tensor's end stay untouched."""
    N, Cz, Ci, H, W = 16, 128, 256, 32, 40
    g = torch.Generator().manual_seed(5)
    dz = torch.randn(N, Cz, H, W, generator=g)
    w = torch.randn(Cz, Ci, 3, 3, generator=g) / (Cz * 9) ** 0.5
    ref = F.conv_transpose2d(dz.double(), w.double())                     # [N, Ci, H + 2, W + 2]
    dzd = dz.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    err = {}
    for name, fl in (("x3", ops.CONV_AUTO), ("fp32", ops.CONV_FP32_MFMA)):
        monkeypatch.setattr(ops, "CONV_FLAGS", fl)
        ops.PROFILE = []
        try:
            dx = ops.conv2d_dgrad(dzd, wd, (H + 2, W + 2), 1, 0)
            lab = [p[0] for p in ops.PROFILE]
        finally:
            ops.PROFILE = None
        assert lab[0].startswith("conv_igemm_x3_kernel<128,64>" if name == "x3" else "conv_igemm_uni_kernel"), lab
        assert dx.shape == ref.shape
        err[name] = (dx.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err["x3"] <= 5e-6 and err["x3"] <= 1.5 * err["fp32"] + 1e-8, err


# ------------------------------------------------------------------ the bf16-split kernels on inputs that are not `randn`
def _field(mask_shape, n, h, w, k):
    """Boolean [N,1,H,W] mask of the output pixels whose k x k window (pad k//2, stride 1) contains input pixel (n, h, w)."""
    m = torch.zeros(mask_shape[0], 1, mask_shape[2], mask_shape[3], dtype=torch.bool)
    r = k // 2
    m[n, 0, max(h - r, 0):h + r + 1, max(w - r, 0):w + r + 1] = True
    return m


def _three(fn):
    """fn() under flags AUTO (bf16-split where eligible) and PD_CONV_FP32_MFMA: returns (split result, fp32-MFMA result)."""
    with ops.conv_flags(conv=ops.CONV_AUTO, wgrad=ops.CONV_AUTO):
        a = fn()
    with ops.conv_flags(conv=ops.CONV_FP32_MFMA, wgrad=ops.CONV_FP32_MFMA):
        b = fn()
    torch.cuda.synchronize()
    return a.cpu(), b.cpu()


BF16_OVERFLOW = 3.3961775292304e38          # (2 - 2^-8) * 2^127: fp32 values from here on round to bf16 infinity


def test_bf16x3_kernels_nonfinite_and_extreme_inputs():
    """What the bf16-split kernels (forward, data gradient, weight gradient) do with operands the fp32 MFMA handles by IEEE
    rules (the reference's arithmetic: plain fp32 nn.Conv2d, pre_encoders.py:8-34) -- documented at pd_conv2d (polardepth.h):
      * NaN stays NaN: every output that contracts a NaN operand is NaN in both kernel families;
      * +-inf: the fp32 kernel returns +-inf (or NaN where infinities cancel), the split kernel a NON-FINITE value -- hi = inf
        makes mid = x - hi = NaN -- at exactly the same outputs; nothing else is touched;
      * |x| >= (2 - 2^-8) 2^127 ~ 3.396e38 rounds to bf16 infinity: non-finite in the split kernel, finite in fp32 (a caller
        who needs that range passes PD_CONV_FP32_MFMA); |x| just below it is exact (hi + mid + lo = x);
      * everywhere else the two families agree with the fp64 reference as on `randn` inputs (<= 5e-6 of the scale, split
        <= 1.5x fp32)."""
    N, C, H, W, Co, k = 2, 64, 256, 320, 64, 3
    g = torch.Generator().manual_seed(20)
    x = torch.randn(N, C, H, W, generator=g)
    w = (torch.rand(Co, C, k, k, generator=g) + 0.5) / (C * k * k)          # all weights positive: an infinity cannot cancel
    dyv = torch.randn(N, Co, H, W, generator=g)
    special = {"nan": (0, 5, 10, 10, float("nan")), "pinf": (0, 7, 100, 50, float("inf")), "ninf": (1, 3, 200, 300, float("-inf")),
               "big": (1, 9, 30, 30, 3.39e38), "over": (1, 11, 60, 200, 3.4e38)}
    assert 3.39e38 < BF16_OVERFLOW < 3.4e38

    def inject(t):
        t = t.clone()
        for n, c, h, ww, v in special.values():
            t[n, c, h, ww] = v
        return t

    fields = {name: _field((N, 1, H, W), n, h, ww, k) for name, (n, c, h, ww, v) in special.items()}
    plain = ~torch.stack(list(fields.values())).any(0)

    def check_maps(name, split, fp32, ref):
        sf, ff, rf = torch.isfinite(split), torch.isfinite(fp32), torch.isfinite(ref)
        for nm in ("nan", "pinf", "ninf"):
            m = fields[nm].expand_as(ref)
            assert not rf[m].any() and not ff[m].any() and not sf[m].any(), f"{name}: {nm} field must be non-finite in every path"
        assert torch.isnan(split[fields["nan"].expand_as(ref)]).all() and torch.isnan(fp32[fields["nan"].expand_as(ref)]).all()
        assert torch.isinf(fp32[fields["pinf"].expand_as(ref)]).all() and torch.isinf(fp32[fields["ninf"].expand_as(ref)]).all()
        m = fields["over"].expand_as(ref)
        assert ff[m].all() and rf[m].all(), f"{name}: 3.4e38 is finite in fp32"
        assert not sf[m].any(), f"{name}: 3.4e38 rounds to bf16 infinity in the split kernel (documented deviation)"
        m = fields["big"].expand_as(ref)
        sc = ref[m].abs().max().item()
        e_s, e_f = (split[m].double() - ref[m]).abs().max().item() / sc, (fp32[m].double() - ref[m]).abs().max().item() / sc
        assert sf[m].all() and e_s <= 5e-6 and e_s <= 1.5 * e_f + 1e-7, f"{name}: 3.39e38 field {e_s:.2e} vs {e_f:.2e}"
        m = plain.expand_as(ref)
        assert sf[m].all() and ff[m].all(), f"{name}: a special value leaked outside its receptive field"
        sc = ref[m].abs().max().item()
        e_s, e_f = (split[m].double() - ref[m]).abs().max().item() / sc, (fp32[m].double() - ref[m]).abs().max().item() / sc
        assert e_s <= 5e-6 and e_f <= 5e-6 and e_s <= 1.5 * e_f + 1e-8, f"{name}: plain region {e_s:.2e} vs {e_f:.2e}"

    # forward: special values in the activation
    xs = inject(x)
    ref = F.conv2d(xs.double(), w.double(), None, padding=1)
    xd = xs.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    assert ops.lib.pd_conv2d_uses_x3(N * H * W, Co, C, k, k, 1, 1, 0, 0, 0, H, W, ops.CONV_AUTO) == 3          # the halo-tile kernel
    check_maps("forward", *_three(lambda: ops.conv2d_fwd(xd, wd, None, stride=1, pad=1)), ref)
    # data gradient: special values in dY (same channel count in and out: the same fields)
    dys = inject(dyv)
    ref = F.conv_transpose2d(dys.double(), w.double(), None, stride=1, padding=1)
    dyd = dys.cuda().contiguous(memory_format=torch.channels_last)
    check_maps("dgrad", *_three(lambda: ops.conv2d_dgrad(dyd, wd, (H, W), stride=1, pad=1)), ref)
    # weight gradient: special values in X reach dW[:, ci, :, :] of their channel; dY small so that 3.39e38 * dy stays finite
    dsm = (dyv * 1e-3).cuda().contiguous(memory_format=torch.channels_last)
    assert ops.lib.pd_conv2d_wgrad_uses_x3(N * H * W, Co, C, k, k, 1, 1, 0, H, W, H, W, ops.CONV_AUTO) == 2          # the halo-tile kernel
    split, fp32 = _three(lambda: ops.conv2d_wgrad(xd, dsm, w.shape, stride=1, pad=1))
    ref = torch.autograd.grad(F.conv2d(xs.double(), wv := w.double().requires_grad_(True), None, padding=1), wv, (dyv * 1e-3).double())[0]
    ch = {name: c for name, (n, c, h, ww, v) in special.items()}
    for nm in ("nan", "pinf", "ninf"):
        assert not torch.isfinite(split[:, ch[nm]]).any() and not torch.isfinite(fp32[:, ch[nm]]).any() and not torch.isfinite(ref[:, ch[nm]]).any()
    assert torch.isnan(split[:, ch["nan"]]).all() and torch.isnan(fp32[:, ch["nan"]]).all()
    assert torch.isfinite(fp32[:, ch["over"]]).all() and not torch.isfinite(split[:, ch["over"]]).any()
    rest = [c for c in range(C) if c not in (ch["nan"], ch["pinf"], ch["ninf"], ch["over"])]
    assert torch.isfinite(split[:, rest]).all() and torch.isfinite(fp32[:, rest]).all()
    for cs in ([ch["big"]], [c for c in rest if c != ch["big"]]):
        sc = ref[:, cs].abs().max().item()
        e_s = (split[:, cs].double() - ref[:, cs]).abs().max().item() / sc
        e_f = (fp32[:, cs].double() - ref[:, cs]).abs().max().item() / sc
        assert e_s <= 2e-5 and e_s <= 3 * e_f + 2e-7, (cs[:2], e_s, e_f)


@pytest.mark.parametrize("magnitude", [1e-39, 1e-36])
def test_bf16x3_kernels_denormal_range(magnitude):
    """fp32 denormals (1e-39) and normal values whose mid / lo terms fall into the bf16 denormal range (1e-36: x - hi ~ 2e-39).
    A priori bound from the formats alone: whatever an implementation flushes, it treats a number below FLT_MIN = 2^-126 as
    zero at worst, so |error| <= K * max|w| * 2^-126 on top of the fp32 kernel's own error -- for forward, data and weight
    gradient.  (The outputs themselves are ~1e-38: this is a statement about absolute noise, not relative accuracy.)"""
    N, C, H, W, Co, k = 2, 64, 256, 320, 64, 3
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(N, C, H, W, generator=g) * 2 - 1) * magnitude
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    K = C * k * k
    slack = K * w.abs().max().item() * 2.0 ** -126
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    split, fp32 = _three(lambda: ops.conv2d_fwd(xd, wd, None, stride=1, pad=1))
    e_s, e_f = (split.double() - ref).abs().max().item(), (fp32.double() - ref).abs().max().item()
    assert torch.isfinite(split).all() and e_s <= 1.5 * e_f + slack, ("fwd", e_s, e_f, slack)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=1, padding=1)          # (x as dY: same shape)
    split, fp32 = _three(lambda: ops.conv2d_dgrad(xd, wd, (H, W), stride=1, pad=1))
    e_s, e_f = (split.double() - ref).abs().max().item(), (fp32.double() - ref).abs().max().item()
    assert torch.isfinite(split).all() and e_s <= 1.5 * e_f + slack, ("dgrad", e_s, e_f, slack)
    dy = torch.randn(N, Co, H, W, generator=g)
    ref = torch.autograd.grad(F.conv2d(x.double(), wv := w.double().requires_grad_(True), None, padding=1), wv, dy.double())[0]
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    split, fp32 = _three(lambda: ops.conv2d_wgrad(xd, dyd, w.shape, stride=1, pad=1))
    slack_w = N * H * W * dy.abs().max().item() * 2.0 ** -126
    e_s, e_f = (split.double() - ref).abs().max().item(), (fp32.double() - ref).abs().max().item()
    assert torch.isfinite(split).all() and e_s <= 3 * e_f + slack_w, ("wgrad", e_s, e_f, slack_w)


def test_bf16x3_kernel_cancellation_heavy_contraction():
    """K = 1600 (5x5x64) products of magnitude ~1e4 that sum to O(1): the three dropped terms of the split (mid*lo, lo*mid,
    lo*lo <= 2^-23 of a product) are no longer small against the RESULT, only against the addends -- as is every fp32
    rounding of the fp32-MFMA kernel's own accumulation.  Same bar as on `randn`: the split kernel's error against an fp64
    convolution is at most 1.5x the fp32 kernel's (forward and data gradient), and both obey this file's rule for an fp32
    accumulation, |err| <= 1e-6 * sum|a*b| (here sum|a*b| ~ K * 1e4: the addends do not change sign at random)."""
    N, C, H, W, Co, k = 2, 64, 256, 320, 64, 5
    g = torch.Generator().manual_seed(22)
    K = C * k * k
    # x: +-100 with a smooth random magnitude; w: +-100-ish, then each filter is made orthogonal to the all-ones window so
    # that a locally constant input cancels; the image is locally constant up to a small perturbation
    base = (torch.rand(N, C, 1, 1, generator=g) + 0.5) * 100.0 * torch.sign(torch.randn(N, C, 1, 1, generator=g))
    x = base.expand(N, C, H, W) + 1e-2 * torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) * 100.0
    w = w - (w * base[0:1]).sum((1, 2, 3), keepdim=True) / (base[0:1] ** 2).sum() / (k * k) * base[0:1]      # sum_k w * base[0] == 0
    ref = F.conv2d(x.double(), w.double(), None, padding=2)
    inner = ref[0, :, 2:-2, 2:-2]
    addends = 100.0 * 100.0
    assert inner.abs().max().item() < 1e-3 * addends * K ** 0.5, "the contraction does not cancel"
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    assert ops.lib.pd_conv2d_uses_x3(N * H * W, Co, C, k, k, 1, 2, 0, 0, 0, H, W, ops.CONV_AUTO) == 3
    split, fp32 = _three(lambda: ops.conv2d_fwd(xd, wd, None, stride=1, pad=2))
    sc = addends * K
    e_s, e_f = (split.double() - ref).abs().max().item() / sc, (fp32.double() - ref).abs().max().item() / sc
    assert e_s <= 1e-6 and e_f <= 1e-6 and e_s <= 1.5 * e_f + 1e-9, ("fwd", e_s, e_f)
    # image 0, interior: the O(1) results themselves
    e_s0 = (split[0, :, 2:-2, 2:-2].double() - inner).abs().max().item()
    e_f0 = (fp32[0, :, 2:-2, 2:-2].double() - inner).abs().max().item()
    assert e_s0 <= 1.5 * e_f0 + 1e-9 * sc, ("fwd, cancelling region", e_s0, e_f0)
    # data gradient with the same operands (dY := x, a Co == C layer)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=1, padding=2)
    split, fp32 = _three(lambda: ops.conv2d_dgrad(xd, wd, (H, W), stride=1, pad=2))
    e_s, e_f = (split.double() - ref).abs().max().item() / sc, (fp32.double() - ref).abs().max().item() / sc
    assert e_s <= 1e-6 and e_f <= 1e-6 and e_s <= 1.5 * e_f + 1e-9, ("dgrad", e_s, e_f)
