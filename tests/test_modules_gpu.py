"""Façade modules (HIP kernels) vs fixtures produced by the reference's own modules (G4) and vs
the CPU oracle where no reference fixture can exist (ResNet stem: torchvision absent).

Tolerances (fp32): forward 2e-5 of the output scale; gradients 5e-4 of the gradient scale
(MFMA accumulation order differs from the CPU's; BatchNorm statistics are reduced in fp64 here)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
sys.path.insert(0, GOLDEN)
from synth_weights import fill_state_dict  # noqa: E402

pytestmark = pytest.mark.gpu
G4 = np.load(os.path.join(GOLDEN, "g4_nets.npz"))
T = lambda a: torch.from_numpy(np.asarray(a))
FWD_TOL, GRAD_TOL = 2e-5, 5e-4


def _close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).float()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _check_module(name, mod, args, input_grad=True):
    fill_state_dict(mod, 0, prefix=name + ".")
    mod.cuda()
    mod.eval()
    cargs = [a.cuda() if a is not None else None for a in args]
    with torch.no_grad():
        y = mod(*cargs)
    ys = y if isinstance(y, (list, tuple)) else [y]
    for i, t in enumerate(ys):
        _close(t, T(G4[f"{name}.eval.{i}"]), FWD_TOL, f"{name}.eval.{i}")
    mod.train()
    args_g = [a.clone().requires_grad_(input_grad) if a is not None else None for a in cargs]
    y = mod(*args_g)
    ys = y if isinstance(y, (list, tuple)) else [y]
    obj = 0
    for i, t in enumerate(ys):
        _close(t, T(G4[f"{name}.train.{i}"]), FWD_TOL, f"{name}.train.{i}")
        w = torch.randn(t.shape, generator=torch.Generator().manual_seed(100 + i)).cuda()
        obj = obj + (t * w).sum()
    obj.backward()
    n = 0
    for k, p in mod.named_parameters():
        if f"{name}.grad.{k}" in G4:
            ref = T(G4[f"{name}.grad.{k}"])
            if k.endswith("conv.bias"):       # conv bias feeds BatchNorm: exact zero here, ~1e-7 noise in torch
                assert p.grad is None or p.grad.abs().max().item() <= 1e-4 * (1 + ref.abs().max().item())
                continue
            _close(p.grad, ref, GRAD_TOL, f"{name}.grad.{k}"); n += 1
        elif f"{name}.gradsample.{k}" in G4:
            g = p.grad.cpu().contiguous().flatten()
            _close(g[::max(1, g.numel() // 4096)], T(G4[f"{name}.gradsample.{k}"]), GRAD_TOL, f"{name}.gradsample.{k}")
            n += 1
    assert n > 10
    for i, a in enumerate(args_g):
        if a is not None and a.grad is not None and f"{name}.ingrad.{i}" in G4:
            _close(a.grad, T(G4[f"{name}.ingrad.{i}"]), GRAD_TOL, f"{name}.ingrad.{i}")
    for k, v in mod.state_dict().items():
        if f"{name}.buf.{k}" in G4:
            _close(v, T(G4[f"{name}.buf.{k}"]), 1e-5, f"{name}.buf.{k}")


def test_xolp_and_normals_encoders_match_reference():
    from manydepth import networks
    xolp = T(G4["xolp"])
    _check_module("xolp_encoder", networks.ShallowEncoder('XOLP', 2, 0.0), [xolp], input_grad=False)
    _check_module("normals_encoder", networks.ShallowNormalsEncoder(9, 0.0), [xolp], input_grad=False)


@pytest.mark.parametrize("name,inc_n,inc_x", [("joint3", True, True), ("joint_x", False, True),
                                              ("joint_n", True, False), ("joint_rgb", False, False)])
def test_joint_encoder_matches_reference(name, inc_n, inc_x):
    from manydepth import networks
    rgbf, xf, nf = T(G4["joint.rgbf"]), T(G4["joint.xf"]), T(G4["joint.nf"])
    _check_module(name, networks.JointEncoder(0.0, inc_n, inc_x), [rgbf, xf if inc_x else None, nf if inc_n else None])


def test_depth_decoder_matches_reference():
    from manydepth import networks
    feats = [T(G4[f"dec.feat.{i}"]).cuda().requires_grad_(True) for i in range(5)]
    dd = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4))
    fill_state_dict(dd, 0, prefix="mono_depth.")
    dd.cuda()
    res = dd(feats)
    obj = 0
    for s in range(4):
        _close(res[("disp", s)], T(G4[f"dec.disp.{s}"]), FWD_TOL, f"disp{s}")
        w = torch.randn(res[("disp", s)].shape, generator=torch.Generator().manual_seed(200 + s)).cuda()
        obj = obj + (res[("disp", s)] * w).sum()
    obj.backward()
    for i, f in enumerate(feats):
        _close(f.grad, T(G4[f"dec.featgrad.{i}"]), GRAD_TOL, f"featgrad{i}")
    n = 0
    for k, p in dd.named_parameters():
        if f"dec.grad.{k}" in G4:
            _close(p.grad, T(G4[f"dec.grad.{k}"]), GRAD_TOL, k); n += 1
        elif f"dec.gradsample.{k}" in G4:
            g = p.grad.cpu().contiguous().flatten()
            _close(g[::max(1, g.numel() // 4096)], T(G4[f"dec.gradsample.{k}"]), GRAD_TOL, k); n += 1
    assert n == 28


@pytest.mark.parametrize("scales", [[0, 1, 2, 3], [0, 2]])
def test_decoder_activation_gradient_handover_is_bit_exact(scales, monkeypatch):
    """ActGrad mailboxes (ELU derivative applied by pd_up_bwd_elu / pd_disphead_bwd, the head summing the two
    gradients of upconv(i,1)'s output) against the plain route (autograd add + pd_act_bwd): identical bits, for all
    heads present and for levels without a head."""
    from manydepth import networks
    from polardepth import functional as PF

    def run(fuse):
        monkeypatch.setattr(PF, "USE_ACT_FUSION", fuse)
        feats = [T(G4[f"dec.feat.{i}"]).cuda().requires_grad_(True) for i in range(5)]
        torch.manual_seed(3)
        dd = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), scales).cuda()
        res = dd(feats)
        obj = 0
        for s in scales:
            w = torch.randn(res[("disp", s)].shape, generator=torch.Generator().manual_seed(200 + s)).cuda()
            obj = obj + (res[("disp", s)] * w).sum()
        obj.backward()
        PF.sync_wgrad_stream()
        torch.cuda.synchronize()
        return [f.grad.clone() for f in feats] + [p.grad.clone() for p in dd.parameters()]

    a, b = run(True), run(False)
    assert len(a) == len(b) and len(a) > 20
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("used", [[0], [2], [1, 3]])
def test_backward_through_a_subset_of_the_decoder_outputs(used, monkeypatch):
    """A decoder built with all four heads, a loss over only some of its outputs (``outputs[("disp", 0)].sum().backward()``):
    the unused heads still run (JoinHeadsFn hands them zero gradients) and collect the data gradient the next level's first
    convolution deposited for them -- gradients equal the plain autograd route (PD_ACT_FUSION=0) instead of silently
    vanishing upstream of the first unused head."""
    from manydepth import networks
    from polardepth import functional as PF

    def run(fuse):
        monkeypatch.setattr(PF, "USE_ACT_FUSION", fuse)
        feats = [T(G4[f"dec.feat.{i}"]).cuda().requires_grad_(True) for i in range(5)]
        torch.manual_seed(3)
        dd = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4)).cuda()
        res = dd(feats)
        obj = 0
        for s in used:
            w = torch.randn(res[("disp", s)].shape, generator=torch.Generator().manual_seed(300 + s)).cuda()
            obj = obj + (res[("disp", s)] * w).sum()
        obj.backward()
        PF.sync_wgrad_stream()
        torch.cuda.synchronize()
        # (levels below the shallowest used scale are outside the graph on the plain route: None there, zeros here)
        return [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in feats + list(dd.parameters())]

    a, b = run(True), run(False)
    assert len(a) == len(b) and len(a) > 20
    assert a[4].abs().max().item() > 0          # the deepest feature map does receive a gradient
    for i, (x, y) in enumerate(zip(a, b)):
        scale = y.abs().max().item()
        assert (x - y).abs().max().item() <= 1e-6 * scale + 1e-12, f"tensor {i}: {(x - y).abs().max().item():.3e} vs {scale:.3e}"


def test_uncollected_activation_gradient_deposit_fails_loudly(monkeypatch):
    """A hand-built graph in which the head of a tensor never runs: the deposit is detected at the end of the backward
    pass and raises, instead of leaving everything upstream without a gradient."""
    from polardepth import functional as PF
    from polardepth import ops
    torch.manual_seed(0)
    conv_a = torch.nn.Conv2d(32, 32, 3).cuda(); conv_b = torch.nn.Conv2d(32, 32, 3).cuda()
    for c in (conv_a, conv_b):
        c.weight.data = c.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(1, 32, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    mail = PF.ActGrad(expect_deposit=True)
    y = PF.reflect_conv_act(x, conv_a, ops.ACT_ELU, act_mail=mail)
    z = PF.reflect_conv_act(y, conv_b, ops.ACT_ELU, dx_mail=mail)      # deposits dL/dy for a head that does not exist
    with pytest.raises(RuntimeError, match="ActGrad"):
        z.sum().backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()
    assert mail.grad is None                    # the mailbox does not stay pinned


def test_resnet_stem_matches_oracle():
    """No reference fixture possible (torchvision absent): HIP façade vs the CPU restatement."""
    from manydepth import networks
    from oracle import nets as onets
    ref = onets.ShallowResnetEncoder(18, False)
    fill_state_dict(ref, 0, prefix="rgb_encoder.")
    mod = networks.ShallowResnetEncoder(18, False)
    mod.load_state_dict(ref.state_dict())
    mod.cuda()
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    img = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(4))
    for mode in ("eval", "train"):
        getattr(ref, mode)(); getattr(mod, mode)()
        ref.zero_grad(); mod.zero_grad()
        fr = ref(img)
        fm = mod(img.cuda())
        obj_r = obj_m = 0
        for i in range(3):
            _close(fm[i], fr[i].detach(), FWD_TOL, f"f{i} {mode}")
            w = torch.randn(fr[i].shape, generator=torch.Generator().manual_seed(300 + i))
            obj_r = obj_r + (fr[i] * w).sum(); obj_m = obj_m + (fm[i] * w.cuda()).sum()
        if mode == "train":
            obj_r.backward(); obj_m.backward()
            for (k, pr), (_, pm) in zip(ref.named_parameters(), mod.named_parameters()):
                if pr.grad is not None:
                    _close(pm.grad, pr.grad, GRAD_TOL, k)
            _close(mod.encoder.bn1.running_mean, ref.encoder.bn1.running_mean, 1e-5)
            _close(mod.encoder.layer2[0].downsample[1].running_var, ref.encoder.layer2[0].downsample[1].running_var, 1e-5)


def test_attention_variant_matches_oracle():
    """BASELINE config 5 (attention at the joint-encoder merge): HIP path vs the CPU restatement (the reference
    branch is absent: parity unpinned).  Also checks the raw attention op against torch softmax attention."""
    from manydepth.networks.pre_encoders import JointEncoder, JointAttention
    from polardepth import functional as PF
    from oracle import nets as onets
    g = torch.Generator().manual_seed(12)
    q, k, v = (torch.randn(2, 128, 8, 12, generator=g) for _ in range(3))
    qc, kc, vc = (t.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in (q, k, v))
    assert PF.USE_FLASH_ATTENTION
    o = PF.self_attention(qc, kc, vc)
    assert o.grad_fn.__class__.__name__.startswith("FlashAttentionFn")      # 8 x 12 = 96 tokens: fused kernels
    # the materialised-score path (any C, any T) must agree
    q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (qc, kc, vc))
    o2 = PF.SelfAttentionFn.apply(q2, k2, v2)
    _close(o2, o.detach().cpu(), FWD_TOL, "materialised vs fused attention")
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    tok = lambda t: t.flatten(2).transpose(1, 2)
    ref = (torch.softmax(tok(qr) @ tok(kr).transpose(1, 2) / 128 ** 0.5, -1) @ tok(vr)).transpose(1, 2).reshape(2, 128, 8, 12)
    _close(o, ref.detach(), FWD_TOL, "attention fwd")
    w = torch.randn(ref.shape, generator=g)
    (o * w.cuda()).sum().backward(); (ref * w).sum().backward()
    for name, a, b in (("dq", qc, qr), ("dk", kc, kr), ("dv", vc, vr)):
        _close(a.grad, b.grad, GRAD_TOL, name)
    # whole joint encoder with the attention block
    ref_m = onets.JointEncoder(0.0, True, True, attention=True)
    fill_state_dict(ref_m, 0, prefix="joint_attn.")
    mod = JointEncoder(0.0, True, True, attention=True)
    mod.load_state_dict(ref_m.state_dict())
    mod.cuda().train(); ref_m.train()
    rgbf, xf, nf = T(G4["joint.rgbf"]), T(G4["joint.xf"]), T(G4["joint.nf"])
    ins = [t.clone().requires_grad_(True) for t in (rgbf, xf, nf)]
    cins = [t.cuda().requires_grad_(True) for t in (rgbf, xf, nf)]
    yr = ref_m(*ins); ym = mod(*cins)
    obj_r = obj_m = 0
    for i in range(2):
        _close(ym[i], yr[i].detach(), FWD_TOL, f"joint+attn out{i}")
        w = torch.randn(yr[i].shape, generator=torch.Generator().manual_seed(400 + i))
        obj_r = obj_r + (yr[i] * w).sum(); obj_m = obj_m + (ym[i] * w.cuda()).sum()
    obj_r.backward(); obj_m.backward()
    for (kname, pr), (_, pm) in zip(ref_m.named_parameters(), mod.named_parameters()):
        if kname.endswith("conv.bias") or kname == "attn.k.bias":
            continue                  # exactly-zero gradients (BN / softmax shift invariance): rounding noise only
        _close(pm.grad, pr.grad, GRAD_TOL, kname)
    for a, b in zip(cins, ins):
        _close(a.grad, b.grad, GRAD_TOL, "input grad")


@pytest.mark.parametrize("shape", [(2, 8, 12), (1, 64, 80)])
def test_bf16_attention_matches_fp32_attention_within_bf16_tolerance(shape):
    """configs[4] as specified: the fused attention on v_mfma_f32_32x32x16_bf16 (csrc/attention_bf16.hip).  Operands are
    rounded to bf16 (8-bit mantissa, 2^-9 relative), softmax statistics and accumulators stay fp32.  Tolerance, of
    the tensor's own scale: forward 1e-2, gradients 2e-2 (the fp32 kernels: 2e-5 / 5e-4); the log-sum-exp statistic
    (fp32 arithmetic on scores of bf16-rounded operands: |score| * 2^-8) 2e-2 absolute.  Reference: fp64 softmax
    attention in torch."""
    from polardepth import functional as PF
    from polardepth._lib import lib, check, ptr, stream_ptr
    N, Hh, Ww = shape
    g = torch.Generator().manual_seed(21)
    q, k, v = (torch.randn(N, 128, Hh, Ww, generator=g) for _ in range(3))
    if Hh * Ww > 1000:          # make the softmax peaky on the big case: a few keys carry most of the mass
        q = q * 2.0
    qc, kc, vc = (t.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in (q, k, v))
    old = PF.USE_BF16_ATTENTION
    PF.USE_BF16_ATTENTION = True
    try:
        o = PF.self_attention(qc, kc, vc)
        assert o.grad_fn.__class__.__name__.startswith("FlashAttentionFn")
        w = torch.randn(N, 128, Hh, Ww, generator=g)
        (o * w.cuda()).sum().backward()
    finally:
        PF.USE_BF16_ATTENTION = old
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    tok = lambda t: t.flatten(2).transpose(1, 2)
    scores = tok(qr) @ tok(kr).transpose(1, 2) / 128 ** 0.5
    ref = (torch.softmax(scores, -1) @ tok(vr)).transpose(1, 2).reshape(N, 128, Hh, Ww)
    (ref * w.double()).sum().backward()
    _close(o, ref.detach().float(), 1e-2, "bf16 attention fwd")
    for name, a, b in (("dq", qc, qr), ("dk", kc, kr), ("dv", vc, vr)):
        _close(a.grad, b.grad.float(), 2e-2, "bf16 attention " + name)
    # lse from the kernel directly
    T_ = Hh * Ww
    o2 = torch.empty_like(qc.detach())
    lse = torch.empty((N, T_), device="cuda")
    ws = torch.empty(lib.pd_attn_bf16_workspace(N, T_, 128, 0), dtype=torch.uint8, device="cuda")
    check(lib.pd_attn_bf16_fwd(ptr(qc.detach()), ptr(kc.detach()), ptr(vc.detach()), ptr(o2), ptr(lse), ptr(ws), ws.numel(),
                               N, T_, 128, 1.0 / 128 ** 0.5, stream_ptr()), "pd_attn_bf16_fwd")
    assert (lse.cpu().double() - torch.logsumexp(scores.detach(), -1)).abs().max().item() < 2e-2
    assert torch.equal(o2, o.detach())


def test_batched_weight_transposes_follow_the_weights():
    """ParamStore serves the data-gradient operands of all its conv weights from one launch per backward pass: the
    copies are redone at the first request after any forward convolution (or optimizer step), so they follow every
    way the weights can change between two passes -- torch in-place ops, writes through .data, the raw-pointer Adam."""
    from polardepth import engine, ops
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.Conv2d(16, 4, 5), torch.nn.Linear(4, 2)).cuda()
    store = engine.ParamStore({"net": net})
    opt = engine.FusedAdam(store, lr=1e-2)
    convs = [m for m in net if isinstance(m, torch.nn.Conv2d)]

    def check_all():
        for m in convs:
            wt = ops.weight_transposed(m.weight.data)
            assert wt.data_ptr() != m.weight.data_ptr()
            assert wt.shape == (m.in_channels, m.out_channels) + m.kernel_size
            assert torch.equal(wt, m.weight.data.permute(1, 0, 2, 3)), "stale or wrong transposed weight"
            assert torch.equal(ops.weight_transposed(m.weight.data.clone()), wt)   # the per-layer kernel (tensor outside any store)

    x = torch.randn(2, 8, 12, 12, device="cuda")
    check_all()
    with torch.no_grad():
        convs[0].weight.mul_(2.0)
        convs[1].weight.data.add_(1.0)
    ops.conv2d_fwd(x, convs[0].weight.data)            # the forward pass every backward pass follows
    check_all()
    launches_before = store._wt_epoch
    check_all()                                        # same pass: served from the same batch
    assert store._wt_epoch == launches_before
    store.grad.normal_()
    store.grad_is_zero = False
    before = convs[0].weight.data.clone()
    opt.step()
    assert not torch.equal(before, convs[0].weight.data)
    check_all()


def test_normals_decoder_variant_matches_oracle():
    """`arch1++_separate_normals_dec` (README.md:54; defined by this build, see manydepth/networks/normals_decoder.py): the
    façade NormalsDecoder + the predicted-normals loss kernels vs the same definition on PyTorch-CPU (oracle/nets.py,
    oracle/losses.py), forward, loss value and every gradient."""
    from manydepth import networks
    from oracle import nets as onets, losses as ol
    from polardepth import functional as PF
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 64, 96
    feat = torch.randn(N, 64, H // 8, W // 8, generator=g)
    gt = 0.3 + 1.5 * torch.rand(N, 1, H, W, generator=g)
    gt = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(gt, (2, 2, 2, 2), mode="replicate"), 5, 1)     # smooth surface
    gt[:, :, :6, :9] = 0.0                                  # invalid region (outside the depth range)
    gt[:, :, 40:44, 50:60] = 2.5
    K = torch.eye(4)[None].repeat(N, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 0.65 * W; K[:, 0, 2] = W / 2; K[:, 1, 2] = H / 2
    ref = onets.NormalsDecoder(64)
    fill_state_dict(ref, 0, prefix="normals_decoder.")
    mod = networks.NormalsDecoder(64)
    fill_state_dict(mod, 0, prefix="normals_decoder.")
    assert list(ref.state_dict()) == list(mod.state_dict())
    mod.cuda()
    fr = feat.clone().requires_grad_(True)
    yr = ref(fr)
    lr = ol.normals_pred_loss(yr, gt, K, 0.1, 2.0)
    lr.backward()
    fg = feat.cuda().requires_grad_(True)
    yg = mod(fg)
    lg = PF.normals_pred_loss(yg, gt.cuda(), K.cuda(), 0.1, 2.0)
    (3.0 * lg).backward()
    PF.sync_wgrad_stream()
    _close(yg, yr, FWD_TOL, "normals_pred")
    assert abs(lg.item() - lr.item()) <= 1e-5 * abs(lr.item()), (lg.item(), lr.item())
    _close(fg.grad / 3.0, fr.grad, GRAD_TOL, "feature gradient")
    pr = dict(ref.named_parameters())
    for k, p in mod.named_parameters():
        _close(p.grad / 3.0, pr[k].grad, GRAD_TOL, k)
    # the loss alone, on a prediction with a zero vector and a huge one: value and gradient
    pred = torch.randn(N, 3, H, W, generator=g)
    pred[0, :, 20, 30] = 0.0
    pred[1, :, 10, 10] *= 1e4
    pc = pred.clone().requires_grad_(True)
    ol.normals_pred_loss(pc, gt, K, 0.1, 2.0).backward()
    pgpu = pred.cuda().requires_grad_(True)
    l2 = PF.normals_pred_loss(pgpu, gt.cuda(), K.cuda(), 0.1, 2.0)
    l2.backward()
    keep = torch.ones(N, 1, H, W, dtype=torch.bool); keep[0, :, 20, 30] = False     # (the clamp's gradient at exactly zero differs)
    _close((pgpu.grad.cpu() * keep), pc.grad * keep, 1e-5, "d loss / d pred")


@pytest.mark.parametrize("shape,kind", [((2, 8, 16), "plain"), ((1, 16, 16), "late_peak"), ((2, 16, 40), "peaky"),
                                        ((1, 8, 24), "plain")])
def test_bf16_attention_pipelined_kernels(shape, kind):
    """The software-pipelined bf16 attention kernels (T % 128 == 0: forward / dK,dV; T % 64 == 0: dQ; (1, 8, 24) has T = 192 and
    mixes the pipelined dQ with the plain forward / dK,dV kernels) across token counts and batch sizes, including the cold
    path of the forward pass's lazy maximum: "late_peak" plants keys in the LAST blocks whose scores exceed every earlier one
    by far more than 2^8, so the running maximum has to move, with O, l and the score tile in flight rescaled.  Reference:
    fp64 softmax attention in torch; tolerances as in the test above."""
    from polardepth import functional as PF
    N, Hh, Ww = shape
    T = Hh * Ww
    g = torch.Generator().manual_seed(33)
    q, k, v = (torch.randn(N, 128, Hh, Ww, generator=g) for _ in range(3))
    if kind == "peaky":
        q = q * 2.5
    if kind == "late_peak":
        kt = k.flatten(2)                       # [N, C, T]
        qt = q.flatten(2)
        for j, t in enumerate((T - 70, T - 3)):      # two late keys aligned with the mean query direction / one query
            kt[:, :, t] = 6.0 * qt[:, :, 5 + j] / qt[:, :, 5 + j].norm(dim=1, keepdim=True) * (1 + j)
        k = kt.view_as(k)
    qc, kc, vc = (t.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in (q, k, v))
    old = PF.USE_BF16_ATTENTION
    PF.USE_BF16_ATTENTION = True
    try:
        o = PF.self_attention(qc, kc, vc)
        w = torch.randn(N, 128, Hh, Ww, generator=g)
        (o * w.cuda()).sum().backward()
    finally:
        PF.USE_BF16_ATTENTION = old
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    tok = lambda t: t.flatten(2).transpose(1, 2)
    scores = tok(qr) @ tok(kr).transpose(1, 2) / 128 ** 0.5
    if kind == "late_peak":      # the planted keys do dominate some rows by more than 2^8 in log2 units
        s2 = scores.detach() * 1.4426950408889634
        assert (s2[:, :, T - 70:].max(-1).values - s2[:, :, :T - 70].max(-1).values).max().item() > 8.0
    ref = (torch.softmax(scores, -1) @ tok(vr)).transpose(1, 2).reshape(N, 128, Hh, Ww)
    (ref * w.double()).sum().backward()
    assert bool(torch.isfinite(o).all())
    _close(o, ref.detach().float(), 1e-2, "bf16 attention fwd")
    for name, a, b in (("dq", qc, qr), ("dk", kc, kr), ("dv", vc, vr)):
        _close(a.grad, b.grad.float(), 2e-2, "bf16 attention " + name)


def test_dpt_fusion_block_matches_the_reference_fixture(golden_dir):
    """manydepth.dpt.blocks (reference manydepth/dpt/blocks.py:138-172, 255-383; dpt/models.py:15-23) against
    g8_dpt_fusion.npz -- outputs of the reference's own FeatureFusionBlock_custom / ResidualConvUnit_custom / Interpolate with
    the seeded weights of tests/golden/synth_weights.py: outputs, input gradients and parameter gradients of the two-input
    and one-input forms (fp32: 2e-5 of the scale forward, 1e-4 backward)."""
    import os
    import sys
    import numpy as np
    import torch.nn as nn
    sys.path.insert(0, golden_dir)
    from synth_weights import fill_state_dict
    from manydepth.dpt import blocks
    from polardepth import functional as PF
    g = np.load(os.path.join(golden_dir, "g8_dpt_fusion.npz"))
    T = lambda k: torch.from_numpy(g[k])

    def close(got, want, tol, what):
        sc = want.abs().max().item() + 1e-12
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= tol * sc, f"{what}: {err:.3e} vs scale {sc:.3e}"

    for tag, n_in in (("two", 2), ("one", 1)):
        blk = blocks.FeatureFusionBlock_custom(64, nn.ReLU(False), deconv=False, bn=False, expand=False, align_corners=True)
        want_keys = [k[len(tag) + 6:] for k in g.files if k.startswith(tag + ".grad.")]
        if tag == "two":
            assert sorted(dict(blk.named_parameters())) == sorted(want_keys)   # the reference's state_dict keys
        fill_state_dict(blk, 0, prefix="fusion.")
        blk.cuda()
        xs = [T(f"{tag}.x{i}").cuda().requires_grad_(True) for i in range(n_in)]
        y = blk(*xs)
        close(y, T(f"{tag}.out"), 2e-5, f"{tag}: output")
        y.backward(T(f"{tag}.gout").cuda())
        PF.sync_wgrad_stream()
        torch.cuda.synchronize()
        for i, x in enumerate(xs):
            close(x.grad, T(f"{tag}.gx{i}"), 1e-4, f"{tag}: d x{i}")
        for k, p in blk.named_parameters():
            if tag == "one" and k.startswith("resConfUnit1."):
                assert p.grad is None or p.grad.abs().max().item() == 0     # unused with one input, as in the reference
                continue
            close(p.grad, T(f"{tag}.grad.{k}"), 1e-4, f"{tag}: grad {k}")
    rcu = blocks.ResidualConvUnit_custom(64, nn.ReLU(False), False)
    fill_state_dict(rcu, 0, prefix="fusion.resConfUnit2.")
    rcu.cuda()
    x = T("rcu.x").cuda().requires_grad_(True)
    y = rcu(x)
    close(y, T("rcu.out"), 2e-5, "rcu: output")
    y.backward(T("rcu.gout").cuda())
    close(x.grad, T("rcu.gx"), 1e-4, "rcu: dx")
    ip = blocks.Interpolate(scale_factor=2, mode="bilinear", align_corners=True)
    x = T("interp.x").cuda().requires_grad_(True)
    y = ip(x)
    close(y, T("interp.out"), 2e-6, "interpolate: output")
    y.backward(T("interp.gout").cuda())
    close(x.grad, T("interp.gx"), 2e-6, "interpolate: dx")
    with pytest.raises(NotImplementedError):
        blocks.FeatureFusionBlock_custom(64, nn.ReLU(False), bn=True)


def test_depth_decoder_with_uncertainty_heads_matches_reference(golden_dir):
    """DepthDecoder(uncertainty=True) (depth_decoder.py:46-50,71-73: two Conv5x5 + sigmoid heads per scale, ModuleList entries
    14..21) against g9_decoder_uncertainty.npz -- outputs of the reference's own class on the feature maps of G4: the twelve
    output maps, feature gradients and parameter gradients; state_dict keys in the reference's order."""
    import os
    from manydepth import networks
    from polardepth import functional as PF
    G9 = np.load(os.path.join(golden_dir, "g9_decoder_uncertainty.npz"))
    feats = [T(G4[f"dec.feat.{i}"]).cuda().requires_grad_(True) for i in range(5)]
    dd = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4), uncertainty=True)
    assert list(dd.state_dict().keys()) == [str(k) for k in G9["keys"]]
    fill_state_dict(dd, 0, prefix="mono_depth.")
    dd.cuda()
    res = dd(feats)
    assert len(res) == 12
    obj = 0
    for j, kind in enumerate(("disp", "uncertainty", "uncertainty_color")):
        for s in range(4):
            t = res[(kind, s)]
            _close(t, T(G9[f"{kind}.{s}"]), FWD_TOL, f"{kind}{s}")
            obj = obj + (t * torch.randn(t.shape, generator=torch.Generator().manual_seed(300 + 10 * j + s)).cuda()).sum()
    obj.backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()
    for i, f in enumerate(feats):
        _close(f.grad, T(G9[f"featgrad.{i}"]), GRAD_TOL, f"featgrad{i}")
    n = 0
    for k, p in dd.named_parameters():
        if f"grad.{k}" in G9:
            _close(p.grad, T(G9[f"grad.{k}"]), GRAD_TOL, k); n += 1
        elif f"gradsample.{k}" in G9:
            g = p.grad.cpu().contiguous().flatten()
            _close(g[::max(1, g.numel() // 4096)], T(G9[f"gradsample.{k}"]), GRAD_TOL, k); n += 1
    assert n == 28 + 16
