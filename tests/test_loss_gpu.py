"""K5 multi-scale loss kernels vs the fixture produced by the reference's Trainer.compute_losses (G5)
and vs the CPU oracle on a second geometry.  Tolerance: 1e-4 relative (north_star), usually ~1e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from polardepth import functional as PF

pytestmark = pytest.mark.gpu
G5 = np.load(os.path.join(GOLDEN, "g5_loss.npz"))
T = lambda a: torch.from_numpy(np.asarray(a))


def _close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).float()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("tag,lam", [("lam0", 0.0), ("lam035", 0.35)])
def test_loss_matches_reference_trainer(tag, lam):
    cfg = PF.LossCfg([0, 1, 2, 3], 0.1, 2.0, lam, 1e-3, 64, 96)
    gt, K = T(G5["in.depth"]).cuda(), T(G5["in.K_0"]).cuda()
    disps = [T(G5[f"disp.{s}"]).cuda().requires_grad_(True) for s in range(4)]
    colors = [T(G5[f"in.color_0_{s}"]).cuda() for s in range(4)]
    vals, depths = PF.multiscale_loss(cfg, gt, K, disps, colors)
    vals[0].backward()
    v = vals.detach().cpu()
    _close(v[0], T(G5[f"{tag}.loss"]), 1e-5, "loss")
    for s in range(4):
        _close(v[1 + 3 * s], T(G5[f"{tag}.loss/{s}"]), 1e-5, f"loss/{s}")
        _close(v[2 + 3 * s], T(G5[f"{tag}.supervised_depth_loss/{s}"]), 1e-5, f"sup/{s}")
        _close(depths[s], T(G5[f"{tag}.depth.{s}"]), 1e-6, f"depth{s}")
        _close(disps[s].grad, T(G5[f"{tag}.ddisp.{s}"]), 1e-4, f"ddisp{s}")


@pytest.mark.parametrize("switch", ["USE_GT_NORMAL_CACHE", "USE_EDGE_WEIGHT_CACHE"])
def test_input_only_caches_are_bit_exact(switch, monkeypatch):
    """pd_gt_normals once per step + cached reads in the eight consumers == recomputing the ground-truth normals in
    every kernel; smoothness edge weights handed from the forward to the backward kernel == recomputing them:
    identical loss values and disparity gradients."""
    cfg = PF.LossCfg([0, 1, 2, 3], 0.1, 2.0, 0.35, 1e-3, 64, 96)
    gt, K = T(G5["in.depth"]).cuda(), T(G5["in.K_0"]).cuda()
    colors = [T(G5[f"in.color_0_{s}"]).cuda() for s in range(4)]
    out = []
    for cache in (True, False):
        monkeypatch.setattr(PF, switch, cache)
        disps = [T(G5[f"disp.{s}"]).cuda().requires_grad_(True) for s in range(4)]
        vals, _ = PF.multiscale_loss(cfg, gt, K, disps, colors)
        vals[0].backward()
        out.append([vals.detach().clone()] + [d.grad.clone() for d in disps])
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("scales,size", [([0, 1, 2, 3], (2, 64, 96)), ([0, 2], (3, 32, 80)), ([1, 3], (1, 128, 64)), ([0], (2, 16, 48))])
def test_all_scales_per_launch_equals_the_per_scale_launches(scales, size, monkeypatch):
    """pd_multiscale_loss_fwd / _bwd (every scale of the loss in one launch per kernel, ground truth read once for all
    scales in the supervised forward pass) against the per-scale launches they replace: identical loss values, depth maps
    and disparity gradients, for all four scales, a subset, and a single scale."""
    N, H, W = size
    g = torch.Generator().manual_seed(sum(size) + len(scales))
    gt = (0.05 + 2.2 * torch.rand(N, 1, H, W, generator=g)).cuda()          # some pixels out of the depth range
    K = torch.eye(4)[None].repeat(N, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.65 * W, 0.65 * W, 0.5 * W, 0.5 * H
    K = K.cuda()
    cfg = PF.LossCfg(scales, 0.1, 2.0, 0.35, 1e-3, H, W)
    colors = [torch.rand(N, 3, H >> s, W >> s, generator=g).cuda() for s in scales]
    out = []
    for multi in (True, False):
        monkeypatch.setattr(PF, "USE_MULTISCALE_LAUNCH", multi)
        disps = [torch.rand(N, 1, H >> s, W >> s, generator=torch.Generator().manual_seed(7 + s)).cuda().requires_grad_(True)
                 for s in scales]
        vals, depths = PF.multiscale_loss(cfg, gt, K, disps, colors)
        (vals[0] * 1.7 + vals[1:].sum()).backward()
        out.append([vals.detach().clone()] + [d.clone() for d in depths] + [d.grad.clone() for d in disps])
    assert len(out[0]) == 1 + 2 * len(scales)
    for i, (a, b) in enumerate(zip(*out)):
        assert torch.isfinite(a).all() and torch.equal(a, b), f"output {i} differs"


@pytest.mark.parametrize("geom", [(2, 64, 96), (1, 21, 70), (3, 8, 130)])
def test_fused_supervised_backward_equals_the_two_pass_form(geom, monkeypatch):
    """pd_sup_loss_bwd: passes A and B through an LDS halo tile == pass A to memory + pass B (same bits), on grids that
    are and are not multiples of the 8 x 64 tile."""
    N, H, W = geom
    g = torch.Generator().manual_seed(H + W)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    gt = (1.0 + 0.5 * torch.sin(xx / 9.0) * torch.cos(yy / 7.0))[None, None].repeat(N, 1, 1, 1)
    gt = (gt + 0.02 * torch.rand(N, 1, H, W, generator=g)).cuda()
    gt[:, :, :2, :3] = 5.0                                      # out-of-range pixels (masked)
    K = torch.eye(4)[None].repeat(N, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.9 * W, 1.1 * H, 0.5 * W, 0.5 * H
    K = K.cuda()
    cfg = PF.LossCfg([0], 0.1, 2.0, 0.35, 1e-3, H, W)
    color = torch.rand(N, 3, H, W, generator=g).cuda()
    out = []
    for two_pass in ("0", "1"):
        monkeypatch.setattr(PF, "SUP_BWD_TWO_PASS", two_pass == "1")       # pd_sup_loss_bwd(two_pass_form=...)
        disp = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
        vals, _ = PF.multiscale_loss(cfg, gt, K, [disp], [color])
        vals[0].backward()
        out.append(disp.grad.clone())
    assert out[0].abs().max() > 0
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("f", [1, 2, 4, 8])
def test_upsample_gradient_gather_specialisations_are_bit_exact(f, monkeypatch):
    """pd_up_gather_bwd: the unrolled power-of-two-zoom kernels against the generic footprint kernel (same bits), and
    both against autograd through F.interpolate on the CPU."""
    import torch.nn.functional as F
    from polardepth._lib import lib, check, ptr
    N, hs, ws = 2, 6, 10
    H, W = hs * f, ws * f
    g = torch.Generator().manual_seed(f)
    gup = torch.randn(N, 1, H, W, generator=g)
    d = torch.zeros(N, 1, hs, ws, requires_grad=True)
    F.interpolate(d, size=(H, W), mode="bilinear", align_corners=False).backward(gup)
    out = []
    for generic in (0, 1):
        gd = torch.empty(N, 1, hs, ws, device="cuda")
        check(lib.pd_up_gather_bwd(ptr(gup.cuda()), ptr(gd), N, hs, ws, H, W, 0, generic, None), "pd_up_gather_bwd")
        torch.cuda.synchronize()
        out.append(gd.cpu())
    assert torch.equal(out[0], out[1])
    _close(out[0], d.grad, 1e-6, "up gather")


def test_loss_vs_oracle_other_geometry_and_partial_scales():
    from oracle import losses as ol
    g = torch.Generator().manual_seed(21)
    N, H, W = 3, 96, 160
    scales = [0, 2]
    # smooth GT surface + mild noise: white-noise depth makes the normals term so ill-conditioned that
    # the fp32 CPU oracle itself is 6e-2 (relative) away from its fp64 evaluation
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    gt = (1.0 + 0.5 * torch.sin(xx / 23.0) * torch.cos(yy / 17.0))[None, None].repeat(N, 1, 1, 1)
    gt = gt + 0.02 * torch.rand(N, 1, H, W, generator=g)
    gt[torch.rand(N, 1, H, W, generator=g) < 0.15] = 0
    K = torch.eye(4)[None].repeat(N, 1, 1)
    K[:, 0, 0] = 100.0; K[:, 1, 1] = 110.0; K[:, 0, 2] = 75.0; K[:, 1, 2] = 50.0
    disps = {s: torch.sigmoid(torch.randn(N, 1, H >> s, W >> s, generator=g)) for s in scales}
    colors = {s: torch.rand(N, 3, H >> s, W >> s, generator=g) for s in scales}
    def oracle(dt):
        dl = {("disp", s): disps[s].clone().to(dt).requires_grad_(True) for s in scales}
        outputs = dict(dl)
        for s in scales:
            outputs[("depth", 0, s)] = ol.upsample_disp_to_depth(dl[("disp", s)], H, W, 0.1, 2.0)
        inputs = {("color", 0, s): colors[s].to(dt) for s in scales}
        inputs["depth"] = gt.to(dt); inputs[("K", 0)] = K.to(dt)
        L = ol.compute_losses(inputs, outputs, scales=scales, normals_loss_weight=0.35)
        L["loss"].backward()
        return L, [dl[("disp", s)].grad for s in scales]

    L, g32 = oracle(torch.float32)
    _, g64 = oracle(torch.float64)
    cfg = PF.LossCfg(scales, 0.1, 2.0, 0.35, 1e-3, H, W)
    dd = [disps[s].cuda().requires_grad_(True) for s in scales]
    vals, depths = PF.multiscale_loss(cfg, gt.cuda(), K.cuda(), dd, [colors[s].cuda() for s in scales])
    vals[0].backward()
    _close(vals[0], L["loss"], 1e-5, "loss")
    for i, s in enumerate(scales):
        _close(vals[3 + 3 * i], L[f"normals_loss/{s}"], 1e-5, "normals")
        # Random (white-noise) predicted disparities make a handful of pixels ill-conditioned: the fp32
        # CPU oracle is itself ~6e-2*scale away from its own fp64 evaluation there.  Criterion: mean error
        # tiny, no more outliers beyond 2e-4*scale than ~2x the fp32 oracle's own, worst pixel within 3x the oracle's own noise.
        ref = g64[i].float()
        scale = ref.abs().max().item()
        err = (dd[i].grad.cpu() - ref).abs()
        noise = (g32[i] - ref).abs().max().item()
        assert err.mean().item() <= max(5e-5 * scale, 2 * (g32[i] - ref).abs().mean().item())
        n_out_oracle = ((g32[i] - ref).abs() > 2e-4 * scale).sum().item()
        assert (err > 2e-4 * scale).sum().item() <= 2 * n_out_oracle + 2
        assert err.max().item() <= 3 * noise + 2e-4 * scale, (err.max().item(), noise, scale)


def test_ssim_and_reprojection_loss_match_reference_fixture():
    from polardepth import ops
    x, y = T(G5["ssim.x"]).cuda(), T(G5["ssim.y"]).cuda()
    got = ops.ssim(x, y)
    _close(got, T(G5["ssim.out"]), 2e-5, "ssim")
    rep = ops.reprojection_loss(x, y)
    ref = 0.85 * T(G5["ssim.out"]).mean(1, True) + 0.15 * (T(G5["ssim.y"]) - T(G5["ssim.x"])).abs().mean(1, True)
    _close(rep, ref, 2e-5, "reprojection")
    _close(ops.reprojection_loss(x, y, no_ssim=True), (T(G5["ssim.y"]) - T(G5["ssim.x"])).abs().mean(1, True), 1e-6, "l1")
    from manydepth.layers import SSIM
    _close(SSIM()(x, y), T(G5["ssim.out"]), 2e-5, "layers.SSIM")


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 3, 3, 4), (2, 1, 2, 9), (1, 3, 37, 21)])
def test_ssim_backward_matches_autograd_of_the_reference_formula(shape):
    """pd_ssim_bwd (both images, SSIM map and the 0.85 SSIM + 0.15 L1 photometric mix of trainer.py:1069-1081) against
    torch autograd through the reference's formula (oracle.losses.ssim, layers.py:468-499) in fp64; images small enough that
    both reflected borders of a window can fold onto the same pixel."""
    from polardepth import ops
    from oracle import losses as ol
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(N, C, H, W, generator=g)
    y = (x + 0.15 * torch.randn(N, C, H, W, generator=g)).clamp(0, 1)        # correlated: the clamp is mostly inactive
    w0 = torch.randn(N, C, H, W, generator=g)
    w1 = torch.randn(N, 1, H, W, generator=g)
    xr, yr = x.double().requires_grad_(True), y.double().requires_grad_(True)
    (ol.ssim(xr, yr) * w0.double()).sum().backward()
    xc, yc = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    out = ops.ssim(xc, yc)
    (out * w0.cuda()).sum().backward()
    inner = ((out > 0) & (out < 1)).float().mean().item()
    assert inner > 0.5
    _close(xc.grad, xr.grad.float(), 2e-4, "d ssim / dx")
    _close(yc.grad, yr.grad.float(), 2e-4, "d ssim / dy")
    # photometric mix; |target - pred| has a kink where they are equal: none in random data
    xr2, yr2 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    rep = 0.85 * ol.ssim(xr2, yr2).mean(1, True) + 0.15 * (yr2 - xr2).abs().mean(1, True)
    (rep * w1.double()).sum().backward()
    xc2, yc2 = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    (ops.reprojection_loss(xc2, yc2) * w1.cuda()).sum().backward()
    _close(xc2.grad, xr2.grad.float(), 2e-4, "d reprojection / d pred")
    _close(yc2.grad, yr2.grad.float(), 2e-4, "d reprojection / d target")
    xc3 = x.cuda().requires_grad_(True)
    (ops.reprojection_loss(xc3, y.cuda(), no_ssim=True) * w1.cuda()).sum().backward()
    xr3 = x.double().requires_grad_(True)
    ((y.double() - xr3).abs().mean(1, True) * w1.double()).sum().backward()
    _close(xc3.grad, xr3.grad.float(), 1e-6, "d L1 / d pred")


def test_depth_metrics_match_reference_fixture():
    from polardepth import ops
    gt, pr = T(G5["err.gt"]), T(G5["err.pred"])          # 500 values in (0.2, 1.7): all inside (0.1, 2.0)
    m = ops.depth_metrics(gt.cuda()[None], pr.cuda()[None], 0.1, 2.0)
    _close(m[0, :7], T(G5["err.out"]), 1e-5, "metrics")
    assert m[0, 7].item() == 500
    # per-material selection + clamping + invalid pixels vs the numpy restatement
    import numpy as np
    from oracle import losses as ol
    g = torch.Generator().manual_seed(2)
    gt2 = 0.05 + 2.2 * torch.rand(3, 1, 32, 40, generator=g)
    pr2 = 0.05 + 2.2 * torch.rand(3, 1, 32, 40, generator=g)
    mask = (torch.randint(0, 11, (3, 1, 32, 40), generator=g) * 20).int()
    got = ops.depth_metrics(gt2.cuda(), pr2.cuda(), 0.1, 2.0, mask=mask.cuda(), mask_value=160).cpu()
    for n in range(3):
        sel = (gt2[n] > 0.1) & (gt2[n] < 2.0) & (mask[n] == 160)
        ref = torch.stack(ol.compute_depth_errors(gt2[n][sel], pr2[n][sel].clamp(0.1, 2.0)))
        _close(got[n, :7], ref, 1e-5, f"img{n}")
        assert got[n, 7].item() == sel.sum().item()
