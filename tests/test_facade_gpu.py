"""The drop-in functions SURVEY.md section 8(b) lists beside the Trainer / network classes, each against the fixture the
reference itself produced (tests/golden/make_golden.py) or the CPU oracle:
  manydepth.normals_vec.rho_diffuse / rho_spec / calc_normals      g2_theta.npz   (bit-equal fp64, incl. extrapolation)
  polarisation.xolp.Iun_and_xolp                                   g1_xolp.npz
  manydepth.layers.Conv5x5                                         torch fp32 CPU (2e-5)
  Trainer.compute_supervised_normals_losses (caller's mask)        oracle.losses.normals_loss (1e-5)"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_rho_diffuse_and_rho_spec_return_the_reference_thetas(golden_dir):
    """normals_vec.py:11-50 through pd_polar_theta: fp64 thetas bit-equal to what scipy's interp1d(extrapolate) returned
    inside the reference, on a DoLP sweep 0 .. 2.2 that includes exact table nodes and the far extrapolation (theta_s1
    beyond +40 rad, theta_s2 below -130 rad), for n = 1.5 and n = 1.3; CPU fp64 tensors of rho's shape like the reference."""
    from manydepth import normals_vec
    g = np.load(os.path.join(golden_dir, "g2_theta.npz"))
    rho = torch.from_numpy(g["rho"]).reshape(1, 1, -1).cuda()
    th_d = normals_vec.rho_diffuse(rho, 1.5)
    th_1, th_2 = normals_vec.rho_spec(rho, 1.5)
    for got, key in ((th_d, "theta_d"), (th_1, "theta_s1"), (th_2, "theta_s2")):
        assert got.dtype == torch.float64 and got.device.type == "cpu" and tuple(got.shape) == g[key].shape
        assert np.array_equal(got.numpy(), g[key]), f"{key}: max diff {np.abs(got.numpy() - g[key]).max():.3e}"
    assert g["theta_s1"].max() > 40 and g["theta_s2"].min() < -130          # the fixture does reach the wild values
    assert np.array_equal(normals_vec.rho_diffuse(rho, 1.3).numpy(), g["theta_d_n13"])
    # a CPU input works like in the reference (rho.cpu().numpy() there)
    assert np.array_equal(normals_vec.rho_diffuse(rho.cpu(), 1.5).numpy(), g["theta_d"])
    # calc_normals: fp32 phi with fp64 theta promotes to fp64, [B,3,H,W]
    phi = torch.linspace(-1.5, 1.5, rho.numel()).reshape(1, 1, -1).cuda()
    N = normals_vec.calc_normals(phi, th_d)
    assert N.dtype == torch.float64 and tuple(N.shape) == (1, 3, 1, rho.numel()) and N.is_cuda
    ref = torch.stack((torch.cos(phi.cpu()) * torch.sin(th_d), torch.sin(phi.cpu()) * torch.sin(th_d), torch.cos(th_d)), 1)
    assert (N.cpu() - ref).abs().max().item() < 1e-6


def test_theta_bins_are_the_searchsorted_indices():
    from polardepth import polar as pdpolar
    from oracle import polar as opolar
    rng = np.random.default_rng(3)
    rho = np.concatenate([rng.random(5000).astype(np.float32) * 2.2, np.float32([0, 1e-30, 0.3846153, 0.999999, 1.0, 2.2])])
    out = pdpolar.theta_from_rho(torch.from_numpy(rho).cuda(), 1.5, want_bins=True)
    td, id_ = opolar.rho_diffuse(rho, 1.5, return_idx=True)
    t1, t2, i1, i2 = opolar.rho_spec(rho, 1.5, return_idx=True)
    bins = out["bins"].cpu().numpy()
    assert np.array_equal(bins[0], id_) and np.array_equal(bins[1], i1) and np.array_equal(bins[2], i2)
    assert np.array_equal(out["d"].cpu().numpy(), np.asarray(td)) and np.array_equal(out["s2"].cpu().numpy(), np.asarray(t2))


def _circ_dist_mod_pi(a, b):
    d = np.abs(a - b) % np.pi
    return np.minimum(d, np.pi - d)


def test_iun_and_xolp_facade_against_the_reference_fixture(golden_dir):
    """polarisation.xolp.Iun_and_xolp on K1 vs the outputs of the reference function: Iun exact, DoLP = the fp32 rounding
    of the reference's fp64 value on > 99.5 % of the pixels and within one fp32 ulp elsewhere (lstsq noise decides the
    rounding), AoLP equal modulo pi to fp32 precision wherever the pixel is polarised, flips only on the branch cut."""
    from polarisation.xolp import Iun_and_xolp
    g = np.load(os.path.join(golden_dir, "g1_xolp.npz"))
    for name in ("edge", "rnd", "phys"):
        img = g[name + "_img"]
        Iun, rho, phi = Iun_and_xolp(img, np.array([0, 45, 90, 135]) * np.pi / 180)
        assert Iun.dtype == np.float64 and rho.shape == img.shape[:2]
        np.testing.assert_allclose(Iun, g[name + "_Iun"], rtol=0, atol=1e-11)
        ref32 = g[name + "_rho"].astype(np.float32)
        assert np.abs(rho - g[name + "_rho"]).max() <= 1.3e-7 * max(1.0, np.abs(g[name + "_rho"]).max())
        if name != "edge":
            assert (rho.astype(np.float32) == ref32).mean() > 0.995
        d1 = img[..., 0].astype(int) - img[..., 2]
        d2 = img[..., 1].astype(int) - img[..., 3]
        pol = (d1 != 0) | (d2 != 0)
        assert np.all(phi[~pol] == 0)
        assert _circ_dist_mod_pi(phi, g[name + "_phi"])[pol].max() < 2e-7
        flips = (np.abs(phi - g[name + "_phi"]) > 1e-6) & pol
        if flips.any():
            assert np.all((d2[flips] == 0) & (d1[flips] < 0))


@pytest.mark.parametrize("case", [(2, 16, 20, 28, 1, True), (1, 32, 9, 11, 1, True), (2, 16, 12, 16, 16, True),
                                  (1, 64, 6, 7, 3, True), (2, 16, 10, 12, 1, False)])
def test_conv5x5_matches_torch(case):
    """manydepth.layers.Conv5x5 (layers.py:345-362: ReflectionPad2d(2) or ZeroPad2d(2) + Conv2d(5)), forward and all three
    gradients against PyTorch fp32 on the CPU, including images whose two reflected borders meet (H = 6)."""
    from manydepth.layers import Conv5x5
    from polardepth import functional as PF
    N, C, H, W, Co, refl = case
    torch.manual_seed(sum(case[:5]))
    m = Conv5x5(C, Co, use_refl=refl)
    assert set(m.state_dict()) == {"conv.weight", "conv.bias"}
    x = torch.randn(N, C, H, W)
    gy = torch.randn(N, Co, H, W)
    xr = x.clone().requires_grad_(True)
    wr = m.conv.weight.detach().clone().contiguous().requires_grad_(True)
    br = m.conv.bias.detach().clone().requires_grad_(True)
    xp = F.pad(xr, (2, 2, 2, 2), mode="reflect") if refl else F.pad(xr, (2, 2, 2, 2))
    yr = F.conv2d(xp, wr, br)
    (yr * gy).sum().backward()
    m = m.cuda()
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(xc)
    (y * gy.cuda()).sum().backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()

    def close(a, b, what):
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item()
        assert err <= 2e-5 * scale, f"{what}: {err:.3e} vs {scale:.3e}"
    close(y.detach().cpu(), yr.detach(), "fwd")
    close(xc.grad.cpu(), xr.grad, "dx")
    close(m.conv.weight.grad.cpu(), wr.grad, "dw")
    close(m.conv.bias.grad.cpu(), br.grad, "db")


def test_supervised_normals_loss_honours_the_callers_mask(tmp_path):
    """Trainer.compute_supervised_normals_losses(depth_gt, depth_pred, intrinsics, mask) (trainer.py:1298-1309) with the
    depth-range mask the reference passes, with an arbitrary 0/1 mask, and with a weighted mask, vs the oracle."""
    from test_step_gpu import _opts
    from manydepth.trainer import Trainer
    from oracle import losses as ol
    tr = Trainer(_opts(tmp_path))
    g = torch.Generator().manual_seed(4)
    N, H, W = 2, 64, 96
    gt = 0.2 + 2.0 * torch.rand(N, 1, H, W, generator=g)
    gt[:, :, :, 90:] = 0
    pred = 0.3 + 1.5 * torch.rand(N, 1, H, W, generator=g)
    K = torch.eye(4)[None].repeat(N, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 0.65 * W; K[:, 0, 2] = W / 2; K[:, 1, 2] = H / 2
    masks = {"range": ((gt >= 0.1) & (gt <= 2.0)).float(),
             "random": (torch.rand(N, 1, H, W, generator=g) < 0.3).float(),
             "weighted": torch.rand(N, 1, H, W, generator=g)}
    for name, mask in masks.items():
        ref = ol.normals_loss(gt, pred, K, mask).item()
        got = tr.compute_supervised_normals_losses(gt.cuda(), pred.cuda(), K.cuda(), mask.cuda()).item()
        assert abs(got - ref) <= 1e-5 * abs(ref), (name, got, ref)
    ref = ol.normals_loss(gt, pred, K, masks["range"]).item()
    got = tr.compute_supervised_normals_losses(gt.cuda(), pred.cuda(), K.cuda()).item()
    assert abs(got - ref) <= 1e-5 * abs(ref)


def test_layers_helpers_run_on_the_kernels_and_match_the_reference_fixture():
    """manydepth.layers.get_smooth_loss / compute_depth_errors / compute_depth_errors_numpy (reference layers.py:452-465,
    539-577) are served by pd_smooth_fwd / pd_smooth_bwd (mean = NULL) and pd_depth_metrics: values against fixture G5
    (outputs of the reference's own functions), the smoothness gradient against autograd of the formula on the CPU."""
    import os
    from conftest import ROOT
    from manydepth import layers
    G5 = np.load(os.path.join(ROOT, "tests", "golden", "g5_loss.npz"))
    T = lambda a: torch.from_numpy(np.asarray(a))
    disp, img = T(G5["smooth.disp"]).float(), T(G5["ssim.x"]).float()
    d = disp.cuda().requires_grad_(True)
    out = layers.get_smooth_loss(d, img.cuda())
    assert out.dim() == 0 and abs(out.item() - float(G5["smooth.out"])) <= 1e-6 * abs(float(G5["smooth.out"])) + 1e-9
    out.backward()
    dc = disp.clone().requires_grad_(True)
    gx = (dc[:, :, :, :-1] - dc[:, :, :, 1:]).abs() * torch.exp(-(img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True))
    gy = (dc[:, :, :-1, :] - dc[:, :, 1:, :]).abs() * torch.exp(-(img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True))
    (gx.mean() + gy.mean()).backward()
    assert (d.grad.cpu() - dc.grad).abs().max().item() <= 1e-6 * dc.grad.abs().max().item() + 1e-10
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.get_smooth_loss(disp, img)
    gt, pred = T(G5["err.gt"]).float(), T(G5["err.pred"]).float()
    want = np.asarray(G5["err.out"], dtype=np.float64)
    got = torch.stack(layers.compute_depth_errors(gt.cuda(), pred.cuda())).cpu().double().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(np.array(layers.compute_depth_errors_numpy(gt.numpy(), pred.numpy())), want, rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError):
        layers.SSIM()(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))          # no CPU branch
