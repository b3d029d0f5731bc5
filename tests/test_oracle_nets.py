"""Pins oracle/nets.py and oracle/losses.py against fixtures produced by running the reference's
own modules (tests/golden/make_golden.py: pre_encoders, depth_decoder, layers, Trainer.compute_losses)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
sys.path.insert(0, GOLDEN)
from synth_weights import fill_state_dict  # noqa: E402

from oracle import nets, losses  # noqa: E402

G4 = np.load(os.path.join(GOLDEN, "g4_nets.npz"))
G5 = np.load(os.path.join(GOLDEN, "g5_loss.npz"))
T = lambda a: torch.from_numpy(np.asarray(a))


def _close(a, b, tol=1e-5):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    scale = b.abs().max().item() + 1e-12
    assert (a - b).abs().max().item() <= tol * scale, ((a - b).abs().max().item(), scale)


def _check_module(name, mod, args, input_grad=True):
    fill_state_dict(mod, 0, prefix=name + ".")
    mod.eval()
    with torch.no_grad():
        y = mod(*args)
    ys = y if isinstance(y, (list, tuple)) else [y]
    for i, t in enumerate(ys):
        _close(t, T(G4[f"{name}.eval.{i}"]), 2e-6)
    mod.train()
    args_g = [a.clone().requires_grad_(input_grad) if a is not None else None for a in args]
    y = mod(*args_g)
    ys = y if isinstance(y, (list, tuple)) else [y]
    obj = 0
    for i, t in enumerate(ys):
        _close(t, T(G4[f"{name}.train.{i}"]), 2e-6)
        obj = obj + (t * torch.randn(t.shape, generator=torch.Generator().manual_seed(100 + i))).sum()
    obj.backward()
    n = 0
    for k, p in mod.named_parameters():
        if f"{name}.grad.{k}" in G4:
            _close(p.grad, T(G4[f"{name}.grad.{k}"]), 2e-4); n += 1
        elif f"{name}.gradsample.{k}" in G4:
            g = p.grad.flatten()
            _close(g[::max(1, g.numel() // 4096)], T(G4[f"{name}.gradsample.{k}"]), 2e-4); n += 1
    assert n > 10
    for i, a in enumerate(args_g):
        if a is not None and a.grad is not None:
            _close(a.grad, T(G4[f"{name}.ingrad.{i}"]), 2e-4)


def test_shallow_encoders_match_reference():
    xolp = T(G4["xolp"])
    _check_module("xolp_encoder", nets.ShallowEncoder('XOLP', 2, 0.0), [xolp], input_grad=False)
    _check_module("normals_encoder", nets.ShallowNormalsEncoder(9, 0.0), [xolp], input_grad=False)


@pytest.mark.parametrize("name,inc_n,inc_x", [("joint3", True, True), ("joint_x", False, True),
                                              ("joint_n", True, False), ("joint_rgb", False, False)])
def test_joint_encoder_matches_reference(name, inc_n, inc_x):
    rgbf, xf, nf = T(G4["joint.rgbf"]), T(G4["joint.xf"]), T(G4["joint.nf"])
    _check_module(name, nets.JointEncoder(0.0, inc_n, inc_x), [rgbf, xf if inc_x else None, nf if inc_n else None])


def test_depth_decoder_matches_reference():
    feats = [T(G4[f"dec.feat.{i}"]).clone().requires_grad_(True) for i in range(5)]
    dd = nets.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4))
    fill_state_dict(dd, 0, prefix="mono_depth.")
    res = dd(feats)
    obj = 0
    for s in range(4):
        _close(res[("disp", s)], T(G4[f"dec.disp.{s}"]), 2e-6)
        obj = obj + (res[("disp", s)] * torch.randn(res[("disp", s)].shape, generator=torch.Generator().manual_seed(200 + s))).sum()
    obj.backward()
    for i, f in enumerate(feats):
        _close(f.grad, T(G4[f"dec.featgrad.{i}"]), 1e-4)
    assert list(dd.state_dict().keys())[:2] == ["decoder.0.conv.conv.weight", "decoder.0.conv.conv.bias"]
    assert "decoder.13.conv.weight" in dd.state_dict()


def test_resnet_stem_structure_and_keys():
    """torchvision is absent: pinned by structure (key names, shapes, parameter count) only."""
    enc = nets.ShallowResnetEncoder(18, False)
    sd = enc.state_dict()
    assert sd["encoder.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["encoder.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["encoder.layer4.1.bn2.running_var"].shape == (512,) and sd["encoder.fc.weight"].shape == (1000, 512)
    assert sum(p.numel() for p in enc.parameters()) == 11689512          # torchvision resnet18
    f = enc(torch.rand(1, 3, 64, 96))
    assert [tuple(t.shape[1:]) for t in f] == [(64, 32, 48), (64, 16, 24), (128, 8, 12)]


def test_losses_match_reference_trainer():
    inputs = {("color", 0, s): T(G5[f"in.color_0_{s}"]) for s in range(4)}
    inputs["depth"] = T(G5["in.depth"]); inputs[("K", 0)] = T(G5["in.K_0"])
    for tag, lam in (("lam0", 0.0), ("lam035", 0.35)):
        disps = {("disp", s): T(G5[f"disp.{s}"]).clone().requires_grad_(True) for s in range(4)}
        outputs = dict(disps)
        for s in range(4):
            outputs[("depth", 0, s)] = losses.upsample_disp_to_depth(disps[("disp", s)], 64, 96, 0.1, 2.0)
            _close(outputs[("depth", 0, s)], T(G5[f"{tag}.depth.{s}"]), 1e-6)
        L = losses.compute_losses(inputs, outputs, normals_loss_weight=lam)
        L["loss"].backward()
        for k in ("loss", "loss/0", "loss/3", "supervised_depth_loss/0", "supervised_depth_loss/2"):
            _close(L[k], T(G5[f"{tag}.{k}"]), 1e-6)
        for s in range(4):
            _close(disps[("disp", s)].grad, T(G5[f"{tag}.ddisp.{s}"]), 1e-5)


def test_small_layer_functions_match_reference():
    _close(losses.ssim(T(G5["ssim.x"]), T(G5["ssim.y"])), T(G5["ssim.out"]), 1e-6)
    _close(losses.get_smooth_loss(T(G5["smooth.disp"]), T(G5["ssim.x"])), T(G5["smooth.out"]), 1e-6)
    sd, dp = losses.disp_to_depth(T(G5["smooth.disp"]), 0.1, 2.0)
    assert torch.equal(sd, T(G5["d2d.scaled"])) and torch.equal(dp, T(G5["d2d.depth"]))
    _close(torch.stack(losses.compute_depth_errors(T(G5["err.gt"]), T(G5["err.pred"]))), T(G5["err.out"]), 1e-6)


def test_depth_to_normals_restatement_properties():
    """kornia is absent (parity unpinned): check the analytic case of a fronto-parallel plane and a tilted plane."""
    B, H, W = 1, 12, 16
    K = torch.eye(3)[None].clone(); K[:, 0, 0] = K[:, 1, 1] = 20.0; K[:, 0, 2] = 8.0; K[:, 1, 2] = 6.0
    flat = torch.full((B, 1, H, W), 1.5)
    n = losses.depth_to_normals(flat, K)
    assert torch.allclose(n[:, 2, 2:-2, 2:-2], torch.ones(B, H - 4, W - 4), atol=1e-6)
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    # plane z = z0 / (1 - a*x) with x = (u-cx)/fx has normal ~ (-a, 0, 1)/|.|
    a = 0.3
    depth = (1.0 / (1 - a * (u - 8.0) / 20.0))[None, None]
    n = losses.depth_to_normals(depth, K)[0, :, 3:-3, 3:-3]
    exp = torch.tensor([-a, 0.0, 1.0]) / (1 + a * a) ** 0.5
    assert (n - exp[:, None, None]).abs().max() < 2e-2


def test_normals_decoder_variant_oracle_definition():
    """`arch1++_separate_normals_dec` (README.md:54) as this build defines it: three ConvBlock + bilinear x2 stages from the
    normals encoder's 64 x H/8 x W/8 features to a 3-channel map at H x W; its loss is 1 for a perfect prediction
    (2 - cos = 1), 3 for the opposite normal, and ignores pixels outside the depth range."""
    from oracle import nets as onets, losses as ol
    torch.manual_seed(0)
    dec = onets.NormalsDecoder(64)
    assert list(dec.state_dict()) == [f"decoder.{i}.conv.conv.{p}" for i in range(3) for p in ("weight", "bias")] + \
        ["decoder.3.conv.weight", "decoder.3.conv.bias"]
    assert dec(torch.randn(1, 64, 4, 6)).shape == (1, 3, 32, 48)
    gt = 0.5 + torch.rand(2, 1, 16, 24)
    gt[:, :, :3] = 0.0
    K = torch.eye(4)[None].repeat(2, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 20.0; K[:, 0, 2] = 12; K[:, 1, 2] = 8
    n = ol.depth_to_normals(gt, K[:, :3, :3])
    assert abs(ol.normals_pred_loss(5.0 * n, gt, K).item() - 1.0) < 1e-6
    assert abs(ol.normals_pred_loss(-n, gt, K).item() - 3.0) < 1e-6
    junk = n.clone(); junk[:, :, :3] = 123.0            # rows outside the depth range do not count
    assert abs(ol.normals_pred_loss(junk, gt, K).item() - 1.0) < 1e-6
