"""One full training step (K1 -> 3 encoders -> joint -> decoder -> K5 loss -> backward -> fused Adam)
through the Trainer façade vs the CPU oracle with identical weights and batch (dropout 0, BN in
training mode).  Tolerance: loss 1e-4 relative (north_star); gradients 5e-3 relative L2 per tensor -- the
end-to-end gradient is ill-conditioned at this tiny size (batch 2, 64x96, training-mode BatchNorm over ~20
layers): perturbing the oracle's own input by 1e-7 relative moves its gradients by 1e-5, i.e. fp32 rounding
(6e-8 per operation, different summation orders on the two sides) is amplified ~100x.  The tight per-module
gradient checks (5e-4) are in test_modules_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
sys.path.insert(0, GOLDEN)
from synth_weights import fill_state_dict  # noqa: E402

pytestmark = pytest.mark.gpu


def _opts(tmp, extra=(), encoders=("--augment_xolp", "--augment_normals")):
    from manydepth.options import MonodepthOptions
    return MonodepthOptions().parse([
        "--png", "--batch_size", "2", "--height", "64", "--width", "96", "--dataset", "HAMMER", "--split", "HAMMER",
        "--eval_split", "HAMMER_unseen", "--min_depth", "0.1", "--max_depth", "2.0", "--depth_supervision_only", "True",
        "--depth_supervision", "True", "--normals_loss_weight", "0.35", *encoders,
        "--log_dir", str(tmp), "--data_path", "synthetic", "--data_path_val", "synthetic", "--num_workers", "0",
        "--weights_init", "scratch", "--learning_rate", "1e-4", *extra])


@pytest.mark.parametrize("encoders", [(), ("--augment_xolp",), ("--augment_normals",)])
def test_training_step_other_encoder_sets_match_oracle(tmp_path, encoders):
    """BASELINE configs[0] (RGB only) and configs[1] (RGB + XOLP) -- and RGB + normals -- through the same façade:
    disparities 2e-5, loss 1e-4 relative vs the CPU oracle, and a finite optimizer step."""
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from oracle import nets as onets, losses as ol, polar as opolar
    tr = Trainer(_opts(tmp_path, ["--dropout_rate", "0.0"], encoders))
    ax, an = "--augment_xolp" in encoders, "--augment_normals" in encoders
    ref = onets.build_models(ax, an, 0.0)
    assert set(ref) == set(tr.models)
    for name, m in ref.items():
        fill_state_dict(m, 0, prefix=name + ".")
        tr.models[name].load_state_dict(m.state_dict())
        m.train()
    tr.set_train()
    batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=9)
    cpu = {k: v.cpu() for k, v in batch.items()}
    tr.model_optimizer.zero_grad()
    outputs, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    tr.model_optimizer.step()
    torch.cuda.synchronize()
    xolp, _, _, _ = opolar.polar_forward(cpu[("pol", 0, 0)].numpy())
    outs = onets.forward_models(ref, cpu[("color_aug", 0, 0)], xolp)
    ro = dict(outs)
    for s in range(4):
        ro[("depth", 0, s)] = ol.upsample_disp_to_depth(outs[("disp", s)], 64, 96, 0.1, 2.0)
        assert (outputs[("disp", s)].detach().cpu() - outs[("disp", s)].detach()).abs().max().item() < 2e-5, f"disp {s}"
    L = ol.compute_losses(cpu, ro, normals_loss_weight=0.35)
    rel = abs(losses["loss"].item() - L["loss"].item()) / abs(L["loss"].item())
    assert rel < 1e-4, (losses["loss"].item(), L["loss"].item())
    assert torch.isfinite(tr.store.flat).all()


def test_one_training_step_matches_oracle(tmp_path):
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from oracle import nets as onets, losses as ol, polar as opolar
    tr = Trainer(_opts(tmp_path, ["--dropout_rate", "0.0"]))
    ref = onets.build_models(True, True, 0.0)
    for name, m in ref.items():
        fill_state_dict(m, 0, prefix=name + ".")
        tr.models[name].load_state_dict(m.state_dict())
        m.train()
    tr.set_train()
    batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=5)
    cpu = {k: v.cpu() for k, v in batch.items()}

    tr.model_optimizer.zero_grad()
    outputs, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    gpu_grads = {n: p.grad.detach().cpu().clone() for mn in tr.models for n, p in
                 ((f"{mn}.{k}", v) for k, v in tr.models[mn].named_parameters())}
    before = tr.store.flat.clone()
    tr.model_optimizer.step()
    torch.cuda.synchronize()

    # ---- oracle
    xolp, _, _, _ = opolar.polar_forward(cpu[("pol", 0, 0)].numpy())
    assert torch.equal(batch[("xolp", 0, 0)].cpu() if ("xolp", 0, 0) in batch else xolp, xolp)
    outs = onets.forward_models(ref, cpu[("color_aug", 0, 0)], xolp)
    ro = dict(outs)
    for s in range(4):
        ro[("depth", 0, s)] = ol.upsample_disp_to_depth(outs[("disp", s)], 64, 96, 0.1, 2.0)
    L = ol.compute_losses(cpu, ro, normals_loss_weight=0.35)
    L["loss"].backward()

    for s in range(4):
        d = outputs[("disp", s)].detach().cpu()
        assert (d - outs[("disp", s)].detach()).abs().max().item() < 2e-5, f"disp {s}"
    rel = abs(losses["loss"].item() - L["loss"].item()) / abs(L["loss"].item())
    assert rel < 1e-4, (losses["loss"].item(), L["loss"].item())
    for k in ("loss/0", "loss/3", "supervised_depth_loss/1"):
        assert abs(losses[k].item() - L[k].item()) <= 1e-4 * abs(L[k].item())
    checked = 0
    for mn, m in ref.items():
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            g = gpu_grads[f"{mn}.{k}"]
            if k.endswith("conv.bias") and mn != "mono_depth":
                continue                     # bias in front of BatchNorm: exact 0 here, rounding noise in torch
            # relative L2 error per tensor: isolated ReLU / max-pool decision flips under fp32 re-ordering give
            # O(1) deviations in single entries at this size (48 samples per channel at 1/16 resolution)
            rel_l2 = ((g - p.grad).norm() / (p.grad.norm() + 1e-20)).item()
            assert rel_l2 <= 5e-3, f"{mn}.{k}: rel L2 {rel_l2:.2e}"
            checked += 1
    assert checked > 150
    # unused resnet parameters keep a zero gradient and are not touched by Adam
    assert gpu_grads["rgb_encoder.encoder.layer4.1.conv2.weight"].abs().max().item() == 0
    off, n = tr.store.offsets["rgb_encoder.encoder.fc.weight"]
    assert torch.equal(tr.store.flat[off:off + n], before[off:off + n])
    # Adam first step: every used parameter with a non-negligible gradient moves by ~lr against its sign
    off, n = tr.store.offsets["mono_depth.decoder.0.conv.conv.weight"]
    delta = (tr.store.flat[off:off + n] - before[off:off + n]).cpu()
    # (FusedAdam.step clears the gradient buffer in the pass that consumes it: use the copy taken before the step)
    g = gpu_grads["mono_depth.decoder.0.conv.conv.weight"].permute(0, 2, 3, 1).reshape(-1)   # storage order of the flat buffer
    assert tr.store.grad[off:off + n].abs().max().item() == 0 and tr.store.grad_is_zero
    big = g.abs() > 1e-6
    assert torch.allclose(delta[big], -1e-4 * torch.sign(g[big]), atol=2e-6)


def test_checkpoint_roundtrip_and_eval_mode(tmp_path):
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    tr = Trainer(_opts(tmp_path))
    batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=6)
    tr.set_train()
    tr.model_optimizer.zero_grad()
    _, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    tr.model_optimizer.step()
    tr.save_model()
    folder = os.path.join(tr.log_path, "models", "weights_0")
    assert sorted(os.listdir(folder)) == ["adam.pth", "joint_encoder.pth", "mono_depth.pth", "normals_encoder.pth",
                                          "rgb_encoder.pth", "trainer_state.pth", "xolp_encoder.pth"]
    sd = torch.load(os.path.join(folder, "rgb_encoder.pth"))
    assert "encoder.layer4.1.bn2.running_var" in sd and sd["encoder.conv1.weight"].is_contiguous()
    tr.set_eval()
    with torch.no_grad():
        out1, l1, _ = tr.process_batch(dict(batch))
    opts2 = _opts(tmp_path, ["--load_weights_folder", folder, "--models_to_load", "rgb_encoder", "xolp_encoder",
                             "normals_encoder", "joint_encoder", "mono_depth"])
    tr2 = Trainer(opts2)
    tr2.set_eval()
    with torch.no_grad():
        out2, l2, _ = tr2.process_batch(dict(batch))
    assert torch.equal(out1[("disp", 0)], out2[("disp", 0)]) and torch.equal(l1["loss"], l2["loss"])
    assert tr2.model_optimizer.step_count == 1
    assert (tr2.resume_epoch, tr2.resume_step) == (tr.epoch + 1, tr.step)        # trainer_state.pth: resume point
    assert torch.equal(tr2.model_optimizer.exp_avg, tr.model_optimizer.exp_avg)


def test_dropout_is_deterministic_per_seed_and_matches_backward(tmp_path):
    from polardepth import functional as PF
    from manydepth.networks.pre_encoders import ConvBlock
    torch.manual_seed(0)
    blk = ConvBlock(64, 64, 3, 'none', 1, 0.3).cuda().train()
    x = torch.randn(2, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    PF.DropoutState.manual_seed(123)
    y1 = blk(x)
    PF.DropoutState.manual_seed(123)
    y2 = blk(x)
    assert torch.equal(y1, y2)
    frac = (y1 == 0).float().mean().item()
    assert 0.55 < frac < 0.75          # relu zeros (~50%) + 30% dropout of the rest
    y1.sum().backward()
    assert torch.isfinite(x.grad).all()


def test_trainer_test_loop_runs_on_device_metrics(tmp_path, capsys):
    """Trainer.test(): forward in eval mode + device-side per-material metrics (smoke: finite numbers printed)."""
    from manydepth.trainer import Trainer
    tr = Trainer(_opts(tmp_path))
    tr.test()
    out = capsys.readouterr().out
    assert "abs_rel" in out and "glass" in out and "nan" not in out.lower()


def test_inference_with_folded_batchnorm_matches_oracle_and_unfolded_path(tmp_path):
    """Eval mode under no_grad folds every BatchNorm into its conv epilogue (pd_conv2d out_scale); the disparities
    must match the CPU oracle in eval mode and the unfolded kernels (conv -> finalize -> chain) to 2e-5 (fp32 rounding:
    fma(acc, s, t) vs (acc + b) * s + t through ~20 layers)."""
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from polardepth import functional as PF
    from oracle import nets as onets, polar as opolar
    tr = Trainer(_opts(tmp_path))
    ref = onets.build_models(True, True, 0.1)
    for name, m in ref.items():
        fill_state_dict(m, 0, prefix=name + ".")
        # non-trivial running statistics
        for k, b in m.named_buffers():
            if k.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=torch.Generator().manual_seed(len(k))))
            elif k.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=torch.Generator().manual_seed(len(k) + 1)))
        tr.models[name].load_state_dict(m.state_dict())
        m.eval()
    tr.set_eval()
    batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=11)
    assert PF.USE_BN_FOLDING
    with torch.no_grad():
        folded = tr._forward_models(dict(batch))
        PF.USE_BN_FOLDING = False
        try:
            plain = tr._forward_models(dict(batch))
        finally:
            PF.USE_BN_FOLDING = True
        xolp, _, _, _ = opolar.polar_forward(batch[("pol", 0, 0)].cpu().numpy())
        outs = onets.forward_models(ref, batch[("color_aug", 0, 0)].cpu(), xolp)
    for s in range(4):
        f, p = folded[("disp", s)].cpu(), plain[("disp", s)].cpu()
        assert (f - p).abs().max().item() < 2e-5, f"folded vs unfolded, scale {s}"
        assert (f - outs[("disp", s)]).abs().max().item() < 2e-5, f"folded vs oracle, scale {s}"


def test_raw_planes_are_resized_on_the_device_like_pillow(tmp_path):
    """Loader hand-over of native-size polarizer frames (HAMMER_Dataset(raw_pol=True)): the Trainer's device LANCZOS
    resize + K1 gives the same XOLP input and loss as PIL-resized planes."""
    from PIL import Image
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    tr = Trainer(_opts(tmp_path))
    tr.set_eval()
    batch = synthetic.make_batch(2, 64, 96, frame_w=96, device="cuda", seed=4)
    raw = torch.randint(0, 256, (2, 4, 160, 208), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    host = torch.from_numpy(np.stack([[np.asarray(Image.fromarray(raw[b, c].numpy(), "L").resize((96, 64), Image.LANCZOS))
                                       for c in range(4)] for b in range(2)]))
    with torch.no_grad():
        b1 = dict(batch); b1[("pol", 0, 0)] = raw.cuda()
        o1, l1, _ = tr.process_batch(b1)
        b2 = dict(batch); b2[("pol", 0, 0)] = host.cuda()
        o2, l2, _ = tr.process_batch(b2)
    assert torch.equal(b1[("xolp", 0, 0)], b2[("xolp", 0, 0)])
    assert torch.equal(o1[("disp", 0)], o2[("disp", 0)]) and torch.equal(l1["loss"], l2["loss"])


def test_reference_format_checkpoint_loads_and_continues_like_torch_adam(tmp_path):
    """A checkpoint folder as the REFERENCE writes it (trainer.py:1597-1617): per-model state_dicts of the reference's
    modules (here: the oracle restatement, same keys) and adam.pth = torch.optim.Adam(parameters_to_train) -- no
    pd_order, parameters numbered normals, xolp, joint, rgb, mono.  The Trainer loads it and its next step equals the
    step torch.optim.Adam takes from that state on the same gradients."""
    from manydepth.trainer import Trainer
    from oracle import nets as onets
    from polardepth import synthetic
    torch.manual_seed(0)
    ref = onets.build_models(True, True, 0.0)
    for m in ref.values():
        fill_state_dict(m)
    ref_order = ["normals_encoder", "xolp_encoder", "joint_encoder", "rgb_encoder", "mono_depth"]
    params = [p for n in ref_order for p in ref[n].parameters()]
    names = [f"{n}.{pn}" for n in ref_order for pn, _ in ref[n].named_parameters()]
    used = [not (n.startswith("rgb_encoder.encoder.") and n.split(".")[2] in ("layer3", "layer4", "fc")) for n in names]
    topt = torch.optim.Adam(params, 1e-4)
    g = torch.Generator().manual_seed(1)
    for p, u in zip(params, used):
        if u:
            p.grad = 1e-2 * torch.randn(p.shape, generator=g)
    topt.step()                                   # the state the reference would have saved after its first step
    folder = tmp_path / "ref_ckpt"
    folder.mkdir()
    for n, m in ref.items():
        torch.save(m.state_dict(), folder / f"{n}.pth")
    torch.save(topt.state_dict(), folder / "adam.pth")

    tr = Trainer(_opts(tmp_path, ["--dropout_rate", "0", "--load_weights_folder", str(folder), "--models_to_load",
                                  *ref_order]))
    assert tr.model_optimizer.step_count == 1
    for n, m in ref.items():                      # weights arrived
        for k, v in m.state_dict().items():
            assert torch.equal(tr.models[n].state_dict()[k].cpu(), v), (n, k)
    # one more step on both sides with the same gradients (taken from the GPU backward)
    batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=9)
    tr.set_train()
    tr.model_optimizer.zero_grad()
    _, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    torch.cuda.synchronize()
    mine = dict((f"{mn}.{pn}", p) for mn in tr.models for pn, p in tr.models[mn].named_parameters())
    for n, p, u in zip(names, params, used):
        p.grad = mine[n].grad.detach().cpu().clone() if u else None
    tr.model_optimizer.step()
    topt.step()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p, u in zip(names, params, used):
        if u:
            worst = max(worst, (mine[n].detach().cpu() - p.detach()).abs().max().item())
    assert worst < 2e-7, worst                    # second Adam step from the reference's moments: same update (fp32 rounding)


def test_evaluation_runs_on_the_device_with_per_material_metrics(tmp_path, capsys):
    """manydepth.evaluation.Evaluation (evaluation.py:120-288): forward in eval mode + per-material metrics reduced by
    pd_depth_metrics; the device numbers equal the reference's per-image loop (evaluation.py:215-288 with
    compute_depth_errors, restated in oracle/losses.py and pinned by fixture G5) on the same predictions."""
    from manydepth.evaluation import Evaluation, _MATERIAL_GREY
    from oracle import losses as ol

    def compute_depth_errors_numpy(gt, pred):
        return [float(v) for v in ol.compute_depth_errors(torch.from_numpy(gt), torch.from_numpy(pred))]
    with pytest.raises(FileNotFoundError):
        Evaluation(data_path=str(tmp_path / "missing"))
    ev = Evaluation(data_path="synthetic", height=64, width=96, batch_size=4)
    ev.load_mono_model()
    res = ev.test()
    assert "all" in res and res["all"].shape == (7,) and np.isfinite(res["all"]).all()
    errs = {o: [] for o in ["all"] + list(_MATERIAL_GREY)}
    for inputs in ev.test_loader:
        inputs = {k: v.cuda() for k, v in inputs.items()}
        pred = ev.predict(inputs).cpu().numpy()
        gt, mk = inputs["depth_gt"].cpu().numpy(), inputs[("mask", 0, 0)].cpu().numpy()
        for b in range(gt.shape[0]):
            for o in errs:
                m = (gt[b, 0] > 0.1) & (gt[b, 0] < 2.0)
                if o != "all":
                    m &= mk[b, 0] == _MATERIAL_GREY[o]
                if m.any():
                    errs[o].append(compute_depth_errors_numpy(gt[b, 0][m], pred[b, 0][m]))
    for o, e in errs.items():
        if e:
            np.testing.assert_allclose(res[o], np.array(e).mean(0), rtol=2e-5, atol=1e-6)


def test_separate_normals_decoder_variant_trains(tmp_path, monkeypatch):
    """`arch1++_separate_normals_dec` end to end through the Trainer façade (PD_NORMALS_DECODER=1): the extra model is built,
    trained (its parameters and the normals encoder's move), its loss is reported and drops, and it is checkpointed and
    reloaded like the other models."""
    import glob
    monkeypatch.setenv("PD_NORMALS_DECODER", "1")
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    tr = Trainer(_opts(tmp_path, ["--dropout_rate", "0.0", "--learning_rate", "1e-3"]))
    assert "normals_decoder" in tr.models
    tr.set_train()
    batch = synthetic.make_batch(2, 64, 96, frame_w=96, device="cuda", seed=3)
    batch.pop("depth_gt"); batch.pop(("mask", 0, 0))
    before = [p.detach().clone() for p in tr.models["normals_decoder"].parameters()]
    first = None
    for it in range(12):
        tr.model_optimizer.zero_grad()
        outputs, losses, _ = tr.process_batch(dict(batch), is_train=True)
        losses["loss"].backward()
        tr.model_optimizer.step()
        if first is None:
            first = float(losses["normals_decoder_loss"])
            assert outputs[("normals_pred", 0)].shape == (2, 3, 64, 96)
    last = float(losses["normals_decoder_loss"])
    assert 0.9 < first < 3.1 and last < first - 0.05, (first, last)
    after = list(tr.models["normals_decoder"].parameters())
    assert all((a - b).abs().max().item() > 0 for a, b in zip(after, before))
    tr.epoch = 0
    tr.save_model()
    folder = glob.glob(os.path.join(tr.log_path, "models", "weights_0"))[0]
    sd = torch.load(os.path.join(folder, "normals_decoder.pth"))
    assert list(sd) == list(tr.models["normals_decoder"].state_dict())
    for k, v in tr.models["normals_decoder"].state_dict().items():
        assert torch.equal(v.cpu(), sd[k])
