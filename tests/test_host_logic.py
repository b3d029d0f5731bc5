"""Host-side logic that needs no GPU: CLI contract vs the reference's own argparse (fixture),
parameter store layout, checkpoint key names, façade import surface."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_options_match_reference_argparse():
    from manydepth.options import MonodepthOptions
    g = json.load(open(os.path.join(GOLDEN, "g6_options.json")))
    mine = vars(MonodepthOptions().parse([]))
    assert mine == g["defaults"]
    script = vars(MonodepthOptions().parse(g["script_flags"]))
    assert script == g["script"]
    # the reference's type=bool quirk: any non-empty string is True (options.py:33-86)
    assert MonodepthOptions().parse(["--depth_supervision_only", "False"]).depth_supervision_only is True
    assert g["bool_quirk_False_string"] is True


def test_facade_import_surface():
    import manydepth.networks as nw
    for name in ("ShallowResnetEncoder", "ShallowEncoder", "ShallowNormalsEncoder", "JointEncoder", "DepthDecoder",
                 "ResnetEncoder", "ResnetEncoderMatching", "PoseDecoder", "PoseCNN"):
        assert hasattr(nw, name)
    with pytest.raises(NotImplementedError):
        nw.PoseCNN(2)
    import manydepth.layers as L
    for name in ("disp_to_depth", "ConvBlock", "Conv3x3", "upsample", "get_smooth_loss", "SSIM",
                 "compute_depth_errors", "compute_depth_errors_numpy"):
        assert hasattr(L, name)
    import manydepth.normals_vec as nv
    assert all(hasattr(nv, n) for n in ("rho_diffuse", "rho_spec", "calc_normals"))
    from polarisation.xolp import Iun_and_xolp  # noqa: F401
    from manydepth.trainer import Trainer
    for m in ("train", "run_epoch", "process_batch", "compute_losses", "compute_supervised_normals_losses", "val",
              "test", "set_train", "set_eval", "save_opts", "save_model", "load_model", "load_mono_model", "log",
              "log_time"):
        assert callable(getattr(Trainer, m))


def test_state_dict_keys_match_oracle_modules():
    """Checkpoint key compatibility: façade modules vs the oracle restatement (itself pinned on the
    reference's modules by tests/test_oracle_nets.py via load_state_dict of shared synthetic weights)."""
    from manydepth import networks
    from oracle import nets as onets
    pairs = [(networks.ShallowEncoder('XOLP', 2, 0.1), onets.ShallowEncoder('XOLP', 2, 0.1)),
             (networks.ShallowNormalsEncoder(9, 0.1), onets.ShallowNormalsEncoder(9, 0.1)),
             (networks.JointEncoder(0.1, True, True), onets.JointEncoder(0.1, True, True)),
             (networks.ShallowResnetEncoder(18, False), onets.ShallowResnetEncoder(18, False)),
             (networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4)),
              onets.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4)))]
    for a, b in pairs:
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)


def test_param_store_views_and_order():
    from manydepth import networks
    from polardepth.engine import ParamStore
    models = {"rgb_encoder": networks.ShallowResnetEncoder(18, False),
              "mono_depth": networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4))}
    ref_w = models["mono_depth"].decoder[0].conv.conv.weight.detach().clone()
    unused = lambda m, p: m == "rgb_encoder" and p.split(".")[1] in ("layer3", "layer4", "fc")
    st = ParamStore(models, order=["rgb_encoder", "mono_depth"], unused=unused, device=torch.device("cpu"))
    w = models["mono_depth"].decoder[0].conv.conv.weight
    assert torch.equal(w.detach(), ref_w) and w.is_contiguous(memory_format=torch.channels_last)
    assert w.grad is not None and w.grad.stride() == w.stride()
    assert w.data.untyped_storage().data_ptr() == st.flat.untyped_storage().data_ptr()
    # backward-completion order: the decoder's parameters come first, unused resnet tail last
    first = st.entries[0][0]
    assert first.startswith("mono_depth.")
    assert all(n.split(".")[2] in ("layer3", "layer4", "fc") for n, _ in st.entries[st.n_used_params:])
    used_numel = sum(p.numel() for _, p in st.entries[:st.n_used_params])
    assert used_numel <= st.n_used <= used_numel + 4 * st.n_used_params
    # writing through the flat buffer is visible in the module, load_state_dict keeps the views
    st.flat.zero_()
    assert w.abs().sum().item() == 0
    models["mono_depth"].load_state_dict({k: torch.ones_like(v) for k, v in models["mono_depth"].state_dict().items()})
    off, n = st.offsets["mono_depth.decoder.0.conv.conv.weight"]
    assert st.flat[off:off + n].sum().item() == n
    w.grad.fill_(2.0)
    assert st.grad[off:off + n].sum().item() == 2 * n
    st.zero_grad()
    assert w.grad.abs().sum().item() == 0


def test_no_cpu_fallback_anywhere():
    from manydepth.options import MonodepthOptions
    from manydepth.trainer import Trainer
    from polardepth import ops
    opts = MonodepthOptions().parse(["--no_cuda", "--depth_supervision_only", "True", "--depth_supervision", "True"])
    with pytest.raises(RuntimeError, match="no CPU path"):
        Trainer(opts)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 3, 3))


def test_hammer_dataset_reads_a_real_tree_and_is_synthetic_only_on_request(tmp_path):
    """File-backed HAMMER items (PIL) have the keys / dtypes / shapes of indoor_dataset.py:277-425."""
    from PIL import Image
    from manydepth.datasets import HAMMER_Dataset
    rng = np.random.default_rng(0)
    scene = tmp_path / "scene1_traj1_1" / "polarization"
    for d in ("rgb", "pol00", "pol01", "pol10", "pol11", "_gt", "_instance"):
        (scene / d).mkdir(parents=True)
    for idx in (3, 4):
        Image.fromarray(rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(scene / "rgb" / f"{idx:06d}.png")
        for d in ("pol00", "pol01", "pol10", "pol11"):
            Image.fromarray(rng.integers(0, 256, (96, 128), dtype=np.uint8)).save(scene / d / f"{idx:06d}.png")
        Image.fromarray(rng.integers(300, 1800, (96, 128)).astype(np.uint16)).save(scene / "_gt" / f"{idx:06d}.png")
        Image.fromarray((rng.integers(0, 11, (96, 128)) * 20).astype(np.uint8)).save(scene / "_instance" / f"{idx:06d}.png")
    (scene / "intrinsics.txt").write_text("80 0 64\n0 82 48\n0 0 1\n")
    ds = HAMMER_Dataset(str(tmp_path), ["scene1_traj1_1"], 64, 96, [0], 4, is_train=True)
    assert len(ds) == 2
    it = ds[0]
    assert it[("pol", 0, 0)].dtype == torch.uint8 and it[("pol", 0, 0)].shape == (4, 64, 96)
    assert it[("color", 0, 2)].shape == (3, 16, 24) and 0 <= it[("color", 0, 0)].min() and it[("color", 0, 0)].max() <= 1
    assert it["depth"].shape == (1, 64, 96) and 0.3 <= it["depth"].min() and it["depth"].max() < 1.8
    assert it[("mask", 0, 0)].dtype == torch.int32 and set(np.unique(it[("mask", 0, 0)].numpy())) <= set(range(0, 201, 20))
    K = it[("K", 0)]
    assert abs(K[0, 0].item() - 80 / 128 * 96) < 1e-4 and abs(K[1, 2].item() - 48 / 96 * 64) < 1e-4
    assert torch.allclose(it[("K", 1)][0, 0], K[0, 0] / 2)
    raw = HAMMER_Dataset(str(tmp_path), ["scene1_traj1_1"], 64, 96, [0], 4, is_train=True, raw_pol=True)[0]
    assert raw[("pol", 0, 0)].shape == (4, 96, 128) and raw[("color", 0, 0)].shape == (3, 64, 96)   # planes stay native
    # synthetic items only for the literal "synthetic"; a wrong path or an empty scene fails loudly like the reference
    synth = HAMMER_Dataset("synthetic", ["a"], 64, 96, [0], 4)
    s = synth[0]
    assert set(it.keys()) == set(s.keys())
    for k in it:
        assert it[k].dtype == s[k].dtype and it[k].shape == s[k].shape, k
    with pytest.raises(FileNotFoundError, match="not a directory"):
        HAMMER_Dataset("does/not/exist", ["a"], 64, 96, [0], 4)
    with pytest.raises(FileNotFoundError, match="no complete HAMMER frame"):
        HAMMER_Dataset(str(tmp_path), ["scene_that_is_missing"], 64, 96, [0], 4)


def test_color_augmentation_is_torchvision_color_jitter_on_pil(tmp_path):
    """indoor_dataset.py:92-106, 226-233, 300, 404-407: p = 0.5, factors in [0.8, 1.2] / hue [-0.1, 0.1], random order,
    the same draw for every scale of an item; evaluation items are never augmented."""
    import random
    from PIL import Image, ImageEnhance
    from manydepth import datasets
    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8))
    random.seed(5)
    params = datasets.color_jitter_params()
    assert sorted(n for n, _ in params) == ["brightness", "contrast", "hue", "saturation"]
    f = dict(params)
    assert 0.8 <= f["brightness"] <= 1.2 and 0.8 <= f["contrast"] <= 1.2 and 0.8 <= f["saturation"] <= 1.2 and abs(f["hue"]) <= 0.1
    assert len({tuple(n for n, _ in datasets.color_jitter_params()) for _ in range(40)}) > 4     # the order is shuffled
    # each property alone equals the PIL operation torchvision 0.8.2 calls
    np.testing.assert_array_equal(np.asarray(datasets.apply_color_jitter(img, [("brightness", 1.1)])),
                                  np.asarray(ImageEnhance.Brightness(img).enhance(1.1)))
    np.testing.assert_array_equal(np.asarray(datasets.apply_color_jitter(img, [("saturation", 0.9)])),
                                  np.asarray(ImageEnhance.Color(img).enhance(0.9)))
    hsv = np.asarray(img.convert("HSV")).copy()
    hsv[..., 0] = hsv[..., 0] + np.uint8(int(0.05 * 255))
    np.testing.assert_array_equal(np.asarray(datasets.apply_color_jitter(img, [("hue", 0.05)])),
                                  np.asarray(Image.fromarray(hsv, "HSV").convert("RGB")))
    assert np.asarray(datasets.apply_color_jitter(img, [("hue", 0.0)])).shape == (24, 32, 3)
    # dataset level
    scene = tmp_path / "s" / "polarization"
    for d in ("rgb", "pol00", "pol01", "pol10", "pol11", "_gt"):
        (scene / d).mkdir(parents=True)
    Image.fromarray(rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(scene / "rgb" / "000001.png")
    for d in ("pol00", "pol01", "pol10", "pol11"):
        Image.fromarray(rng.integers(0, 256, (96, 128), dtype=np.uint8)).save(scene / d / "000001.png")
    Image.fromarray(rng.integers(300, 1800, (96, 128)).astype(np.uint16)).save(scene / "_gt" / "000001.png")
    (scene / "intrinsics.txt").write_text("80 0 64\n0 82 48\n0 0 1\n")
    train = datasets.HAMMER_Dataset(str(tmp_path), ["s"], 64, 96, [0], 4, is_train=True)
    test = datasets.HAMMER_Dataset(str(tmp_path), ["s"], 64, 96, [0], 4, is_train=False)
    random.seed(0)
    n_aug = 0
    for _ in range(24):
        it = train[0]
        aug = not torch.equal(it[("color_aug", 0, 0)], it[("color", 0, 0)])
        n_aug += aug
        for s_ in range(1, 4):      # all scales of one item are augmented together
            assert (not torch.equal(it[("color_aug", 0, s_)], it[("color", 0, s_)])) == aug
    assert 5 <= n_aug <= 19
    it = test[0]
    assert all(torch.equal(it[("color_aug", 0, s_)], it[("color", 0, s_)]) for s_ in range(4))


def test_adam_checkpoint_interop_with_torch_optim_adam():
    """adam.pth of the reference = torch.optim.Adam(parameters_to_train).state_dict() with parameters numbered
    normals, xolp, joint, rgb, mono in model.parameters() order (trainer.py:194-219).  A state written by a real
    torch.optim.Adam over the oracle modules loads into FusedAdam's flat buffers parameter by parameter, FusedAdam's
    own file loads into torch.optim.Adam, and a wrong order is rejected before anything is copied."""
    from manydepth import networks
    from oracle import nets as onets
    from polardepth.engine import ParamStore, FusedAdam
    order = ["rgb_encoder", "xolp_encoder", "normals_encoder", "joint_encoder", "mono_depth"]
    ref_order = ["normals_encoder", "xolp_encoder", "joint_encoder", "rgb_encoder", "mono_depth"]
    mine = {"rgb_encoder": networks.ShallowResnetEncoder(18, False), "xolp_encoder": networks.ShallowEncoder('XOLP', 2, 0.1),
            "normals_encoder": networks.ShallowNormalsEncoder(9, 0.1), "joint_encoder": networks.JointEncoder(0.1, True, True),
            "mono_depth": networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4))}
    ref = onets.build_models(True, True, 0.1)
    unused = lambda m, p: m == "rgb_encoder" and p.split(".")[1] in ("layer3", "layer4", "fc")
    store = ParamStore(mine, order=order, unused=unused, device=torch.device("cpu"))
    adam = FusedAdam(store, 1e-4)
    # the reference side: one real Adam step with synthetic gradients on every parameter that is trained
    params, names = [], []
    for m in ref_order:
        for pn, p in ref[m].named_parameters():
            params.append(p); names.append(f"{m}.{pn}")
    assert adam.reference_order() == names
    topt = torch.optim.Adam(params, 1e-4)
    g = torch.Generator().manual_seed(0)
    for n, p in zip(names, params):
        if not unused(*n.split(".", 1)):
            p.grad = torch.randn(p.shape, generator=g)
    topt.step()
    sd = topt.state_dict()
    assert "pd_order" not in sd
    adam.load_state_dict(sd)
    assert adam.step_count == 1
    for i, n in enumerate(names):
        off, cnt = store.offsets[n]
        p = dict(store.entries)[n]
        if i in sd["state"]:
            assert torch.equal(store._view(adam.exp_avg, p, off), sd["state"][i]["exp_avg"]), n
            assert torch.equal(store._view(adam.exp_avg_sq, p, off), sd["state"][i]["exp_avg_sq"]), n
        else:
            assert adam.exp_avg[off:off + cnt].abs().sum().item() == 0
    # and back: torch.optim.Adam accepts FusedAdam's file and holds the same moments at the same indices
    out = adam.state_dict()
    assert out["param_groups"][0]["params"] == list(range(len(names))) and out["pd_order"] == names
    topt2 = torch.optim.Adam(params, 1e-4)
    topt2.load_state_dict({"state": out["state"], "param_groups": out["param_groups"]})
    for i in sd["state"]:
        assert torch.equal(topt2.state[params[i]]["exp_avg"], sd["state"][i]["exp_avg"])
    # a state whose indices follow another order (here: the flat buffer's own) is refused, nothing is half-loaded
    wrong = {"state": {i: sd["state"][j] for i, j in zip(sorted(sd["state"]), reversed(sorted(sd["state"])))},
             "param_groups": sd["param_groups"]}
    before = adam.exp_avg.clone()
    with pytest.raises(ValueError, match="shape"):
        adam.load_state_dict(wrong)
    assert torch.equal(adam.exp_avg, before)


def test_resume_state_distinguishes_mid_epoch_from_end_of_epoch_checkpoints():
    """run_epoch saves on logging steps before `self.step += 1` and before lr_scheduler.step(); train() saves after the
    epoch.  Loading resumes in the same epoch behind the consumed batches, or at the next epoch -- and the restored
    StepLR is consistent with the epoch counter in both cases."""
    import types
    from manydepth.trainer import Trainer
    from polardepth import functional as PF

    def fake_trainer(tmp, epoch, step):
        t = types.SimpleNamespace()
        t.rank, t.distributed, t.epoch, t.step, t.log_path = 0, False, epoch, step, str(tmp)
        t.models = {"m": torch.nn.Linear(2, 2)}
        opt = torch.optim.SGD(t.models["m"].parameters(), lr=1.0)
        t.model_optimizer = opt
        t.model_lr_scheduler = torch.optim.lr_scheduler.StepLR(opt, 2, 0.1)
        t.opt = types.SimpleNamespace(load_weights_folder=None, models_to_load=["m"], resume_state=True)
        t.device = "cpu"
        t._load_into = lambda n, path, strict=False: None
        return t
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        # mid-epoch save in epoch 3 (three full epochs done -> scheduler stepped 3 times), batch 7, step 1234
        a = fake_trainer(tmp, 3, 1234)
        for _ in range(3):
            a.model_optimizer.step(); a.model_lr_scheduler.step()
        Trainer.save_model(a, epoch_complete=False, batch_idx=7)
        b = fake_trainer(tmp, 0, 0)
        b.opt.load_weights_folder = os.path.join(tmp, "models", "weights_3")
        Trainer.load_model(b)
        assert (b.resume_epoch, b.resume_step, b.resume_batch) == (3, 1235, 8)
        assert b.model_lr_scheduler.last_epoch == b.resume_epoch == 3
        # end-of-epoch save of the same epoch: scheduler stepped once more
        a.model_optimizer.step(); a.model_lr_scheduler.step()
        a.step = 1300
        Trainer.save_model(a, epoch_complete=True)
        c = fake_trainer(tmp, 0, 0)
        c.opt.load_weights_folder = os.path.join(tmp, "models", "weights_3")
        Trainer.load_model(c)
        assert (c.resume_epoch, c.resume_step, c.resume_batch) == (4, 1300, 0)
        assert c.model_lr_scheduler.last_epoch == c.resume_epoch == 4
        assert not [f for f in os.listdir(b.opt.load_weights_folder) if f.endswith(".tmp")]    # atomic writes leave no temp files


def test_lanczos_coefficient_tables_reproduce_pillow_on_the_host():
    """polardepth.resize.lanczos_coeffs (Pillow's precompute_coeffs / normalize_coeffs_8bpc) applied with NumPy
    integer arithmetic equals PIL's LANCZOS resize -- pins the tables the device kernels consume."""
    from PIL import Image
    from polardepth.resize import lanczos_coeffs, PRECISION_BITS
    rng = np.random.default_rng(2)
    for Hs, Ws, Hd, Wd in ((104, 136, 64, 76), (40, 50, 64, 96)):
        img = rng.integers(0, 256, (Hs, Ws), dtype=np.uint8)
        cur = img
        for axis, (n_in, n_out) in ((1, (Ws, Wd)), (0, (Hs, Hd))):
            kk, b = lanczos_coeffs(n_in, n_out)
            src = cur if axis == 1 else cur.T
            out = np.zeros((src.shape[0], n_out), np.uint8)
            for xx in range(n_out):
                x0, n = b[xx]
                acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n].astype(np.int64) * kk[xx, :n]).sum(1)
                out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
            cur = out if axis == 1 else out.T
        ref = np.asarray(Image.fromarray(img, "L").resize((Wd, Hd), Image.LANCZOS))
        np.testing.assert_array_equal(cur, ref)


def test_torchvision_keyed_resnet18_weights_load_through_env(tmp_path, monkeypatch):
    """--weights_init pretrained (options.py:261-265) without network: a local torchvision resnet18 state_dict
    ($PD_RESNET18_WEIGHTS; keys conv1.weight, bn1.*, layerN.M.*, fc.*) lands in ShallowResnetEncoder.encoder."""
    from manydepth import networks
    from oracle import nets as onets
    tv = onets.ShallowResnetEncoder(18, False).encoder.state_dict()           # torchvision key names
    assert "layer2.0.downsample.0.weight" in tv and "fc.bias" in tv and "bn1.num_batches_tracked" in tv
    tv = {k: torch.randn(v.shape) if v.dtype.is_floating_point else v for k, v in tv.items()}
    path = tmp_path / "resnet18-5c106cde.pth"
    torch.save(tv, path)
    monkeypatch.setenv("PD_RESNET18_WEIGHTS", str(path))
    enc = networks.ShallowResnetEncoder(18, True)
    got = enc.encoder.state_dict()
    assert list(got.keys()) == list(tv.keys())
    for k in tv:
        assert torch.equal(got[k], tv[k]), k
    monkeypatch.delenv("PD_RESNET18_WEIGHTS")
    with pytest.warns(UserWarning, match="from scratch"):
        networks.ShallowResnetEncoder(18, True)


def test_quadrant_mosaic_split_matches_the_reference_splitter_layout():
    """pol_split_and_save.py:16-25: pol00 = top-left, pol10 = bottom-left, pol01 = top-right, pol11 = bottom-right;
    K1 wants 0/45/90/135 degrees = pol00, pol01, pol10, pol11 (indoor_dataset.py:435-438)."""
    import torch
    from polardepth import polar as pdpolar
    h, w = 6, 8
    m = torch.arange(2 * 2 * h * 2 * w, dtype=torch.int32).reshape(2, 2 * h, 2 * w).to(torch.uint8)
    planes = pdpolar.split_mosaic(m)
    assert planes.shape == (2, 4, h, w) and planes.is_contiguous()
    assert torch.equal(planes[:, 0], m[:, :h, :w])      # pol00, 0 deg
    assert torch.equal(planes[:, 1], m[:, :h, w:])      # pol01, 45 deg
    assert torch.equal(planes[:, 2], m[:, h:, :w])      # pol10, 90 deg
    assert torch.equal(planes[:, 3], m[:, h:, w:])      # pol11, 135 deg
    import pytest
    with pytest.raises(ValueError):
        pdpolar.split_mosaic(torch.zeros(1, 5, 8, dtype=torch.uint8))
