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


def test_hammer_dataset_reads_a_real_tree_and_falls_back_to_synthetic(tmp_path):
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
    synth = HAMMER_Dataset("does/not/exist", ["a"], 64, 96, [0], 4)
    s = synth[0]
    assert set(it.keys()) == set(s.keys())
    for k in it:
        assert it[k].dtype == s[k].dtype and it[k].shape == s[k].shape, k


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
