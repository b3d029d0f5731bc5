#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The outputs are data only (.npz of inputs + expected outputs);
no reference source text is stored.  Re-run with:

    python tests/golden/make_golden.py [group ...]      # groups: polar stokes nets loss options dpt unc

How the reference is imported:
  * polarisation.xolp, manydepth.layers, manydepth.normals_vec import directly;
  * manydepth/networks/{pre_encoders,depth_decoder}.py are loaded by file path
    under an empty stub parent package (their __init__ pulls torchvision, which
    is not installed here);
  * manydepth.trainer imports once tensorboard / kornia / roma / datasets /
    networks / dpt are stubbed; Trainer.compute_losses is then called unbound;
  * ppp_code/physical_normals_channels.py and polarisation/xolp_and_normals.py (stand-alone scripts guarded by
    __main__) import once `cv2` is an empty stub module and matplotlib.use("TkAgg") is a no-op: their functions
    are pure NumPy / SciPy.
Everything is seeded; all modules are run with pretrained=False / seeded init.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PD_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))  # repo root (oracle/)
sys.path.insert(0, OUT)


def _ref_imports():
    if REF not in sys.path:
        sys.path.insert(0, REF)


def _load_by_path(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_networks():
    """pre_encoders + depth_decoder of the reference, bypassing networks/__init__."""
    _ref_imports()
    import manydepth  # noqa: F401  (namespace package of the reference)
    if "manydepth.networks" not in sys.modules or not hasattr(sys.modules["manydepth.networks"], "_pd_stub"):
        stub = types.ModuleType("manydepth.networks")
        stub.__path__ = [os.path.join(REF, "manydepth", "networks")]
        stub._pd_stub = True
        sys.modules["manydepth.networks"] = stub
    pre = _load_by_path("manydepth.networks.pre_encoders", "manydepth/networks/pre_encoders.py")
    dec = _load_by_path("manydepth.networks.depth_decoder", "manydepth/networks/depth_decoder.py")
    return pre, dec


# ------------------------------------------------------------------ group: polar
def make_polar():
    _ref_imports()
    from polarisation.xolp import Iun_and_xolp
    import manydepth.normals_vec as nv
    pre, _ = ref_networks()
    angles = np.array([0, 45, 90, 135]) * np.pi / 180
    rng = np.random.default_rng(0)

    # G1 -- Iun_and_xolp: edge tuples, random uint8, physically generated low-DoLP
    edge = np.array([
        [0, 0, 0, 0], [255, 255, 255, 255], [255, 0, 0, 0], [0, 255, 0, 0], [0, 0, 255, 0],
        [0, 0, 0, 255], [10, 77, 200, 77], [200, 77, 10, 77], [90, 90, 90, 90], [1, 0, 0, 0],
        [0, 1, 0, 1], [255, 0, 255, 0], [0, 255, 0, 255], [128, 64, 128, 192], [3, 250, 7, 2],
        [100, 100, 101, 100]], dtype=np.uint8).reshape(4, 4, 4)
    rnd = rng.integers(0, 256, (32, 48, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:32, 0:48]
    iun = 120 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    rho_t = 0.02 + 0.25 * (0.5 + 0.5 * np.sin(xx / 5.0 + yy / 11.0)) ** 2
    phi_t = (np.pi / 2) * np.sin(xx / 13.0 - yy / 6.0)
    phys = np.stack([iun * (1 + rho_t * np.cos(2 * a - 2 * phi_t)) for a in angles], axis=2)
    phys = np.clip(np.rint(phys + rng.normal(0, 1.5, phys.shape)), 0, 255).astype(np.uint8)
    g1 = {}
    for name, img in (("edge", edge), ("rnd", rnd), ("phys", phys)):
        Iun, rho, phi = Iun_and_xolp(img.astype(np.float64), angles)   # uint8 -> float64 like np.stack of PIL 'L'
        g1[name + "_img"] = img
        g1[name + "_Iun"] = Iun
        g1[name + "_rho"] = rho
        g1[name + "_phi"] = phi
        # also what the dataset hands over: the reference passes uint8 arrays straight in
        Iun8, rho8, phi8 = Iun_and_xolp(img, angles)
        assert np.array_equal(rho8, rho) and np.array_equal(phi8, phi)
    np.savez_compressed(os.path.join(OUT, "g1_xolp.npz"), **g1)

    # G2 -- theta tables through the reference's own rho_diffuse / rho_spec
    sweep = np.concatenate([
        np.linspace(0, 2.2, 701), np.array([0.0, 1e-12, 0.3846, 0.38461538, 0.5, 0.999, 1.0, 1.0001, 2.0]),
        rng.random(300) * 0.45]).astype(np.float32)
    rt = torch.from_numpy(sweep).reshape(1, 1, -1)
    th_d = nv.rho_diffuse(rt, 1.5).numpy()
    th_s1, th_s2 = [t.numpy() for t in nv.rho_spec(rt, 1.5)]
    th_d13 = nv.rho_diffuse(rt, 1.3).numpy()
    np.savez_compressed(os.path.join(OUT, "g2_theta.npz"), rho=sweep, theta_d=th_d, theta_s1=th_s1,
                        theta_s2=th_s2, theta_d_n13=th_d13)

    # G3 -- get_normals on xolp from the phys + rnd images
    xs = []
    for img in (phys, rnd):
        _, rho, phi = Iun_and_xolp(img, angles)
        xs.append(np.stack((rho, phi), 0))
    xolp64 = torch.from_numpy(np.stack(xs, 0))               # fp64 [2,2,32,48] == ("xolp",0,0)
    normals = pre.ShallowNormalsEncoder.get_normals(xolp64.float()).float()
    xstd = pre.ShallowEncoder.normalizeInput(xolp64.float(), 'XOLP')
    np.savez_compressed(os.path.join(OUT, "g3_normals.npz"), xolp64=xolp64.numpy(), normals=normals.numpy(),
                        xolp_std=xstd.numpy())
    print("polar goldens written")


# ------------------------------------------------------------------ group: stokes (A2 + the numpy twins of A3-A5)
def _script_module(modname, relpath):
    """Import a stand-alone reference script whose top level needs cv2 / a Tk matplotlib backend."""
    import matplotlib
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    real_use = matplotlib.use
    matplotlib.use = lambda *a, **k: None
    try:
        return _load_by_path(modname, relpath)
    finally:
        matplotlib.use = real_use


def make_stokes():
    _ref_imports()
    ppp = _script_module("ref_physical_normals_channels", "ppp_code/physical_normals_channels.py")
    # xolp_and_normals imports `pol_split_and_save` from its own directory
    sys.path.insert(0, os.path.join(REF, "polarisation"))
    xn = _script_module("ref_xolp_and_normals", "polarisation/xolp_and_normals.py")
    rng = np.random.default_rng(7)
    angles = np.array([0, 45, 90, 135]) * np.pi / 180
    H, W = 24, 40
    rnd = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rnd[0, :6] = 0                                   # s0 = 0, s1 = s2 = 0  -> 0/0
    rnd[1, :6, 0] = 0; rnd[1, :6, 2] = 0             # s0 = 0, s2 != 0      -> x/0
    rnd[2, :6] = 255
    rnd[3, :6, 1] = rnd[3, :6, 3]                    # s2 = 0: branch cut rows
    yy, xx = np.mgrid[0:H, 0:W]
    iun = 110 + 50 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
    rho_t = 0.03 + 0.3 * (0.5 + 0.5 * np.sin(xx / 4.0 + yy / 9.0)) ** 2
    phi_t = (np.pi / 2) * np.sin(xx / 11.0 - yy / 5.0)
    phys = np.stack([iun * (1 + rho_t * np.cos(2 * a - 2 * phi_t)) for a in angles], axis=2)
    phys = np.clip(np.rint(phys + rng.normal(0, 1.5, phys.shape)), 0, 255).astype(np.uint8)
    g7 = {}
    for name, img in (("rnd", rnd), ("phys", phys)):
        mask = rng.random((H, W)) > 0.2
        mask[:4, :6] = True                          # keep the degenerate pixels inside the mask
        # the script masks the images first (physical_normals_channels.py:117-121), then calls the functions
        images = np.zeros((H, W, 4))
        for c in range(4):
            images[:, :, c][mask] = img[:, :, c][mask]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            rho2, phi2, Iun2 = ppp.PolarisationImage_channel(images, angles, mask)
            th_d = ppp.rho_diffuse_channel(rho2, 1.5)
            th_s1, th_s2 = ppp.rho_spec_channel(rho2, 1.5)
            N_d = ppp.calc_normals_channel(phi2, th_d, mask)
            N_s1 = ppp.calc_normals_channel(phi2 + np.pi / 2, th_s1, mask)
            N_s2 = ppp.calc_normals_channel(phi2 + np.pi / 2, th_s2, mask)
            # polarisation/xolp_and_normals.py twins (LS-fit XOLP + unmasked fp64 normals, :13-98)
            Iun_ls, rho_ls, phi_ls = xn.Iun_and_xolp(img.astype(np.float64), angles)
            t1, t2 = xn.rho_spec(rho_ls, 1.5)
            td = xn.rho_diffuse(rho_ls, 1.5)
            Nn_d = xn.calc_normals(phi_ls, td)
            Nn_s1 = xn.calc_normals(phi_ls + np.pi / 2, t1)
        g7.update({name + "_img": img, name + "_mask": mask, name + "_rho": rho2, name + "_phi": phi2,
                   name + "_Iun": Iun2, name + "_theta_d": th_d, name + "_theta_s1": th_s1, name + "_theta_s2": th_s2,
                   name + "_N_d": N_d, name + "_N_s1": N_s1, name + "_N_s2": N_s2,
                   name + "_ls_rho": rho_ls, name + "_ls_phi": phi_ls, name + "_ls_theta_d": td,
                   name + "_ls_theta_s1": t1, name + "_ls_theta_s2": t2, name + "_ls_N_d": Nn_d, name + "_ls_N_s1": Nn_s1})
    np.savez_compressed(os.path.join(OUT, "g7_stokes.npz"), **g7)
    print("stokes goldens written")


# ------------------------------------------------------------------ group: nets
def _grad_summary(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def make_nets():
    from synth_weights import fill_state_dict
    pre, dec = ref_networks()
    rng = torch.Generator().manual_seed(1)
    out = {}
    xolp = torch.stack([torch.rand(2, 64, 96, generator=rng) * 0.6,
                        (torch.rand(2, 64, 96, generator=rng) - 0.5) * 3.1], 1)          # [2,2,64,96] rho, phi
    out["xolp"] = xolp

    def run(name, mod, args, input_grad=True):
        """eval output, train(dropout=0) output, and the train-mode grads of a seeded scalar objective."""
        fill_state_dict(mod, 0, prefix=name + ".")
        mod.eval()
        with torch.no_grad():
            y = mod(*args)
        ys = y if isinstance(y, (list, tuple)) else [y]
        for i, t in enumerate(ys):
            out[f"{name}.eval.{i}"] = t.clone()
        mod.train()
        for p in mod.parameters():
            p.grad = None
        args_g = [a.clone().requires_grad_(input_grad) if a is not None else None for a in args]
        y = mod(*args_g)
        ys = y if isinstance(y, (list, tuple)) else [y]
        obj = 0
        for i, t in enumerate(ys):
            out[f"{name}.train.{i}"] = t.detach().clone()
            w = torch.randn(t.shape, generator=torch.Generator().manual_seed(100 + i))
            obj = obj + (t * w).sum()
        obj.backward()
        for k, g in _grad_summary(mod).items():
            # keep fixtures small: full tensor for small grads, strided sample + abs-sum for big ones
            if g.numel() <= 20000:
                out[f"{name}.grad.{k}"] = g
            else:
                out[f"{name}.gradsample.{k}"] = g.flatten()[::max(1, g.numel() // 4096)].clone()
                out[f"{name}.gradabssum.{k}"] = g.abs().sum()
        for i, a in enumerate(args_g):
            if a is not None and a.grad is not None:
                out[f"{name}.ingrad.{i}"] = a.grad.clone()
        for k, v in mod.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                if v.numel() <= 64:
                    out[f"{name}.buf.{k}"] = v.clone()

    run("xolp_encoder", pre.ShallowEncoder('XOLP', 2, 0.0), [xolp], input_grad=False)
    run("normals_encoder", pre.ShallowNormalsEncoder(9, 0.0), [xolp], input_grad=False)   # numpy bounce: no input grad
    rgbf = torch.randn(2, 128, 8, 12, generator=rng)
    xf = torch.randn(2, 64, 8, 12, generator=rng)
    nf = torch.randn(2, 64, 8, 12, generator=rng)
    out["joint.rgbf"], out["joint.xf"], out["joint.nf"] = rgbf, xf, nf
    run("joint3", pre.JointEncoder(0.0, True, True), [rgbf, xf, nf])
    run("joint_x", pre.JointEncoder(0.0, False, True), [rgbf, xf, None])
    run("joint_n", pre.JointEncoder(0.0, True, False), [rgbf, None, nf])
    run("joint_rgb", pre.JointEncoder(0.0, False, False), [rgbf, None, None])

    # DepthDecoder: list-of-features input
    import numpy as _np
    feats = [torch.randn(2, c, 64 // d, 96 // d, generator=rng) for c, d in
             ((64, 2), (64, 4), (128, 8), (256, 16), (512, 32))]
    for i, f in enumerate(feats):
        out[f"dec.feat.{i}"] = f
    dd = dec.DepthDecoder(_np.array([64, 64, 128, 256, 512]), range(4))
    fill_state_dict(dd, 0, prefix="mono_depth.")
    fg = [f.clone().requires_grad_(True) for f in feats]
    res = dd(fg)
    obj = 0
    for s in range(4):
        out[f"dec.disp.{s}"] = res[("disp", s)].detach().clone()
        obj = obj + (res[("disp", s)] * torch.randn(res[("disp", s)].shape, generator=torch.Generator().manual_seed(200 + s))).sum()
    obj.backward()
    for i, f in enumerate(fg):
        out[f"dec.featgrad.{i}"] = f.grad.clone()
    for k, g in _grad_summary(dd).items():
        if g.numel() <= 20000:
            out[f"dec.grad.{k}"] = g
        else:
            out[f"dec.gradsample.{k}"] = g.flatten()[::max(1, g.numel() // 4096)].clone()
            out[f"dec.gradabssum.{k}"] = g.abs().sum()
    np.savez_compressed(os.path.join(OUT, "g4_nets.npz"), **{k: v.numpy() for k, v in out.items()})
    print("nets goldens written:", len(out), "arrays")


# ------------------------------------------------------------------ group: loss
def _import_ref_trainer():
    """manydepth.trainer under harmless stubs."""
    _ref_imports()
    from oracle import losses as olosses
    tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    for name in ("kornia", "kornia.geometry", "kornia.geometry.depth", "roma", "manydepth.datasets", "manydepth.dpt"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    # kornia is not installed: the normals term is served by the build's own restatement
    # (self-consistency fixture); the pure-reference fixture uses normals_loss_weight = 0.
    sys.modules["kornia.geometry.depth"].depth_to_normals = olosses.depth_to_normals
    if "manydepth.networks" not in sys.modules:
        sys.modules["manydepth.networks"] = types.ModuleType("manydepth.networks")
    import manydepth.trainer as T
    return T


def synth_batch(B=2, H=64, W=96, seed=3):
    g = torch.Generator().manual_seed(seed)
    inputs = {}
    for s in range(4):
        inputs[("color", 0, s)] = torch.rand(B, 3, H >> s, W >> s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 1.0 + 0.6 * torch.sin(xx / 17.0)[None] * torch.cos(yy / 11.0)[None] + 0.1 * torch.rand(B, H, W, generator=g)
    depth[torch.rand(B, H, W, generator=g) < 0.1] = 0.0          # invalid pixels
    depth[:, :, -5:] = 0.0
    depth[0, 3, 3] = 2.5                                           # beyond max_depth
    inputs["depth"] = depth[:, None].contiguous()
    K = torch.eye(4)[None].repeat(B, 1, 1)
    K[:, 0, 0] = 0.65 * W; K[:, 1, 1] = 0.65 * W; K[:, 0, 2] = W / 2; K[:, 1, 2] = H / 2
    inputs[("K", 0)] = K
    disps = {("disp", s): torch.sigmoid(torch.randn(B, 1, H >> s, W >> s, generator=g)) for s in range(4)}
    return inputs, disps


def make_loss():
    T = _import_ref_trainer()
    import manydepth.layers as L
    import torch.nn.functional as F
    out = {}
    inputs, disps = synth_batch()
    for k, v in inputs.items():
        out["in." + "_".join(map(str, k)) if isinstance(k, tuple) else "in." + k] = v
    for (_, s), v in disps.items():
        out[f"disp.{s}"] = v
    for tag, lam in (("lam0", 0.0), ("lam035", 0.35)):
        opt = types.SimpleNamespace(scales=[0, 1, 2, 3], v1_multiscale=False, train_dpt=False, depth_supervision=True,
                                    depth_supervision_only=True, min_depth=0.1, max_depth=2.0,
                                    normals_loss_weight=lam, disparity_smoothness=1e-3, supervise_pose=False,
                                    height=64, width=96)
        fake = types.SimpleNamespace(opt=opt, num_scales=4, device=torch.device("cpu"))
        fake.compute_supervised_normals_losses = types.MethodType(T.Trainer.compute_supervised_normals_losses, fake)
        dl = {k: v.clone().requires_grad_(True) for k, v in disps.items()}
        outputs = dict(dl)
        for s in range(4):       # trainer.py:531-545
            up = F.interpolate(dl[("disp", s)], [64, 96], mode="bilinear", align_corners=False)
            outputs[("depth", 0, s)] = L.disp_to_depth(up, 0.1, 2.0)[1]
        losses = T.Trainer.compute_losses(fake, {k: v for k, v in inputs.items()}, outputs, is_multi=False)
        losses["loss"].backward()
        for k, v in losses.items():
            out[f"{tag}.{k}"] = torch.as_tensor(v).detach()
        for s in range(4):
            out[f"{tag}.ddisp.{s}"] = dl[("disp", s)].grad.clone()
            out[f"{tag}.depth.{s}"] = outputs[("depth", 0, s)].detach().clone()
    # G6: small layer functions straight from manydepth.layers
    g = torch.Generator().manual_seed(11)
    a, b = torch.rand(2, 3, 24, 32, generator=g), torch.rand(2, 3, 24, 32, generator=g)
    out["ssim.x"], out["ssim.y"], out["ssim.out"] = a, b, L.SSIM()(a, b)
    d = torch.rand(2, 1, 24, 32, generator=g)
    out["smooth.disp"], out["smooth.out"] = d, L.get_smooth_loss(d, a)
    sd, dp = L.disp_to_depth(d, 0.1, 2.0)
    out["d2d.scaled"], out["d2d.depth"] = sd, dp
    gt, pr = 0.2 + 1.5 * torch.rand(500, generator=g), 0.2 + 1.5 * torch.rand(500, generator=g)
    out["err.gt"], out["err.pred"] = gt, pr
    out["err.out"] = torch.stack(L.compute_depth_errors(gt, pr))
    np.savez_compressed(os.path.join(OUT, "g5_loss.npz"), **{k: v.numpy() for k, v in out.items()})
    print("loss goldens written:", len(out), "arrays")


# ------------------------------------------------------------------ group: options
SCRIPT_FLAGS = ["--png", "--num_depth_bins", "96", "--num_workers", "8", "--batch_size", "12",
                "--data_path", "/data/HAMMER/train/", "--data_path_val", "/data/HAMMER/test_unseen/",
                "--log_dir", "/tmp/exp", "--num_epochs", "50", "--scheduler_step_size", "15",
                "--freeze_teacher_epoch", "50", "--learning_rate", "1e-4", "--width", "480", "--height", "320",
                "--dataset", "HAMMER", "--split", "HAMMER", "--eval_split", "HAMMER_unseen", "--min_depth", "0.1",
                "--max_depth", "2.0", "--disparity_smoothness", "1e-3", "--offset", "10",
                "--no_matching_augmentation", "--depth_supervision_only", "True", "--depth_supervision", "True",
                "--modality", "polarization", "--depth_modality", "_gt", "--normals_loss_weight", "0.35",
                "--model_name", "ABLATIONS_rgb_xolp_normals", "--augment_xolp", "--augment_normals"]


def make_options():
    """Defaults and the train_supervised_GT.sh command line through the reference's own argparse."""
    import json
    _ref_imports()
    spec = importlib.util.spec_from_file_location("ref_options", os.path.join(REF, "manydepth", "options.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    try:
        sys.argv = ["x"]
        defaults = vars(mod.MonodepthOptions().parse())
        sys.argv = ["x"] + SCRIPT_FLAGS
        script = vars(mod.MonodepthOptions().parse())
        sys.argv = ["x", "--depth_supervision_only", "False"]
        quirk = vars(mod.MonodepthOptions().parse())["depth_supervision_only"]
    finally:
        sys.argv = old
    with open(os.path.join(OUT, "g6_options.json"), "w") as f:
        json.dump({"defaults": defaults, "script_flags": SCRIPT_FLAGS, "script": script,
                   "bool_quirk_False_string": quirk}, f, indent=1, sort_keys=True)
    print("options golden written:", len(defaults), "flags")


# ------------------------------------------------------------------ group: dpt (fusion block of the DPT decoder)
def make_dpt():
    """G8 -- manydepth/dpt/blocks.py: FeatureFusionBlock_custom / ResidualConvUnit_custom / Interpolate exactly as
    dpt/models.py:15-23 builds them (activation nn.ReLU(False), bn = False, expand = False, align_corners = True): outputs,
    input gradients and parameter gradients of the two-input form (refinenet1..3) and the one-input form (refinenet4), and
    of the output head's Interpolate.  blocks.py is loaded by file path; its `.vit` sibling (timm ViT factories, absent
    here) is an empty stub -- the fusion blocks do not touch it."""
    from synth_weights import fill_state_dict
    _ref_imports()
    import manydepth  # noqa: F401
    pkg = types.ModuleType("manydepth.dpt")
    pkg.__path__ = [os.path.join(REF, "manydepth", "dpt")]
    sys.modules["manydepth.dpt"] = pkg
    vit = types.ModuleType("manydepth.dpt.vit")
    for name in ("_make_pretrained_vitb_rn50_384", "_make_pretrained_vitl16_384", "_make_pretrained_vitb16_384", "forward_vit"):
        setattr(vit, name, None)
    sys.modules["manydepth.dpt.vit"] = vit
    blocks = _load_by_path("manydepth.dpt.blocks", "manydepth/dpt/blocks.py")
    import torch.nn as nn
    out = {}
    g = torch.Generator().manual_seed(8)
    F_, N, H, W = 64, 2, 12, 16
    for tag, n_in in (("two", 2), ("one", 1)):
        blk = blocks.FeatureFusionBlock_custom(F_, nn.ReLU(False), deconv=False, bn=False, expand=False, align_corners=True)
        fill_state_dict(blk, 0, prefix="fusion.")
        xs = [torch.randn(N, F_, H, W, generator=g).requires_grad_(True) for _ in range(n_in)]
        y = blk(*xs)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out[f"{tag}.out"] = y.detach().numpy(); out[f"{tag}.gout"] = gy.numpy()
        for i, x in enumerate(xs):
            out[f"{tag}.x{i}"] = x.detach().numpy(); out[f"{tag}.gx{i}"] = x.grad.numpy()
        for k, v in _grad_summary(blk).items():
            out[f"{tag}.grad.{k}"] = v.numpy()
    rcu = blocks.ResidualConvUnit_custom(F_, nn.ReLU(False), False)
    fill_state_dict(rcu, 0, prefix="fusion.resConfUnit2.")
    x = torch.randn(N, F_, H, W, generator=g).requires_grad_(True)
    y = rcu(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update({"rcu.x": x.detach().numpy(), "rcu.out": y.detach().numpy(), "rcu.gout": gy.numpy(), "rcu.gx": x.grad.numpy()})
    ip = blocks.Interpolate(scale_factor=2, mode="bilinear", align_corners=True)
    x = torch.randn(N, 32, 7, 9, generator=g).requires_grad_(True)
    y = ip(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update({"interp.x": x.detach().numpy(), "interp.out": y.detach().numpy(), "interp.gout": gy.numpy(), "interp.gx": x.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "g8_dpt_fusion.npz"), **out)
    print("dpt golden written:", len(out), "arrays")


# ------------------------------------------------------------------ group: unc (DepthDecoder with uncertainty heads)
def make_unc():
    """G9 -- depth_decoder.py:46-50,71-73: DepthDecoder(uncertainty=True) on the feature maps of G4 (dec.feat.*): the twelve
    output maps, the gradients of the input features and of the parameters (heads in full, the rest sampled as in G4)."""
    from synth_weights import fill_state_dict
    _, dec = ref_networks()
    g4 = np.load(os.path.join(OUT, "g4_nets.npz"))
    feats = [torch.from_numpy(g4[f"dec.feat.{i}"]) for i in range(5)]
    dd = dec.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4), uncertainty=True)
    fill_state_dict(dd, 0, prefix="mono_depth.")
    fg = [f.clone().requires_grad_(True) for f in feats]
    res = dd(fg)
    out, obj = {}, 0
    for j, kind in enumerate(("disp", "uncertainty", "uncertainty_color")):
        for s in range(4):
            t = res[(kind, s)]
            out[f"{kind}.{s}"] = t.detach().clone()
            obj = obj + (t * torch.randn(t.shape, generator=torch.Generator().manual_seed(300 + 10 * j + s))).sum()
    obj.backward()
    for i, f in enumerate(fg):
        out[f"featgrad.{i}"] = f.grad.clone()
    for k, g in _grad_summary(dd).items():
        if g.numel() <= 20000:
            out[f"grad.{k}"] = g
        else:
            out[f"gradsample.{k}"] = g.flatten()[::max(1, g.numel() // 4096)].clone()
            out[f"gradabssum.{k}"] = g.abs().sum()
    out["keys"] = np.array(list(dd.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "g9_decoder_uncertainty.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("uncertainty-decoder golden written:", len(out), "arrays")


GROUPS = {"unc": make_unc, "polar": make_polar, "stokes": make_stokes, "nets": make_nets, "loss": make_loss, "options": make_options,
          "dpt": make_dpt}

if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
