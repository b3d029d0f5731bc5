#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The outputs are data only (.npz of inputs + expected outputs);
no reference source text is stored.  Re-run with:

    python tests/golden/make_golden.py [group ...]      # groups: polar nets loss

How the reference is imported:
  * polarisation.xolp, manydepth.layers, manydepth.normals_vec import directly;
  * manydepth/networks/{pre_encoders,depth_decoder}.py are loaded by file path
    under an empty stub parent package (their __init__ pulls torchvision, which
    is not installed here);
  * manydepth.trainer imports once tensorboard / kornia / roma / datasets /
    networks / dpt are stubbed; Trainer.compute_losses is then called unbound.
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


GROUPS = {"polar": make_polar}

if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
