"""Kernel-only timing of K1 (pd_polar_fwd) with HIP events: GB/s vs batch and output variant.

Every launch works on a different buffer set (``--sets`` rotating copies, sized so that their total exceeds the
256 MB Infinity Cache): the figure is an HBM figure, as in the training step, where 79 ms of other traffic
separate two K1 launches.  ``nt_loads`` (None | True | False) is pd_polar_fwd's PD_POLAR_NT_LOADS / PD_POLAR_PLAIN_LOADS flag.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import polar as pdpolar  # noqa: E402


def make_planes(B, H, W, realistic, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if realistic:
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
        iun = 120 + 60 * torch.sin(xx / 40.0 + seed) * torch.cos(yy / 30.0)
        rho = 0.02 + 0.25 * (0.5 + 0.5 * torch.sin(xx / 25.0 + yy / 50.0 + seed)) ** 2
        phi = 1.57 * torch.sin(xx / 60.0 - yy / 35.0)
        planes = [iun * (1 + rho * torch.cos(2 * a - 2 * phi)) for a in (0.0, 0.7853981, 1.5707963, 2.3561945)]
        pol = torch.stack(planes)[None].repeat(B, 1, 1, 1)
        return (pol + 1.5 * torch.randn(pol.shape, device="cuda", generator=g)).round().clamp(0, 255).to(torch.uint8)
    return torch.randint(0, 256, (B, 4, H, W), dtype=torch.uint8, device="cuda", generator=g)


def time_variant(B, want, iters=24, H=512, W=612, realistic=True, precise=False, sets=None, out_width=None, nt_loads=None):
    bpp = 4 + (8 if "xolp" in want else 0) + (8 if "xolp_std" in want else 0) + (36 if "normals" in want else 0)
    if sets is None:                       # enough rotating sets to exceed 2.5x the Infinity Cache
        per_set = B * H * (out_width or W) * bpp
        sets = max(1, min(8, -(-int(2.5 * 256e6) // per_set)))
    pols = [make_planes(B, H, W, realistic, seed=s) for s in range(sets)]
    outs = []
    for p in pols:
        o = pdpolar.polar_forward(p, want=want, precise=precise, out_width=out_width)
        outs.append(o)
    kw = dict(want=want, precise=precise, out_width=out_width, nt_loads=nt_loads)
    for i in range(sets):
        pdpolar.polar_forward(pols[i], out=outs[i], **kw)
    torch.cuda.synchronize()
    # one event pair per launch (kernel-only time: host launch gaps are excluded)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (e0, e1) in enumerate(evs):
        e0.record()
        pdpolar.polar_forward(pols[i % sets], out=outs[i % sets], **kw)
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    ms = ts[iters // 2]
    gbs = B * H * W * bpp / (ms * 1e-3) / 1e9
    return {"B": B, "want": list(want), "realistic": realistic, "precise": precise, "sets": sets,
            "out_width": out_width, "nt_loads": nt_loads,
            "ms": round(ms, 4), "ms_min": round(ts[0], 4), "bytes_px": bpp,
            "GBps": round(gbs, 1), "frac_8TBps": round(gbs / 8000, 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="only the bench.py configuration (B=16, 612->640) and B=128")
    args = ap.parse_args()
    if args.quick:
        print(json.dumps(time_variant(16, ("xolp", "normals"), out_width=640)), flush=True)
        print(json.dumps(time_variant(128, ("xolp", "normals"), out_width=640)), flush=True)
        print(json.dumps(time_variant(1, ("xolp", "normals"), out_width=640, sets=8)), flush=True)
        print(json.dumps(time_variant(1, ("xolp",), out_width=640, sets=8)), flush=True)
        print(json.dumps(time_variant(16, ("xolp", "normals"), sets=1, out_width=640)), flush=True)
        print(json.dumps(time_variant(128, ("xolp", "normals"))), flush=True)
        print(json.dumps(time_variant(16, ("xolp", "normals"), realistic=False)), flush=True)
        print(json.dumps(time_variant(16, ("xolp", "normals"), precise=True)), flush=True)
        print(json.dumps(time_variant(64, ("xolp",))), flush=True)
        sys.exit(0)
    for realistic in (True, False):
        for want, precise in ((("xolp",), False), (("xolp", "normals"), False), (("xolp", "normals"), True)):
            for B in (8, 16, 64, 128):
                print(json.dumps(time_variant(B, want, realistic=realistic, precise=precise)), flush=True)
