"""Kernel-only timing of K1 (pd_polar_fwd) with HIP events: GB/s vs batch and output variant."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import polar as pdpolar  # noqa: E402


def time_variant(B, want, iters=20, H=512, W=612, realistic=True, precise=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    if realistic:
        yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
        iun = 120 + 60 * torch.sin(xx / 40.0) * torch.cos(yy / 30.0)
        rho = 0.02 + 0.25 * (0.5 + 0.5 * torch.sin(xx / 25.0 + yy / 50.0)) ** 2
        phi = 1.57 * torch.sin(xx / 60.0 - yy / 35.0)
        planes = [iun * (1 + rho * torch.cos(2 * a - 2 * phi)) for a in (0.0, 0.7853981, 1.5707963, 2.3561945)]
        pol = torch.stack(planes)[None].repeat(B, 1, 1, 1)
        pol = (pol + 1.5 * torch.randn(pol.shape, device="cuda", generator=g)).round().clamp(0, 255).to(torch.uint8)
    else:
        pol = torch.randint(0, 256, (B, 4, H, W), dtype=torch.uint8, device="cuda", generator=g)
    outs = pdpolar.polar_forward(pol, want=want, precise=precise)   # outputs are allocated once and reused
    for _ in range(3):
        pdpolar.polar_forward(pol, want=want, out=outs, precise=precise)
    torch.cuda.synchronize()
    # one event pair per launch (kernel-only time: host launch gaps are excluded)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in evs:
        e0.record()
        pdpolar.polar_forward(pol, want=want, out=outs, precise=precise)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[iters // 2]
    bpp = 4 + (8 if "xolp" in want else 0) + (8 if "xolp_std" in want else 0) + (36 if "normals" in want else 0)
    gbs = B * H * W * bpp / (ms * 1e-3) / 1e9
    return {"B": B, "want": list(want), "realistic": realistic, "precise": precise, "ms": round(ms, 4), "bytes_px": bpp,
            "GBps": round(gbs, 1), "frac_8TBps": round(gbs / 8000, 3)}


if __name__ == "__main__":
    for realistic in (True, False):
        for want, precise in ((("xolp",), False), (("xolp", "normals"), False), (("xolp", "normals"), True)):
            for B in (8, 16, 64, 128):
                print(json.dumps(time_variant(B, want, realistic=realistic, precise=precise)), flush=True)
