"""Timing-only ablations of the uniform-tap conv kernel (library built with -DPD_CONV_ABLATE; results are wrong by design).
PD_ABL bits: 1 no LDS-DMA issue, 2 zero-record descriptors (loads dropped by the range check), 4 no vmcnt wait + barrier,
32 no epilogue.  (Never drop the validity bits: the gather would leave the tensor.)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402

SHAPES = [("3x3 64 @256x320", 64, 256, 320, 64, 3, 1), ("3x3 64 @128x160", 64, 128, 160, 64, 3, 1),
          ("5x5 64 @256x320", 64, 256, 320, 64, 5, 2), ("3x3 128 @64x80", 128, 64, 80, 128, 3, 1),
          ("3x3 512 @16x20", 512, 16, 20, 512, 3, 1)]
ABLS = [int(v) for v in os.environ.get("ABLS", "0,1,2,4,32,6,38,39").split(",")]
B = 16
for name, C, H, W, Co, k, p in SHAPES:
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0)
    flops = 2.0 * B * H * W * Co * C * k * k
    out = []
    for abl in ABLS:
        os.environ["PD_ABL"] = str(abl)
        for _ in range(20):
            ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0, out=y)
        e1.record(); torch.cuda.synchronize()
        out.append("%d:%.1f" % (abl, flops / (e0.elapsed_time(e1) / 20) / 1e9))
    print(name, " ".join(out), flush=True)
