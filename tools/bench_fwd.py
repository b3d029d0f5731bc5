"""Warm timing of the forward / data-gradient implicit-GEMM kernels on the step's dominant shapes (one process, 20+20 launches)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402

SHAPES = [("3x3 64 @256x320", 64, 256, 320, 64, 3, 1), ("3x3 64 @128x160", 64, 128, 160, 64, 3, 1),
          ("5x5 64 @256x320", 64, 256, 320, 64, 5, 2), ("3x3 128 @64x80", 128, 64, 80, 128, 3, 1),
          ("3x3 256 @32x40", 256, 32, 40, 256, 3, 1), ("3x3 512 @16x20", 512, 16, 20, 512, 3, 1),
          ("5x5 128->256 @64x80", 128, 64, 80, 256, 5, 2)]
B = 16


def t(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for name, C, H, W, Co, k, p in SHAPES:
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0)
    dy = torch.randn_like(y)
    wt = ops.weight_transposed(w)
    flops = 2.0 * B * H * W * Co * C * k * k
    tf = t(lambda: ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0, out=y))
    td = t(lambda: ops.conv2d_dgrad(dy, w, (H, W), stride=1, pad=p, wt=wt))
    tw = t(lambda: ops.conv2d_wgrad(x, dy, w.shape, stride=1, pad=p, mode=0))
    print("%-22s fwd %.1f  dgrad %.1f  wgrad %.1f TF" % (name, flops / tf / 1e9, flops / td / 1e9, flops / tw / 1e9), flush=True)
