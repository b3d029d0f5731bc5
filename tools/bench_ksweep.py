"""Fixed per-tile cost vs per-chunk cost of the uniform-tap kernel: same output grid and Co, contraction length swept."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402

B, H, W, Co = 16, 128, 160, 64
tiles = B * H * W // 128
for C in (32, 64, 128, 256, 512):
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(x, w, None, stride=1, pad=1, mode=0)
    for _ in range(20):
        ops.conv2d_fwd(x, w, None, stride=1, pad=1, mode=0, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_fwd(x, w, None, stride=1, pad=1, mode=0, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    flops = 2.0 * B * H * W * Co * C * 9
    nch = 9 * C // 32
    # 768 resident workgroups: a tile's wall time = ms * 768 / tiles
    print("C=%3d chunks/tile=%3d  %.3f ms  %.1f TF  us per tile-slot %.2f  us per chunk %.3f" % (
        C, nch, ms, flops / ms / 1e9, ms * 1e3 * 768 / tiles, ms * 1e3 * 768 / tiles / nch), flush=True)
