"""Timing of the 16-channel tail kernels (pd_conv16 fwd / dgrad, pd_conv16_wgrad) at the decoder's shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402


def t(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for C, H, W in ((16, 512, 640), (32, 256, 320)):
    B = 16
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(16, C, 3, 3, device="cuda") * 0.1).contiguous(memory_format=torch.channels_last)
    b = torch.randn(16, device="cuda")
    y = ops.conv2d_fwd(x, w, b, 1, 1, mode=1, act=ops.ACT_ELU)
    dy = torch.randn_like(y)
    wt = ops.weight_transposed(w)
    fl = 2.0 * B * H * W * 16 * C * 9
    tf = t(lambda: ops.conv2d_fwd(x, w, b, 1, 1, mode=1, act=ops.ACT_ELU, out=y))
    td = t(lambda: ops.conv2d_dgrad(dy, w, (H + 2, W + 2), 1, 0, wt=wt))
    tw = t(lambda: ops.conv2d_wgrad(x, dy, w.shape, 1, 1, mode=1, want_bias=True))
    print("C=%d @%dx%d  fwd %.3f ms %.1f TF | dgrad %.3f ms %.1f TF | wgrad %.3f ms %.1f TF" % (
        C, H, W, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9), flush=True)
