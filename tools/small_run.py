"""Eleven launches of one small-plane layer (3x3x512 @16x20 by default: the gather kernel's 128-row tiles) for the counter passes
of tools/sq_prof_k.sh.  OP = fwd | dgrad | wgrad, C / HH / WW (environment)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402
op = os.environ.get("OP", "fwd")
B, C, H, W = 16, int(os.environ.get("C", 512)), int(os.environ.get("HH", 16)), int(os.environ.get("WW", 20))
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(C, C, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
wt = ops.weight_transposed(w)
for _ in range(11):
    if op == "fwd":
        ops.conv2d_fwd(x, w, None, stride=1, pad=1, mode=0)
    elif op == "dgrad":
        ops.conv2d_dgrad(dy, w, (H, W), stride=1, pad=1, wt=wt)
    else:
        ops.conv2d_wgrad(x, dy, (C, C, 3, 3), stride=1, pad=1, mode=0)
torch.cuda.synchronize()
