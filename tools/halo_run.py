"""Eleven launches of one halo-tile kernel on a 64 -> 64 layer @256x320 (B = 16) for the counter passes of tools/sq_prof_k.sh.
OP = fwd | dgrad | wgrad, KS = 3 | 5 (environment)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402
op, k = os.environ.get("OP", "fwd"), int(os.environ.get("KS", 3))
B, C, H, W, Co, p = 16, 64, 256, 320, 64, k // 2
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(Co, C, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
wt = ops.weight_transposed(w)
for _ in range(11):
    if op == "fwd":
        ops.conv2d_fwd(x, w, None, stride=1, pad=p, mode=0)
    elif op == "dgrad":
        ops.conv2d_dgrad(dy, w, (H, W), stride=1, pad=p, wt=wt)
    else:
        ops.conv2d_wgrad(x, dy, (Co, C, k, k), stride=1, pad=p, mode=0)
torch.cuda.synchronize()
