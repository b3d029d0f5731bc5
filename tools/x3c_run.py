"""Ten weight-gradient launches of the 5x5 64 -> 64 @256x320 layer (B = 16) for the counter passes of tools/sq_prof_k.sh."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402
B, C, H, W, Co, k, p = 16, 64, 256, 320, 64, 5, 2
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(11):
    ops.conv2d_wgrad(x, dy, (Co, C, k, k), stride=1, pad=p, mode=0)
torch.cuda.synchronize()
