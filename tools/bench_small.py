"""Timing of the decoder-tail direct kernels (pd_smallconv_*, pd_disphead_*) at the train-step shapes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth import ops  # noqa: E402
from polardepth import functional as PF  # noqa: E402

B = int(os.environ.get("B", 16))


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


PF.USE_WGRAD_STREAM = False
for name, C, Co, H, W, act in (("upconv(0,1) 16->16 @512x640", 16, 16, 512, 640, ops.ACT_ELU),
                               ("upconv(0,0) 32->16 @256x320", 32, 16, 256, 320, ops.ACT_ELU),
                               ("dispconv0 16->1 @512x640", 16, 1, 512, 640, ops.ACT_SIGMOID),
                               ("dispconv1 32->1 @256x320", 32, 1, 256, 320, ops.ACT_SIGMOID)):
    conv = torch.nn.Conv2d(C, Co, 3).cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = PF.reflect_conv_act(x, conv, act)
    gy = torch.randn_like(y)
    t_f = timeit(lambda: PF.reflect_conv_act(x, conv, act))
    def fb():
        yy = PF.reflect_conv_act(x, conv, act)
        yy.backward(gy)
    t_fb = timeit(fb)
    print(json.dumps({"layer": name, "fn": y.grad_fn.__class__.__name__, "fwd_ms": round(t_f, 3), "fwd+bwd_ms": round(t_fb, 3)}), flush=True)
