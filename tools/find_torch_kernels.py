"""Which Python lines launch the remaining ATen kernels of a train step (torch.profiler with stacks)."""
import os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
import bench  # noqa: E402
from polardepth import synthetic  # noqa: E402

tr = bench.build_trainer(16, bench.H, bench.W, tempfile.mkdtemp())
batch = synthetic.make_batch(16, bench.H, bench.W, frame_w=bench.FRAME_W, device="cuda")
for _ in range(2):
    bench.train_step(tr, batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(tr, batch)
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.key.startswith("aten::") and ev.device_time_total > 20:
        print("%-28s n=%d cuda_us=%.0f" % (ev.key, ev.count, ev.device_time_total))
        for fr in ev.stack[:6]:
            if "site-packages" not in fr and "torch/" not in fr:
                print("      ", fr)
