"""Which ATen ops (copies, fills, adds) still run inside one training step, and from which Python frames?"""
import os, sys, tempfile, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
import torch
import bench
from polardepth import synthetic
from torch.utils._python_dispatch import TorchDispatchMode

tr = bench.build_trainer(16, 512, 640, tempfile.mkdtemp())
tr.set_train()
batch = synthetic.make_batch(16, 512, 640, frame_w=612, device="cuda", seed=0)
batch[("pol", 0, 0)] = batch[("pol", 0, 0)][..., :612].contiguous()
batch.pop("depth_gt"); batch.pop(("mask", 0, 0))
for _ in range(2):
    bench.train_step(tr, batch)
torch.cuda.synchronize()

counts = collections.Counter()
where = collections.defaultdict(collections.Counter)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        counts[name] += 1
        if any(k in name for k in ("copy", "fill", "zero", "add", "mul", "cat", "clone", "contiguous", "to.")):
            fr = [f for f in traceback.extract_stack() if "polarized-images_amd" in f.filename or f.filename.endswith("bench.py")]
            if fr:
                f = fr[-1]
                where[name][f"{os.path.basename(f.filename)}:{f.lineno}"] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    bench.train_step(tr, batch)
torch.cuda.synchronize()
for k, v in counts.most_common(40):
    print(v, k, dict(where[k].most_common(6)) if k in where else "")
