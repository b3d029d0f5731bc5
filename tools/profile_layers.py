"""Per-layer time of the conv kernels inside one real train step (HIP events around every launch, serial stream)."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
import bench  # noqa: E402
from polardepth import ops, synthetic  # noqa: E402
from polardepth import functional as PF  # noqa: E402

if __name__ == "__main__":
    B = int(os.environ.get("B", 16))
    tr = bench.build_trainer(B, bench.H, bench.W, tempfile.mkdtemp())
    batch = synthetic.make_batch(B, bench.H, bench.W, frame_w=bench.FRAME_W, device="cuda")
    for _ in range(2):
        bench.train_step(tr, batch)
    PF.USE_WGRAD_STREAM = False
    tr.encoder_streams = False
    ops.PROFILE = []
    bench.train_step(tr, batch)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for name, flops, e0, e1, shape in prof:
        a = agg.setdefault(shape, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
    tot = sum(a[1] for a in agg.values())
    print(f"conv kernels: {tot:.2f} ms per step")
    for shape, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(json.dumps({"shape": shape, "calls": n, "ms": round(ms, 3), "TF": round(fl / ms / 1e9, 1)}))
