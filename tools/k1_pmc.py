"""A handful of K1 launches (one per variant) for `rocprofv3 --pmc ...` counter collection."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
import bench_polar  # noqa: E402

if __name__ == "__main__":
    B = int(os.environ.get("K1_B", 64))
    for realistic in (True, False):
        for want, precise in ((("xolp",), False), (("xolp", "normals"), False), (("xolp", "normals"), True)):
            print(bench_polar.time_variant(B, want, iters=4, realistic=realistic, precise=precise), flush=True)
