"""Why is K1 slower inside the training step than back to back?  Kernel-only HIP-event time of the bench.py
configuration (B=16, 612->640) after different predecessors / with different output buffers."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from polardepth import polar as pdpolar
from bench_polar import make_planes

NB = int(os.environ.get('PROBE_B', '16'))
pols = [make_planes(NB, 512, 612, True, seed=s) for s in range(3)]
outs = [pdpolar.polar_forward(p, want=("xolp", "normals"), out_width=640) for p in pols]
big = torch.zeros(256 * 1000 * 1000 // 4, device="cuda")
big2 = torch.zeros(1000 * 1000 * 1000 // 4, device="cuda")


WANT = ("xolp", "normals")
PREC = False


def run(name, pre, fresh):
    ts = []
    for i in range(14):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = pdpolar.polar_forward(pols[i % 3], want=WANT, precise=PREC, out_width=640, out=None if fresh else {k: outs[i % 3][k] for k in WANT})
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
        del o
    ts = sorted(ts[2:])
    print(json.dumps({"B": NB, "want": WANT, "precise": PREC, "case": name, "fresh_outputs": fresh, "median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1)}), flush=True)


for WANT, PREC in ((("xolp", "normals"), False), (("xolp", "normals"), True)):
    run("back to back", lambda: None, False)
    run("after zero_ of 85 MB", lambda: big[:85 * 250000].zero_(), False)
    run("after add_ on 1 GB", lambda: big2.add_(1.0), False)
