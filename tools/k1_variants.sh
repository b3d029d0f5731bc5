#!/bin/bash
# K1 on the training step's output set (B=16 and B=128, 612 -> 640, back to back on rotating buffers): the nontemporal hint on
# the plane loads forced on / off (pd_polar_fwd flags PD_POLAR_NT_LOADS / PD_POLAR_PLAIN_LOADS) and the library's size rule
python3 - <<PY
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, "supervised-depth-estimation-from-polarized-images_amd")
from bench_polar import time_variant
for nt in (True, False, None):
    for B in (16, 128):
        r = time_variant(B, ("xolp", "normals"), out_width=640, nt_loads=nt)
        print("nt_loads=%s" % nt, json.dumps({k: r[k] for k in ("B", "ms", "ms_min", "GBps", "frac_8TBps")}), flush=True)
PY
