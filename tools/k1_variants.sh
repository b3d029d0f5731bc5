#!/bin/bash
# K1 on the training step's output set (B=16 and B=128, 612 -> 640, back to back on rotating buffers): the nontemporal hint on
# the plane loads on / off (PD_POLAR_NT_LOADS; unset = the host's size rule: on for launches of up to 32 frames)
run() {
  env "$@" python - <<PY
import sys, json, os
sys.path.insert(0, "tools"); sys.path.insert(0, "supervised-depth-estimation-from-polarized-images_amd")
from bench_polar import time_variant
tag = "PD_POLAR_NT_LOADS=" + os.environ.get("PD_POLAR_NT_LOADS", "unset")
for B in (16, 128):
    r = time_variant(B, ("xolp", "normals"), out_width=640)
    print(tag, json.dumps({k: r[k] for k in ("B", "ms", "ms_min", "GBps", "frac_8TBps")}), flush=True)
PY
}
run PD_POLAR_NT_LOADS=1
run PD_POLAR_NT_LOADS=0
run PD_X=0
