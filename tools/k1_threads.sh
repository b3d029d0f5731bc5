#!/bin/bash
# K1 workgroup-size sweep (bench configuration B=16 and B=128, 612 -> 640, xolp + normals)
for T in 1024 512 256; do
  PD_POLAR_THREADS=$T python - <<PY
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, "supervised-depth-estimation-from-polarized-images_amd")
from bench_polar import time_variant
for B in (16, 128):
    print(json.dumps(time_variant(B, ("xolp", "normals"), out_width=640)), flush=True)
PY
done
