#!/bin/bash
# GPU box: 3x3 weight gradients, rolling-row kernel vs one filter row per workgroup (PD_WGRAD_ROLL=0)
cd $GRAFT_REPO_ROOT
for h in 1 0; do
  echo "== PD_WGRAD_ROLL=$h"
  for only in "enc.ResBlock1 3x3" "enc.ResBlock2" "dec.upconv(2,1) 128"; do
    PD_WGRAD_ROLL=$h ONLY="$only" timeout -k 10 120 python3 tools/bench_conv.py 2>/dev/null | grep layer | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print("%-52s wgrad %6.3f ms %6.1f TF" % (d["layer"], d["wgrad_ms"], d["wgrad_TF"]))'
  done
done
