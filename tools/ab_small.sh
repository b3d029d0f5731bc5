cd $GRAFT_REPO_ROOT
for h in 1 0; do
  echo "== PD_CONV_HALO=$h"
  for only in "enc.ResBlock3" "joint.ResBlock3" "joint.Conv2" "dec.upconv(3,1)"; do
    PD_CONV_HALO=$h ONLY="$only" timeout -k 10 120 python3 tools/bench_conv.py 2>/dev/null | grep layer | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print("%-52s fwd %6.3f ms %6.1f TF   dgrad %6.3f ms %6.1f TF   wgrad %6.3f ms %6.1f TF" % (d["layer"], d["fwd_ms"], d["fwd_TF"], d.get("dgrad_ms", 0), d.get("dgrad_TF", 0), d["wgrad_ms"], d["wgrad_TF"]))'
  done
done
