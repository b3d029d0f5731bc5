#!/bin/bash
# GPU box: the conv tail (stems in space-to-depth form, 32-column decoder layers, small planes) with the product library and
# A/B builds.  usage: tools/tail_ab.sh <out.log> [probe names...]
cd $GRAFT_REPO_ROOT
OUT=$1; shift
: > $OUT
run() {
  for only in "stem s2d normals" "stem s2d rgb" "stem s2d xolp" "dec.upconv(1,1) 96->32" "dec.upconv(1,1) as zero" "dec.upconv(2,1)b" "enc.ResBlock3" "joint.ResBlock3" "joint.ResBlock5"; do
    ONLY="$only" timeout -k 10 120 python3 tools/bench_conv.py 2>/dev/null | grep layer | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print("%-52s fwd %6.3f ms %6.1f TF   dgrad %6.3f ms %6.1f TF   wgrad %6.3f ms %6.1f TF" % (d["layer"], d["fwd_ms"], d["fwd_TF"], d.get("dgrad_ms", 0), d.get("dgrad_TF", 0), d["wgrad_ms"], d["wgrad_TF"]))' >> $OUT || exit 1
  done
}
echo "== product" >> $OUT; unset PD_LIB; run
for lib in "$@"; do echo "== $lib" >> $OUT; PD_LIB=$GRAFT_REPO_ROOT/tools/bin/libpd_probe_$lib.so run; done
