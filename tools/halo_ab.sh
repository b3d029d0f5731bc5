#!/bin/bash
# GPU box: forward / data-gradient times of the halo-tile kernel's layers: product library, A/B builds (tools/build_probe.sh)
# and the per-tap gather kernel (PD_CONV_HALO=0).  usage: tools/halo_ab.sh <out.log> [probe names...]
cd $GRAFT_REPO_ROOT
OUT=$1; shift
: > $OUT
run() {
  for only in "enc.ResBlock1 3x3" "enc.Conv2 5x5" "ResBlock2 3x3" "enc.Conv3" "joint.ResBlock1" "joint.Conv1" "joint.ResBlock3" "joint.Conv2" "dec.upconv(2,1)" "dec.upconv(3,1)"; do
    ONLY="$only" timeout -k 10 120 python3 tools/bench_conv.py 2>/dev/null | grep layer | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print("%-44s fwd %6.3f ms %6.1f TF   dgrad %6.3f ms %6.1f TF   wgrad %6.3f ms %6.1f TF" % (d["layer"], d["fwd_ms"], d["fwd_TF"], d.get("dgrad_ms", 0), d.get("dgrad_TF", 0), d["wgrad_ms"], d["wgrad_TF"]))' >> $OUT || exit 1
  done
}
echo "== product (halo)" >> $OUT; unset PD_LIB; PD_CONV_HALO=1 run
for lib in "$@"; do echo "== $lib" >> $OUT; PD_LIB=$GRAFT_REPO_ROOT/tools/bin/libpd_probe_$lib.so PD_CONV_HALO=1 run; done
echo "== product, PD_CONV_HALO=0 (conv_igemm_x3_kernel)" >> $OUT; unset PD_LIB; PD_CONV_HALO=0 run
