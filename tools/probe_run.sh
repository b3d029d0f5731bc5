#!/bin/bash
# GPU box: kernel times of the dominant conv shapes with the product library and with the timing-probe libraries
# (tools/build_probe.sh).  usage: tools/probe_run.sh <out.log> <probe names...>
cd $GRAFT_REPO_ROOT
OUT=$1; shift
: > $OUT
for lib in product "$@"; do
  echo "== $lib" >> $OUT
  if [ $lib = product ]; then unset PD_LIB; else export PD_LIB=$GRAFT_REPO_ROOT/tools/bin/libpd_probe_$lib.so; fi
  for only in "ResBlock1 3x3" "enc.Conv2 5x5" "ResBlock2 3x3" "joint.ResBlock1" "joint.ResBlock3" "joint.Conv2"; do
    ONLY="$only" timeout -k 10 120 python3 tools/bench_conv.py 2>/dev/null | grep layer >> $OUT || exit 1
  done
done
