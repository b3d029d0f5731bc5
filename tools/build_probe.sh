#!/bin/bash
# Timing-probe / A-B builds of the library (NOT the product): conv.hip compiled with extra macros, linked with the product's
# other objects into tools/bin/libpd_probe_<name>.so.  Load one with PD_LIB=<path> (polardepth/_lib.py).  A PD_PROBE_* macro
# makes results meaningless by construction (only kernel times are read); other macros select a code variant under test.
#   usage: tools/build_probe.sh NAME:MACRO[,MACRO...] ...      e.g.  NOSPLIT:PD_PROBE_NOSPLIT  SPREAD:PD_X3C_SPREAD
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/supervised-depth-estimation-from-polarized-images_amd/csrc
make -C $CS -j4 >/dev/null
mkdir -p $ROOT/tools/bin
for spec in "$@"; do
  name=${spec%%:*}; macros=${spec#*:}
  defs=""; for m in ${macros//,/ }; do defs="$defs -D$m"; done
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
      $defs -c $CS/conv.hip -o $ROOT/tools/bin/conv_probe_$name.o 2>/dev/null
    objs=$(ls $CS/build/*.o | grep -v '/conv.o')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libpd_probe_$name.so $ROOT/tools/bin/conv_probe_$name.o $objs
    echo built tools/bin/libpd_probe_$name.so "($defs)" ) &
done
wait
