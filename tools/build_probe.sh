#!/bin/bash
# Timing-probe builds of the library (NOT the product): conv.hip compiled with one PD_PROBE_* macro, linked with the product's
# other objects into tools/bin/libpd_probe_<name>.so.  Load one with PD_LIB=<path> (polardepth/_lib.py).  Results of a probe
# library are meaningless by construction -- only its kernel times are read.
#   usage: tools/build_probe.sh NOSPLIT NOSPLIT_W NOWALK ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/supervised-depth-estimation-from-polarized-images_amd/csrc
make -C $CS -j4 >/dev/null
mkdir -p $ROOT/tools/bin
for name in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
      -DPD_PROBE_$name -c $CS/conv.hip -o $ROOT/tools/bin/conv_probe_$name.o
  objs=$(ls $CS/build/*.o | grep -v '/conv.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/bin/libpd_probe_$name.so $ROOT/tools/bin/conv_probe_$name.o $objs
  echo built tools/bin/libpd_probe_$name.so
done
