#!/bin/bash
# builds variants of libtbk.so under gpurun_exp/<name>/ : name = flags for wgroup.hip and collapse.hip (the other objects are shared)
# usage: wgopt_build.sh name "-DSOME_SWITCH=0" ...   (round 5 timed the window kernel's cuts this way: -DWG_OPT=<bits>, since removed)
set -e
cd /root/repo/tiebrush_amd/csrc
B=../_build
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift; shift
  D=/root/repo/gpurun_exp/$name; mkdir -p $D
  ( /opt/rocm/bin/hipcc $FL $flags -c wgroup.hip -o $D/wgroup.o 2>&1 | grep -E "error|spill" ; true ) &
  ( /opt/rocm/bin/hipcc $FL $flags -c collapse.hip -o $D/collapse.o 2>&1 | grep -E "error|spill" ; true ) &
  wait
  objs=""
  for o in tbk_api prims msort cov shard bamdev pack bgzdef; do objs="$objs $B/$o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libtbk.so $objs $D/wgroup.o $D/collapse.o
  rm -f $D/*.o
  echo built $name
done
