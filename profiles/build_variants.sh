#!/bin/bash
# developer helper: builds profiles/bin/variants/<name>/libmtadgat.so with extra flags for ONE translation unit
# (TU=mtadgat_gru_cm by default; e.g. TU=mtadgat_gat), the other objects taken from the regular build
# usage: [TU=<name>] build_variants.sh name1 "flags1" name2 "flags2" ...
set -e
TU=${TU:-mtadgat_gru_cm}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/mtad-gat-pytorch_amd/build
mkdir -p $ROOT/profiles/bin/variants
pids=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  d=$ROOT/profiles/bin/variants/$name; mkdir -p $d
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $flags -x hip -c $ROOT/mtad-gat-pytorch_amd/csrc/$TU.hip -o $d/cm.o $SHOW &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v /$TU.o) $d/cm.o -o $d/libmtadgat.so && rm $d/cm.o && echo built $name ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
