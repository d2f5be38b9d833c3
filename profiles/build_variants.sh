#!/bin/bash
# developer helper: builds profiles/bin/variants/<name>/libmtadgat.so with extra flags for the chunk-major recurrence TU only
# usage: build_variants.sh name1 "flags1" name2 "flags2" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/mtad-gat-pytorch_amd/build
mkdir -p $ROOT/profiles/bin/variants
pids=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  d=$ROOT/profiles/bin/variants/$name; mkdir -p $d
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $flags -x hip -c $ROOT/mtad-gat-pytorch_amd/csrc/mtadgat_gru_cm.hip -o $d/cm.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v mtadgat_gru_cm.o) $d/cm.o -o $d/libmtadgat.so && rm $d/cm.o && echo built $name ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
