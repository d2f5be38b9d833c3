#!/bin/bash
# developer helper: the regular library against variant builds of one translation unit (profiles/build_variants.sh),
# full forward at the flagship shape through profiles/bin/gru_ab (per-family kernel times)
# usage: run_gat_ab.sh <windows> <variant> [<variant> ...]
mkdir -p gpurun_out
N=${1:-65536}; shift
echo "== regular build"; timeout 200 profiles/bin/gru_ab $N 3 2 msl 2>&1 | grep -E "kernel 2|recons"
for v in "$@"; do
  echo "== variant $v"
  LD_PRELOAD=$PWD/profiles/bin/variants/$v/libmtadgat.so timeout 200 profiles/bin/gru_ab $N 3 2 msl 2>&1 | grep -E "kernel 2|recons"
done
