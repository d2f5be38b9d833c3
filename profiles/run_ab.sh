#!/bin/bash
# developer helper: A/B of the recurrence kernels on the GPU box (see profiles/gru_ab.cpp)
mkdir -p gpurun_out
N=${1:-65536}
KB=${2:-2}
SHAPE=$3
timeout 300 profiles/bin/gru_ab $N 3 $KB $SHAPE > gpurun_out/ab_${N}_$KB.log 2>&1
echo "exit $?" >> gpurun_out/ab_${N}_$KB.log
tail -60 gpurun_out/ab_${N}_$KB.log
