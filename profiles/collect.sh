#!/bin/bash
# Regenerates the files under profiles/ on a GPU box:  bash profiles/collect.sh r04 <git rev>
# (rocprofv3 passes are separate: --kernel-trace --stats, then one --pmc pass per counter group).
set -u
TAG=${1:-r03}
REV=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/bench.py" > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
# 1. kernel trace of the flagship forward (fp32), of the same workload with bf16 operands, and of the training step
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-sub > "$OUT/kt_bench_line.json" 2> /dev/null
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf /tmp/ktb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-sub --precision bf16 > "$OUT/kt_bf16_bench_line.json" 2> /dev/null
cp "$(find /tmp/ktb -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_bf16.csv"
rm -rf /tmp/ktt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python "$ROOT/profiles/train_step.py" 256 > "$OUT/train_step_256.txt" 2> /dev/null
cp "$(find /tmp/ktt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_train256.csv"
rm -rf /tmp/ktt8 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt8 -- python "$ROOT/bench.py" --mode train --steps 3 --warmup 1 > "$OUT/train_bench_line.json" 2> /dev/null
cp "$(find /tmp/ktt8 -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_train8192.csv"
# (the line above was taken UNDER rocprofv3: its ms_per_step carries the tracer's overhead; the plain run is the number to quote)
python "$ROOT/bench.py" --mode train --steps 5 --warmup 2 > "$OUT/train_bench_line_plain.json" 2> /dev/null
rm -rf /tmp/ktf && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -- python "$ROOT/profiles/forward_small.py" 256 > "$OUT/forward_256.txt" 2> /dev/null
cp "$(find /tmp/ktf -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_fwd256.csv"
rm -rf /tmp/kts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -- python "$ROOT/profiles/series_bench.py" > "$OUT/series_65536.txt" 2> /dev/null
cp "$(find /tmp/kts -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_series.csv"
python "$ROOT/profiles/batch_sweep.py" 2048 4096 8192 9216 10240 12288 16384 24576 32768 36864 40960 49152 65536 131072 > "$OUT/batch_sweep.txt" 2> /dev/null
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) and the SQ group
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python "$ROOT/bench.py" --steps 1 --warmup 1 --batch 65536 --no-cpu-baseline --no-sub > /dev/null 2>&1
  cp "$(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$C.csv"
done
rm -rf /tmp/pmc_sq && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-sub > /dev/null 2>&1
cp "$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -1)" "$OUT/pmc_SQ.csv"
python "$ROOT/profiles/summarize.py" "$OUT" "$TAG" "$REV"
