#!/bin/bash
# kernel trace + SQ counters of the attention kernels on the flagship workload (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_gat2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub > /dev/null 2>&1
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p1 -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub > /dev/null 2>&1
cp "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" $OUT/pmc_sq1.csv
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d /tmp/p2 -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub > /dev/null 2>&1
cp "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" $OUT/pmc_sq2.csv
python - <<PY
import csv, collections
for f in ("pmc_sq1.csv", "pmc_sq2.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open("$OUT/" + f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in acc.items():
        if "gat" in k or "conv" in k:
            print(k, {c: "%.3e" % x for c, x in v.items()})
PY
head -8 $OUT/kernel_stats.csv | cut -c1-160
