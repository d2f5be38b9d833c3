#!/bin/bash
# developer helper: SQ counters of the recurrence kernels (A/B harness, MSL shape) -- one rocprofv3 pass per counter group
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${1:-65536}
i=0
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/p$i
  timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/p$i -- $ROOT/profiles/bin/gru_ab $N 1 2 msl > $OUT/run$i.log 2>&1
  f=$(find /tmp/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc$i.csv
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "k_gru" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    seen=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "k_gru" in k: seen[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        print(f, k)
        for c,x in v.items(): print("     %-28s %.4e  (n=%d, per launch %.4e)"%(c,x,seen[(k,c)],x/seen[(k,c)]))
PY
