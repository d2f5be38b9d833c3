#!/bin/bash
# SQ counters of the kernels that collect.sh's flagship passes do not reach: the 8 192-window forward (hidden-tile split on split
# operands) and the 8 192-window training step.   bash profiles/pmc_extra.sh r03   ->  gpurun_out/prof_<tag>/<tag>_pmc_extra.txt
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  rm -rf /tmp/pe_f$i && timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pe_f$i -- python "$ROOT/bench.py" --steps 1 --warmup 1 --batch 8192 --no-cpu-baseline --no-sub > /dev/null 2>&1
  cp "$(find /tmp/pe_f$i -name '*counter_collection.csv' | head -1)" "$OUT/pmcx_fwd8192_$i.csv"
  rm -rf /tmp/pe_t$i && timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pe_t$i -- python "$ROOT/bench.py" --mode train --steps 1 --warmup 1 > /dev/null 2>&1
  cp "$(find /tmp/pe_t$i -name '*counter_collection.csv' | head -1)" "$OUT/pmcx_train8192_$i.csv"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, os, sys
out, tag = sys.argv[1], sys.argv[2]
with open(os.path.join(out, f"{tag}_pmc_extra.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> (one pass per group), MI355X; sums over the dispatches of a kernel / dispatch count.\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (sum of 8 XCDs) in cycles.\n")
    for what in ("fwd8192", "train8192"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
        for path in sorted(glob.glob(os.path.join(out, f"pmcx_{what}_*.csv"))):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if "mtadgat" not in k: continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (path, k, r["Dispatch_Id"]) not in seen:
                    seen.add((path, k, r["Dispatch_Id"])); n[(path, k)] += 1
        f.write(f"== {what}: python bench.py " + ("--batch 8192 --steps 1 --warmup 1" if what == "fwd8192" else "--mode train --steps 1 --warmup 1") + "\n")
        rows = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))
        for k, v in rows[:14]:
            disp = max(c for (p, kk), c in n.items() if kk == k)
            f.write(f"{k[:86]:86s} n={disp}\n    " + "  ".join(f"{c}={x / disp:.4g}" for c, x in sorted(v.items())) + "\n")
PY
cat "$OUT/${TAG}_pmc_extra.txt" | head -70
