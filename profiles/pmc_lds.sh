#!/bin/bash
# LDS / issue counters of the flagship forward's kernels (65 536 windows; k_gath, k_conv_win, k_gru_cm):
#   bash profiles/pmc_lds.sh r04   ->  gpurun_out/prof_<tag>/<tag>_pmc_lds.txt       (one rocprofv3 --pmc pass, kernel trace only)
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
rm -rf /tmp/pl && timeout 300 rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d /tmp/pl -- python "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-sub > /dev/null 2>&1
cp "$(find /tmp/pl -name '*counter_collection.csv' | head -1)" "$OUT/pmc_lds.csv"
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(os.path.join(out, "pmc_lds.csv"))):
    k = r["Kernel_Name"]
    if "mtadgat" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
with open(os.path.join(out, f"{tag}_pmc_lds.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES\n"
            "#   -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sub   (MI355X, 65 536 MSL windows); per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:8]:
        n = len(disp[k]); wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        f.write(f"{k[:90]:90s} n={n}\n    " + "  ".join(f"{c}={x / n:.4g}" for c, x in sorted(v.items())) + "\n"
                f"    -> of the wave cycles: VALU issue {100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} %, LDS issue {100 * v.get('SQ_ACTIVE_INST_LDS', 0) / wc:.0f} %, "
                f"waiting on LDS {100 * v.get('SQ_WAIT_INST_LDS', 0) / wc:.0f} %, waiting on anything {100 * v.get('SQ_WAIT_ANY', 0) / wc:.0f} %; "
                f"bank-conflict cycles per LDS instruction {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_INSTS_LDS', 1), 1):.2f}\n")
PY
cat "$OUT/${TAG}_pmc_lds.txt"
