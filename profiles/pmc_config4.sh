#!/bin/bash
# SQ counters of config 4's kernels (one 896-window chunk of F = 512, W = 256):  bash profiles/pmc_config4.sh r06  ->  gpurun_out/prof_<tag>/<tag>_pmc_config4.txt
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/c4one.py <<PY
import sys, os
ROOT = "$ROOT"
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from mtad_gat import MTAD_GAT
dev = torch.device("cuda", 0)
kw = dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, gru_hid_dim=150, forecast_n_layers=1, forecast_hid_dim=150, recon_hid_dim=150)
torch.manual_seed(0)
m = MTAD_GAT(**kw).to(dev).eval(); m.check_weight_contents = False
x = torch.rand(896, 256, 512, device=dev)
with torch.no_grad():
    for _ in range(2): m(x)
torch.cuda.synchronize()
PY
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  rm -rf /tmp/pc4_$i && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pc4_$i -- python /tmp/c4one.py > /dev/null 2>&1
  cp "$(find /tmp/pc4_$i -name '*counter_collection.csv' | head -1)" "$OUT/pmc_c4_$i.csv"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for path in sorted(glob.glob(os.path.join(out, "pmc_c4_*.csv"))):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "mtadgat" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open(os.path.join(out, f"{tag}_pmc_config4.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> (one pass per group), MI355X, config 4 (F = 512, W = 256), one 896-window chunk; per dispatch averages.\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (sum of 8 XCDs) in cycles.\n")
    for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
        c = {name: acc[k][name] / max(n[k][name], 1) for name in acc[k]}
        if c.get("GRBM_GUI_ACTIVE", 0) < 1e5: continue
        f.write(f"{k[:100]}\n")
        gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
        f.write(f"   dispatches {int(n[k]['SQ_WAVES'])}  engine cycles {gui:.3e}  waves {c.get('SQ_WAVES', 0):.0f}  MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024 + 1):.3f} of SIMD-cycles  "
                f"insts MFMA {c.get('SQ_INSTS_MFMA', 0):.3e} VALU {c.get('SQ_INSTS_VALU', 0):.3e} LDS {c.get('SQ_INSTS_LDS', 0):.3e} VMEM {c.get('SQ_INSTS_VMEM', 0):.3e} SALU {c.get('SQ_INSTS_SALU', 0):.3e}\n")
        wc = c.get("SQ_WAVE_CYCLES", 1)
        f.write(f"   wave-cycles {wc:.3e}: waiting (s_waitcnt / barrier) {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} (LDS {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f})  "
                f"VALU active {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  LDS active {c.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}  bank-conflict cycles per LDS inst {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_INSTS_LDS', 1), 1):.2f}\n")
print(open(os.path.join(out, f"{tag}_pmc_config4.txt")).read())
PY
