#!/bin/bash
# kernel table of the 8192-window training step (GPU box): bash profiles/train_trace.sh [extra bench.py args]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt8 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt8 -- python $ROOT/bench.py --mode train --steps 3 --warmup 1 "$@" > /tmp/ktt8_line.json 2>/dev/null
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktt8/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 3 --warmup 1   (MSL shape, 8192 windows per step, 4 training steps)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%-72s %4d calls %8.3f ms/step %5.1f%%" % (r["Name"][:72], int(r["Calls"]), float(r["TotalDurationNs"]) / 4e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 600 /tmp/ktt8_line.json
