#!/usr/bin/env python3
"""Turn a rocprofv3 results database (rocpd .db, `rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the per-kernel summary committed next to it:  python profiles/summarize_rocpd.py X.db > X.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), avg(grid_x), "
    "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) "
    "from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>5s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s} "
      f"{'grid':>10s} {'vgpr':>5s} {'agpr':>5s} {'lds':>6s} {'scratch':>7s}")
for r in rows:
    if r[2] / tot < 5e-4:
        continue
    print(f"{r[0][:64]:64s} {r[1]:5d} {r[2]/1e6:10.3f} {r[3]/1e6:9.3f} {r[4]/1e6:9.3f} {r[5]/1e6:9.3f} "
          f"{100*r[2]/tot:6.2f} {int(r[6]):10d} {r[7]:5d} {r[8]:5d} {r[9]:6d} {r[10]:7d}")
