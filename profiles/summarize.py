#!/usr/bin/env python3
"""Turns the raw rocprofv3 CSVs written by collect.sh into the committed summaries:
   python profiles/summarize.py gpurun_out/prof_r01 r01  ->  gpurun_out/prof_r01/{r01_kernel_trace_stats.txt,
   r01_pmc_summary.txt, r01_traffic.json, r01_bench_line.json}"""
import collections
import csv
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
rev = sys.argv[3] if len(sys.argv) > 3 else "unknown"


def family(name):
    for f in ("k_conv", "k_gat", "k_gru", "k_rowgemm", "k_attend"):
        if f in name:
            return f
    return None


# --- kernel trace stats
rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(d, f"{tag}_kernel_trace_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline   (MI355X, 65536 windows/step)\n")
    f.write(f"{'kernel':62s} {'calls':>5s} {'total_ms':>9s} {'avg_ms':>8s} {'min_ms':>8s} {'max_ms':>8s} {'pct':>6s}\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if float(r["TotalDurationNs"]) / tot < 5e-4:
            continue
        f.write(f"{r['Name'][:62]:62s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e6:8.3f} "
                f"{float(r['MinNs'])/1e6:8.3f} {float(r['MaxNs'])/1e6:8.3f} {100*float(r['TotalDurationNs'])/tot:6.2f}\n")


def stats_table(csv_name, out_name, header):
    path = os.path.join(d, csv_name)
    if not os.path.exists(path):
        return
    rows_ = list(csv.DictReader(open(path)))
    tot_ = sum(float(r["TotalDurationNs"]) for r in rows_)
    with open(os.path.join(d, out_name), "w") as f:
        f.write(header + "\n")
        f.write(f"{'kernel':70s} {'calls':>5s} {'total_ms':>9s} {'avg_ms':>8s} {'pct':>6s}\n")
        for r in sorted(rows_, key=lambda r: -float(r["TotalDurationNs"])):
            if float(r["TotalDurationNs"]) / tot_ < 2e-3:
                continue
            f.write(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e6:8.3f} "
                    f"{100*float(r['TotalDurationNs'])/tot_:6.2f}\n")


stats_table("kernel_stats_bf16.csv", f"{tag}_kernel_trace_stats_bf16.txt",
            "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --precision bf16   (MI355X, 65536 windows/step)")
stats_table("kernel_stats_train256.csv", f"{tag}_kernel_trace_stats_train256.txt",
            "# rocprofv3 --kernel-trace --stats -- python profiles/train_step.py 256   (SMD shape, batch 256, 23 training steps)")


stats_table("kernel_stats_train8192.csv", f"{tag}_kernel_trace_stats_train8192.txt",
            "# rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 3 --warmup 1   (MSL shape, 8192 windows per step, 4 training steps)")
stats_table("kernel_stats_fwd256.csv", f"{tag}_kernel_trace_stats_fwd256.txt",
            "# rocprofv3 --kernel-trace --stats -- python profiles/forward_small.py 256   (MSL shape, 55 eval forwards of 256 windows)")
stats_table("kernel_stats_series.csv", f"{tag}_kernel_trace_stats_series.txt",
            "# rocprofv3 --kernel-trace --stats -- python profiles/series_bench.py   (MSL shape, score_series over 65 536 stride-1 windows, 8 calls)")
for extra, out in (("train_step_256.txt", None), ("forward_256.txt", None), ("train_bench_line.json", "train_bench_line_under_rocprof.json"),
                   ("train_bench_line_plain.json", None), ("series_65536.txt", None), ("batch_sweep.txt", None)):
    if os.path.exists(os.path.join(d, extra)):
        open(os.path.join(d, f"{tag}_{out or extra}"), "w").write(open(os.path.join(d, extra)).read())


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if family(k) is None:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            n[k] += 1
    return acc, n


# --- PMC summary + traffic
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 1 --warmup 1 --batch 65536; "
                     "sums over the 2 forward passes of that run, divided by 2*65536 windows",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 / windows; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                      "(gfx950 counts 64 B per 128-B read request)",
           "git": rev, "csrc_sha16": None, "bytes_per_window": {}, "raw_KiB_per_forward": {}}
# stamp with the content hash of the kernel sources this was collected on (bench.py only trusts a file whose stamp matches)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    import hashlib
    _c = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mtad-gat-pytorch_amd", "csrc")
    _h = hashlib.sha256()
    for _f in sorted(os.listdir(_c)):
        if _f.endswith((".hip", ".h", ".cpp")):
            _h.update(_f.encode())
            _h.update(open(os.path.join(_c, _f), "rb").read())
    traffic["csrc_sha16"] = _h.hexdigest()[:16]
except Exception:
    pass
with open(os.path.join(d, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <counters> (one pass per group), MI355X; per-dispatch averages.\n"
            "# FETCH_SIZE / WRITE_SIZE in KiB as reported (passes at --batch 65536, i.e. the large-batch kernels); SQ pass at --batch 65536.\n"
            "# MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128-B read request -> double it for bytes;\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (sum of 8 XCDs) in cycles.\n")
    fam_kib = collections.defaultdict(lambda: collections.defaultdict(float))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc, n = pmc(os.path.join(d, f"pmc_{c}.csv"))
        f.write(f"== pass {c}\n")
        for k, v in acc.items():
            f.write(f"{k[:56]:56s} n={n[k]:2d} {c}={v[c]/n[k]:.4g}\n")
            fam_kib[family(k)][c] += v[c] / 2.0          # 2 forward passes (warmup + step) in the run
    for fam, v in fam_kib.items():
        traffic["raw_KiB_per_forward"][fam] = dict(v)
        traffic["bytes_per_window"][fam] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 65536)
    acc, n = pmc(os.path.join(d, "pmc_SQ.csv"))
    f.write("== pass SQ\n")
    for k, v in acc.items():
        a = {c: x / n[k] for c, x in v.items()}
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        f.write(f"{k[:56]:56s} n={n[k]:2d} " + " ".join(f"{c}={x:.4g}" for c, x in sorted(a.items())) + "\n")
        f.write(f"{'':56s}      -> {cyc/1e6:.2f} Mcycles/launch, waves/CU {4*a['SQ_WAVE_CYCLES']/cyc/256:.1f}, "
                f"matrix pipe busy {100*a['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.0f} %, VALU instr/wave {a['SQ_INSTS_VALU']/a['SQ_WAVES']:.0f}\n")
json.dump(traffic, open(os.path.join(d, f"{tag}_traffic.json"), "w"), indent=1)
line = [l for l in open(os.path.join(d, "bench_line.json")) if l.startswith("{")][-1]
open(os.path.join(d, f"{tag}_bench_line.json"), "w").write(json.dumps(json.loads(line), indent=1) + "\n")
print(open(os.path.join(d, f"{tag}_kernel_trace_stats.txt")).read())
print(open(os.path.join(d, f"{tag}_pmc_summary.txt")).read())
print(json.dumps(traffic["bytes_per_window"]))
