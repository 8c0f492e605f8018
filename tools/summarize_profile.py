#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into the small, tracked
profiles/<tag>_* files: the rocprofv3 --stats kernel table, the bench line of the same run and
the per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel_filter = sys.argv[2] if len(sys.argv) > 2 else "gae_columns_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
krow = [r for r in rows if kernel_filter in r["Name"]][0]


def pmc(kind):
    f = glob.glob(os.path.join(src, f"pmc_{kind}", "*", "*_counter_collection.csv"))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel_filter in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


fetch_kib, nf = pmc("fetch")
write_kib, nw = pmc("write")
bench = None
for line in open(os.path.join(src, "bench_trace.log")):
    if line.startswith("{"):
        bench = json.loads(line)
summary = {
    "tag": tag,
    "kernel": krow["Name"].split("(")[0],
    "calls": int(krow["Calls"]),
    "avg_ns_rocprof": float(krow["AverageNs"]),
    "avg_ms_bench_hip_events": bench["roofline"]["avg_kernel_ms"] if bench else None,
    "FETCH_SIZE_KiB_avg": fetch_kib,
    "WRITE_SIZE_KiB_avg": write_kib,
    "hbm_read_bytes_per_launch": 2 * fetch_kib * 1024,  # gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads
    "hbm_write_bytes_per_launch": write_kib * 1024,
    "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None,
    "workload": bench["config"]["workload"] if bench else None,
    "bench_line": bench,
}
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
