#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into small tracked files under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table of the bench.py run
  <tag>_top_kernels.md     the same, top 25 by total time, readable
  <tag>_summary.json       bench line of that run + per-launch HBM traffic of the GAE scan from the
                           FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
scan_filter = "gae_columns_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_top_kernels.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats, bench.py run `{tag}` (top 25 by total time)\n\n")
    f.write("| % | calls | avg us | kernel |\n|---|---|---|---|\n")
    for r in rows[:25]:
        name = r["Name"].replace("ppo::(anonymous namespace)::", "").replace("void ", "")
        f.write("| %.2f | %d | %.1f | `%s` |\n" % (100 * float(r["TotalDurationNs"]) / total, int(r["Calls"]),
                                                  float(r["AverageNs"]) / 1e3, name[:120]))


def pmc(kind):
    fs = glob.glob(os.path.join(src, f"pmc_{kind}", "*", "*_counter_collection.csv"))
    if not fs:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if scan_filter in r["Kernel_Name"]]
    return sum(vals) / len(vals) if vals else None


bench = None
for line in open(os.path.join(src, "bench_trace.log")):
    if line.startswith("{"):
        bench = json.loads(line)
summary = {"tag": tag, "bench_line": bench}
krow = [r for r in rows if scan_filter in r["Name"]]
fetch_kib, write_kib = pmc("fetch"), pmc("write")
if krow and fetch_kib and write_kib:
    scan = (bench or {}).get("gae_scan", {}).get("bandwidth", {})
    summary["gae_scan_traffic"] = {
        "kernel": scan_filter, "calls": int(krow[0]["Calls"]), "avg_ns_rocprof": float(krow[0]["AverageNs"]),
        "avg_us_bench_hip_events": scan.get("avg_kernel_us"),
        "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib,
        "hbm_read_bytes_per_launch": 2 * fetch_kib * 1024, "hbm_write_bytes_per_launch": write_kib * 1024,
        "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
        "algorithmic_bytes_per_launch": 17 * scan.get("N", 256) * scan.get("A", 1 << 20)}
dom = (bench or {}).get("roofline", {})
if dom:
    key = dom.get("kernel", "").split("<")[0].split(" ")[0]
    match = [r for r in rows if key and key in r["Name"] and "16, 16, 42, 42" in r["Name"]]
    if match:
        summary["dominant_kernel_rocprof"] = {"name": match[0]["Name"][:140], "calls": int(match[0]["Calls"]),
                                              "avg_ns": float(match[0]["AverageNs"]),
                                              "bench_event_avg_ms_incl_reduce": dom.get("avg_kernel_ms")}
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
print("value:", bench and bench.get("value"), bench and bench.get("unit"))
