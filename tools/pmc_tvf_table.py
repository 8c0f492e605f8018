#!/usr/bin/env python3
"""Tabulate tools/pmc_tvf.sh: HBM bytes per launch of the TVF returns kernels, (2 x FETCH_SIZE + WRITE_SIZE) KiB as the
guide prescribes for gfx950, beside the algorithmic bytes and the kernel-trace durations.  Writes
profiles/<tag>_tvf_hbm_traffic.md."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmct_{tag}")


def load(kind):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\s+", "", r["Kernel_Name"].replace("ppo::(anonymous namespace)::", "").replace("void ", ""))
            if name.startswith("tvf_"):
                acc[name.split("(")[0]].append(float(r["Counter_Value"]))
    return acc


fetch, write = load("fetch"), load("write")
dur = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\s+", "", r["Name"].replace("ppo::(anonymous namespace)::", "").replace("void ", "")).split("(")[0]
        if name.startswith("tvf_"):
            dur[name] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
line = json.loads([ln for ln in open(os.path.join(src, "fetch.log")) if ln.startswith("{")][-1])["tvf_returns"]
sha = open(os.path.join(src, "kernel_source_sha16.txt")).read().strip()
with open(os.path.join(root, "profiles", f"{tag}_tvf_hbm_traffic.md"), "w") as out:
    out.write(f"# TVF returns (`ppo_tvf_returns_f32`) HBM traffic per launch (`{tag}`, kernel sources {sha})\n\n"
              f"`bench.py --tvf-only`: N = {line['N']}, A = {line['A']}, K = V = {line['K']}, {line['samples_per_head']} samples per head, "
              f"{line['distinct_n']} distinct n (max {line['max_n']}); algorithmic bytes = 4 (N+1) A V + 4 N A K = "
              f"{line['algorithmic_bytes'] / 1e6:.1f} MB.  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
              "(2 x FETCH + WRITE) KiB; durations from a --kernel-trace --stats pass of the same command.\n\n"
              "| kernel | launches | FETCH KiB | WRITE KiB | MB per launch | x algorithmic | avg us (trace) |\n|---|---|---|---|---|---|---|\n")
    for k in sorted(set(fetch) & set(write)):
        f_, w_ = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        mb = (2 * f_ + w_) * 1024 / 1e6
        out.write(f"| `{k}` | {len(fetch[k])} | {f_:.1f} | {w_:.1f} | {mb:.2f} | {mb * 1e6 / line['algorithmic_bytes']:.2f} | "
                  f"{dur.get(k, (0, float('nan')))[1]:.1f} |\n")
    out.write(f"\nbench line of the FETCH pass (under the profiler): avg {line['avg_kernel_us']} us per call (pack + column kernel), "
              f"bit_exact_vs_oracle {line['bit_exact_vs_oracle']}\n")
print(open(os.path.join(root, "profiles", f"{tag}_tvf_hbm_traffic.md")).read())
