#!/usr/bin/env python3
"""Tabulate the two rocprofv3 --pmc passes of tools/pmc_bench.sh (FETCH_SIZE, WRITE_SIZE over one real bench.py
iteration): HBM bytes per launch for every kernel, per launch geometry.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports half of the bytes of a wide
coalesced read, so bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  Writes profiles/<tag>_bench_hbm_traffic.json (read
by bench.py for `roofline.traffic`) and a readable .md next to it."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmcb_{tag}")


def load(kind):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("ppo::(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\s+", "", name)
            acc[(name, int(r.get("Grid_Size", 0) or 0))].append(float(r["Counter_Value"]))
    return acc


fetch, write = load("fetch"), load("write")
rows = []
for key in sorted(set(fetch) & set(write)):
    f, w = fetch[key], write[key]
    rows.append({"kernel": key[0], "grid_size": key[1], "launches": len(f), "FETCH_SIZE_KiB": sum(f) / len(f),
                 "WRITE_SIZE_KiB": sum(w) / len(w),
                 "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024})
rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
how = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --n-steps 8 --steps 1 --warmup 1` (tools/pmc_bench.sh: the bench iteration with a short rollout, every training kernel at its minibatch geometry), "
       "average over the training-minibatch launches, (2 x FETCH + WRITE) KiB")
sha_file = os.path.join(src, "kernel_source_sha16.txt")
out = {"tag": tag, "kernel_source_sha16": open(sha_file).read().strip() if os.path.exists(sha_file) else None,
       "rows": rows[:60]}
for short in ("stack_full_kernel", "stack_tail_kernel", "conv3x3_kernel"):
    cand = [r for r in rows if r["kernel"].startswith(short)]
    if cand:
        best = max(cand, key=lambda r: (r["grid_size"], r["launches"]))  # the minibatch-sized launches have the larger grid
        out[short] = {"kernel": best["kernel"], "grid_size": best["grid_size"], "launches": best["launches"],
                      "hbm_bytes_per_launch": best["hbm_bytes_per_launch"], "how": how}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_bench_hbm_traffic.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", f"{tag}_bench_hbm_traffic.md"), "w") as f:
    f.write(f"# HBM traffic per launch over one real bench.py iteration (`{tag}`)\n\n{how}\n\n")
    f.write("| kernel | grid | launches | FETCH KiB | WRITE KiB | MB per launch |\n|---|---|---|---|---|---|\n")
    for r in rows[:40]:
        f.write("| `%s` | %d | %d | %.1f | %.1f | %.2f |\n" % (r["kernel"][:110], r["grid_size"], r["launches"],
                                                               r["FETCH_SIZE_KiB"], r["WRITE_SIZE_KiB"],
                                                               r["hbm_bytes_per_launch"] / 1e6))
print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))
