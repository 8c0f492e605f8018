#!/usr/bin/env python3
"""Turn one verification run (tools/profile_bench.sh <tag>, tools/pmc_bench.sh <tag>, bench.py > gpurun_out/bench_<tag>.log)
into the tracked files under profiles/: kernel stats + top kernels + summary (summarize_profile.py), HBM traffic table
(pmc_bench_table.py, also copied to profiles/bench_hbm_traffic.* which bench.py reads), the bench line and the
per-kernel roofline table.  Usage: python tools/round_artifacts.py <tag>"""
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_profile.py"), tag], check=True, stdout=subprocess.DEVNULL)
subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_bench_table.py"), tag], check=True, stdout=subprocess.DEVNULL)
if os.path.isdir(os.path.join(root, "gpurun_out", f"pmct_{tag}")):
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_tvf_table.py"), tag], check=True, stdout=subprocess.DEVNULL)
for ext in ("json", "md"):
    shutil.copy(os.path.join(prof, f"{tag}_bench_hbm_traffic.{ext}"), os.path.join(prof, f"bench_hbm_traffic.{ext}"))
line = [l for l in open(os.path.join(root, "gpurun_out", f"bench_{tag}.log")) if l.startswith("{")][-1]
d = json.loads(line)
json.dump(d, open(os.path.join(prof, f"{tag}_bench_line.json"), "w"), indent=1)
t = d["roofline_by_kernel"]
with open(os.path.join(prof, f"{tag}_roofline_by_kernel.md"), "w") as f:
    f.write(f"# Time-weighted roofline, bench.py run `{tag}` (the `roofline_by_kernel` object of the bench line)\n\n")
    f.write(f"value {d['value']} env-steps/s, {d['ms_per_step']} ms per PPO iteration; one extra iteration of the same run with "
            f"every C-ABI launch\nbracketed by HIP events on its launch stream: {t['kernel_ms_total']} ms of kernel time.  MFMA "
            f"kernels: {t['mfma_kernels']['share'] * 100:.1f} % of it at\n{t['mfma_kernels']['time_weighted_frac_of_peak']} of the "
            f"fp32 MFMA peak (157.3 TFLOP/s) time-weighted; whole step incl. rollout host waits: {d['model_tflops_whole_step']} "
            f"TFLOP/s of model FLOPs = {d['model_frac_of_mfma_peak_whole_step']} of peak.\n\n")
    f.write("| kernel class | share of kernel time | launches | avg µs | achieved | fraction of its peak |\n|---|---|---|---|---|---|\n")
    for r in t["rows"]:
        ach = f"{r['achieved']} {r['unit']}" if "achieved" in r else "latency-bound helper"
        f.write(f"| {r['kernel']} | {r['share'] * 100:.1f} % | {r['launches']} | {r['avg_us']} | {ach} | {r.get('frac', '–')} |\n")
print(d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"], "whole step",
      d["model_frac_of_mfma_peak_whole_step"], "time-weighted", t["mfma_kernels"]["time_weighted_frac_of_peak"],
      "kernel ms", t["kernel_ms_total"], "cpu", d["cpu_baseline"]["value"], "scan", d["gae_scan"]["bandwidth"]["achieved_GBps"])
