#!/usr/bin/env python3
"""Tabulate tools/pmc_mfma.sh: per kernel (name + grid) of one real bench.py iteration, the share of the kernel's
duration in which a SIMD's MFMA pipe was busy, next to the instruction mix.

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / n_simds / (duration x shader clock)
(the counter sums over the chip's 1024 SIMDs; the clock is taken as SQ_BUSY_CYCLES-independent 2.4 GHz nominal, so the
figure is a lower bound when the clock droops).  frac_of_peak = MFMA instructions x 2048 FLOP (v_mfma_f32_16x16x4_f32) /
duration / 157.3 TFLOP/s, i.e. the roofline fraction the counters themselves imply.
Writes profiles/<tag>_mfma_utilisation.md."""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmcm_{tag}")
N_SIMD, CLOCK, PEAK = 1024, 2.4e9, 157.3e12


def norm(name):
    name = name.replace("ppo::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\s+", "", name)


counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        counters[(norm(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        grid = int(r.get("Grid_Size", 0) or 0) or int(r.get("Grid_Size_X", 0) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
        durs[(norm(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)

rows = []
for key, c in counters.items():
    if key not in durs or "SQ_INSTS_MFMA" not in c:
        continue
    mean = {k: sum(v) / len(v) for k, v in c.items()}
    d = sorted(durs[key])[len(durs[key]) // 2]  # median duration of the launches of this geometry
    if mean["SQ_INSTS_MFMA"] < 1:
        continue
    rows.append({
        "kernel": key[0], "grid": key[1], "launches": len(durs[key]), "us": d * 1e6,
        "share": d * len(durs[key]),
        "mfma_busy": mean["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / (d * CLOCK),
        "cycles_per_mfma": mean["SQ_VALU_MFMA_BUSY_CYCLES"] / mean["SQ_INSTS_MFMA"],
        # (the opt-in split kernels issue v_mfma_f32_16x16x32_bf16: 16 384 FLOP in 16 cycles, priced against the dense bf16 peak)
        "frac_of_peak": (mean["SQ_INSTS_MFMA"] * 16384 / d / 2516.6e12 if "bf16x3" in key[0]
                         else mean["SQ_INSTS_MFMA"] * 2048 * 64 / 64 / d / PEAK),
        "wait_inst_any": mean["SQ_WAIT_INST_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1),
        "wait_any": mean["SQ_WAIT_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1),
        "valu_per_mfma": mean["SQ_INSTS_VALU"] / mean["SQ_INSTS_MFMA"],
        "lds_per_mfma": mean["SQ_INSTS_LDS"] / mean["SQ_INSTS_MFMA"],
    })
rows.sort(key=lambda r: -r["share"])
total = sum(r["share"] for r in rows)
sha = os.path.join(src, "kernel_source_sha16.txt")
with open(os.path.join(root, "profiles", f"{tag}_mfma_utilisation.md"), "w") as f:
    f.write(f"# MFMA-pipe utilisation per kernel over one real bench.py iteration (`{tag}`)\n\n"
            "rocprofv3 --pmc SQ_* (one pass, counters only) + a separate --kernel-trace pass of the same command "
            "(`bench.py --n-steps 8 --steps 1 --warmup 1`: every training kernel at its minibatch geometry, rollout kernels at "
            "128 images); tools/pmc_mfma.sh, tools/pmc_mfma_table.py.  kernel sources "
            f"`{open(sha).read().strip() if os.path.exists(sha) else '?'}`.\n\n"
            "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (median duration x 2.4 GHz): the share of the launch in which "
            "a SIMD's matrix pipe is executing; `frac` = MFMA instructions x 2048 FLOP / duration / 157.3 TFLOP/s (`*_bf16x3_*` kernels of the opt-in precision leg: x 16 384 FLOP / 2 516.6 TFLOP/s dense bf16; includes "
            "padding work the algorithmic count in roofline_by_kernel leaves out); `wait inst` / `wait any` = share of wave "
            "cycles waiting to issue / waiting for anything; VALU (incl. MFMA) and LDS instructions per MFMA.\n\n"
            "| kernel | grid | launches | us | share of MFMA-kernel time | MFMA busy | frac | cycles / MFMA | wait inst | wait any | VALU / MFMA | LDS / MFMA |\n"
            "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows[:28]:
        f.write("| `%s` | %d | %d | %.1f | %.1f %% | %.2f | %.2f | %.1f | %.2f | %.2f | %.1f | %.2f |\n" % (
            r["kernel"][:90], r["grid"], r["launches"], r["us"], 100 * r["share"] / total, r["mfma_busy"], r["frac_of_peak"],
            r["cycles_per_mfma"], r["wait_inst_any"], r["wait_any"], r["valu_per_mfma"], r["lds_per_mfma"]))
print(open(os.path.join(root, "profiles", f"{tag}_mfma_utilisation.md")).read())
