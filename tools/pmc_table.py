#!/usr/bin/env python3
"""Tabulate rocprofv3 --pmc passes of tools/pmc_conv.sh: per kernel, average counter per dispatch."""
import csv, glob, sys, collections, re
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv3x3" not in n: continue
        m = re.search(r"(conv3x3\w*)<([^>]*)>", n)
        key = (m.group(1).replace("conv3x3_", "") + "<" + m.group(2).replace(" ", "") + ">") if m else n[:40]
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]
print("kernel".ljust(44), " ".join(c.replace("SQ_", "")[:13].rjust(13) for c in cols))
for k in sorted(vals):
    print(k.ljust(44), " ".join(("%.3g" % (sum(vals[k][c]) / len(vals[k][c]))).rjust(13) if vals[k][c] else "-".rjust(13) for c in cols))
