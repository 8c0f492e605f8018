#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the LDS-resident stack kernels at batch 256 via tools/stack_tail_probe.py.
# usage (GPU box): tools/pmc_stack_tail.sh <tag>
TAG=${1:-t0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcs_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/stack_tail_probe.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/stack_tail_probe.py > $OUT/write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stack_tail" in r["Kernel_Name"] or "stack_full" in r["Kernel_Name"]:
            geom = r["Kernel_Name"].split("::")[-1].split("(")[0].replace(" ", "")
            acc[geom][r["Counter_Name"]].append(float(r["Counter_Value"]))
for geom, c in sorted(acc.items()):
    print(geom, {k: (len(v), sum(v) / len(v)) for k, v in c.items()})
PY
