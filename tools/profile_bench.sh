#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats around bench.py, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) around the GAE-scan section only.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py "$@" > $OUT/bench_trace.log 2>&1 || { tail -20 $OUT/bench_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --scan-only > $OUT/bench_fetch.log 2>&1 || { tail -20 $OUT/bench_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --scan-only > $OUT/bench_write.log 2>&1 || { tail -20 $OUT/bench_write.log; exit 1; }
# the per-dispatch trace of a full PPO iteration is tens of MB: keep only the stats tables
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" | head -20
