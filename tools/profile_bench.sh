#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes around bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py "$@" > $OUT/bench_trace.log 2>&1 || { tail -20 $OUT/bench_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/bench_fetch.log 2>&1 || { tail -20 $OUT/bench_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/bench_write.log 2>&1 || { tail -20 $OUT/bench_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
# drop bulky per-dispatch traces beyond what the summaries need
find $OUT -name "*.db" -delete
