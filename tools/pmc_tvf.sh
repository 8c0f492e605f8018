#!/bin/bash
# Runs on the GPU box: HBM traffic of ppo_tvf_returns_f32 at SURVEY.md §8(d)'s size (N = A = 256, K = V = 108) — FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes over `bench.py --tvf-only`, plus a kernel-trace pass for the durations.
# usage: tools/pmc_tvf.sh <tag> [heads]   -> gpurun_out/pmct_<tag>/...; then python3 tools/pmc_tvf_table.py <tag>
TAG=${1:-r03}; H=${2:-108}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rc=0
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --tvf-only --tvf-heads $H > $OUT/$c.log 2>&1
  r=$?; echo "$C pass exit $r"; tail -1 $OUT/$c.log | cut -c1-300
  [ $r -ne 0 ] && rc=$r && break
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --tvf-only --tvf-heads $H > $OUT/trace.log 2>&1
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_hash())" > $OUT/kernel_source_sha16.txt
find $OUT -name "*.db" -delete
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
exit $rc
