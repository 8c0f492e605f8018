#!/bin/bash
# Runs on the GPU box: HBM traffic of the real bench.py iteration — FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc
# passes (counters only: no --kernel-trace / --stats beside --pmc), one PPO iteration each.
# usage: tools/pmc_bench.sh <tag>      -> gpurun_out/pmcb_<tag>/{fetch,write}/...counter_collection.csv + maps_*.txt
# then (anywhere): python3 tools/pmc_bench_table.py <tag>  -> profiles/<tag>_bench_hbm_traffic.{json,md}
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcb_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
# --n-steps 8: the same iteration with a short rollout.  rocprofv3's counter-collection interceptor segfaults once a queue has
# carried ~16 k AQL packets (profiles/r02a_pmc_segv_analysis.md); the training kernels keep their minibatch geometry.
ARGS="--n-steps 8 --steps 1 --warmup 1 --no-cpu-baseline --no-scan --no-kernel-table"
rc=0
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | sed 's/_size//')
  PPO_AMD_DUMP_MAPS=$OUT/maps_$c.txt rocprofv3 --pmc $C --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py $ARGS > $OUT/$c.log 2>&1
  r=$?; echo "$C pass exit $r"; tail -3 $OUT/$c.log
  [ $r -ne 0 ] && rc=$r && break
done
# which kernel sources the counters belong to: bench.py reports the traffic only while they are unchanged
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_hash())" > $OUT/kernel_source_sha16.txt
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" | head
exit $rc
