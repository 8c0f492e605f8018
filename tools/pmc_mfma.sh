#!/bin/bash
# Runs on the GPU box: MFMA-pipe utilisation of the kernels of the real bench.py iteration.
#   pass 1  rocprofv3 --pmc <SQ counters>   (counters only; bench with an 8-step rollout: < 16 k packets per queue,
#           profiles/r02a_pmc_segv_analysis.md)
#   pass 2  rocprofv3 --kernel-trace        (durations of the same command, no counters)
# usage: tools/pmc_mfma.sh <tag>   -> gpurun_out/pmcm_<tag>/{sq,trace}/...; then: python3 tools/pmc_mfma_table.py <tag>
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--n-steps 8 --steps 1 --warmup 1 --no-cpu-baseline --no-scan --no-kernel-table"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU \
  --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
r=$?; echo "SQ pass exit $r"; tail -2 $OUT/sq.log
[ $r -ne 0 ] && exit $r
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
r=$?; echo "trace pass exit $r"; tail -2 $OUT/trace.log
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_hash())" > $OUT/kernel_source_sha16.txt
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" | head
exit $r
