#!/bin/bash
# Runs on the GPU box: everything tools/round_artifacts.py <tag> turns into tracked files, in three parts that each fit one
# gpurun call.  usage: tools/round_final.sh <tag> 1|2|3
TAG=${1:-r04}; PART=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
case $PART in
1)  # the bench line (with its opt-in leg and CPU baseline) and the rocprofv3 kernel-trace stats of the same command
    timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
    tail -c 400 gpurun_out/bench_$TAG.log; echo
    bash tools/profile_bench.sh $TAG --steps 5 --warmup 2 --no-cpu-baseline || exit 1 ;;
2)  # counters: HBM traffic of the bench iteration, MFMA-pipe counters, TVF traffic (separate --pmc passes)
    bash tools/pmc_bench.sh $TAG || exit 1
    bash tools/pmc_mfma.sh $TAG || exit 1
    bash tools/pmc_tvf.sh $TAG || exit 1 ;;
3)  # the other BASELINE configs and the opt-in precision mode
    for spec in "procgen:--config procgen" "humanoid:--config humanoid_tvf" "medium:--precision medium" "medium_procgen:--config procgen --precision medium"; do
        name=${spec%%:*}; args=${spec#*:}
        timeout -k 10 400 python bench.py $args > gpurun_out/bench_${TAG}_$name.log 2> gpurun_out/bench_${TAG}_$name.err || { echo "$name failed"; tail -5 gpurun_out/bench_${TAG}_$name.err; exit 1; }
        python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/bench_${TAG}_$name.log') if l.startswith('{')][-1]); print('$name', d['value'], d['ms_per_step'], d.get('dtype'))"
    done ;;
esac
