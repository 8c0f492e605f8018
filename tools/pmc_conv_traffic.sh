#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the convolution kernels at batch 256 via tools/conv_tune.
# usage (GPU box): tools/pmc_conv_traffic.sh <tag>
TAG=${1:-t0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ROOT/tools/conv_tune 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $ROOT/tools/conv_tune 2 > $OUT/write.log 2>&1
find $OUT -name "*counter_collection.csv" | head
