#!/bin/bash
# SQ counter passes over tools/conv_tune (GPU box). usage: tools/pmc_conv.sh <tag>
TAG=${1:-c0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
$ROOT/tools/conv_tune 10 > $OUT/timing.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -- $ROOT/tools/conv_tune 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- $ROOT/tools/conv_tune 2 > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p3 -- $ROOT/tools/conv_tune 2 > $OUT/p3.log 2>&1
cat $OUT/timing.txt
