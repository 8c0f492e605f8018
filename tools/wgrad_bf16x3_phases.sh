#!/bin/bash
# Runs on the GPU box: launch times of the split-bf16 weight-gradient kernel with phases compiled out
# (libppo_amd_w3skip<bits>.so built by tools/build_variant.sh w3skip<bits> wgrad_bf16x3.hip -DPPO_TUNE_W3_SKIP=<bits>;
# bit 0 no global loads, bit 1 no LDS stores, bit 2 no K loop).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== full"; python3 $ROOT/tools/wgrad_bf16x3_speed.py
for v in 1 2 4 6 3; do
  echo "== skip bits $v"; PPO_AMD_LIB=$ROOT/ppo_amd/lib/libppo_amd_w3skip$v.so python3 $ROOT/tools/wgrad_bf16x3_speed.py
done
