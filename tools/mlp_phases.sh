#!/bin/bash
# Where the time of mlp_rows_kernel goes: builds that leave the kernel after phase n (1 x tile, 2 fc1, 3 fc2, 4 heads,
# 5 loss, 6 heads backward), timed by tools/mlp_phases.py.  usage (GPU box): tools/mlp_phases.sh
ROOT=$(cd $(dirname $0)/.. && pwd)
python $ROOT/tools/mlp_phases.py
for n in 1 2 3 4 5 6; do
  PPO_AMD_LIB=$ROOT/ppo_amd/lib/libppo_amd_mlpstop$n.so python $ROOT/tools/mlp_phases.py
done
