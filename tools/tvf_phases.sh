#!/bin/bash
# Where the time of tvf_column_kernel goes: the same launch with one phase switched off at a time
# (PPO_AMD_TVF_SKIP bit 0 = walk, 1 = terms, 2 = stores; the results of those runs are garbage by design).
# usage (GPU box): tools/tvf_phases.sh [heads]
H=${1:-108}
ROOT=$(cd $(dirname $0)/.. && pwd)
export PPO_AMD_LIB=$($ROOT/tools/build_variant.sh timing_aids tvf_returns.hip -DPPO_TUNE_TIMING_AIDS | tail -1)
for s in 0 1 2 4 7; do
  echo -n "skip=$s "
  PPO_AMD_TVF_SKIP=$s python bench.py --tvf-only --tvf-heads $H | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['tvf_returns']; print(d['avg_kernel_us'], 'us', 'bit_exact' if d['bit_exact_vs_oracle'] else '(garbage, as intended)')"
done
