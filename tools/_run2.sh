set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04q_tests.log 2>&1 || { tail -40 gpurun_out/r04q_tests.log; exit 1; }
tail -2 gpurun_out/r04q_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r04q_bench.json 2> gpurun_out/r04q_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for r in d['roofline_by_kernel']['rows']:
    if 'dense' in r['kernel'] or '4->16' in r['kernel'] or 'pool_forward' in r['kernel']: print(r['kernel'], r['launches'], r['avg_us'])
PY
