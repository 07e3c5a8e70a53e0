#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03c
mkdir -p $O
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03c/bench.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['step_frac'])
    print('parity', d['parity']); print('alt', d.get('alt_provider')); print('host_fed', json.dumps(d.get('host_fed'))[:1500]); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['configs0'])
except Exception as e: print('bench parse failed', e)
PY
tail -5 $O/bench.err
