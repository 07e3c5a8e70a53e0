#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03f
mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "fft or filter or full_sizes or reference or ring_producer or lineplot or surfaces" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
  JST_TILED_STATIC=0 timeout 300 python tools/bench_configs.py C3 C5 2>/dev/null | sed 's/^/generic /'
  JST_TILED_STATIC=1 timeout 300 python tools/bench_configs.py C3 C5 2>/dev/null | sed 's/^/static  /'
done > $O/configs_ab.log
python - <<'PY'
import json
for ln in open('gpurun_out/r03f/configs_ab.log'):
    tag, js = ln.split(' ',1)
    try:
        d=json.loads(js.strip()); print(tag, d['config'][:40], {k:d[k] for k in d if k in ('us_per_cycle','ms_per_cycle','value')}, d.get('roofline',{}).get('frac'), d.get('units_us') or d.get('units_ms'))
    except Exception as e: print(tag, js[:200])
PY
