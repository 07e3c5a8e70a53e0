#!/bin/bash
# Round 5, l: the driver-form bench line with every secondary leg (new: cpu_baseline reference-runtime, configs[2].fast, configs[3].stations_64,
# alt_eager_spans), then the new GPU tests.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05l
mkdir -p $O
cd $ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05l/bench_driver_form.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', round(d['ms_per_step']*1e3,2), 'frac', round(d['roofline']['frac'],3), 'frac_rocprof', d['roofline'].get('frac_rocprof'), 'step_frac', round(d['roofline']['step_frac'],3), 'parity', d['parity'].get('bit_exact'))
print('cpu_baseline', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','kind','cores')}, 'dense', d['cpu_baseline'].get('dense_port',{}).get('value'))
print('alt_eager', d.get('alt_eager_spans'))
for c in d.get('configs', []):
    print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('frac','ms_per_cycle','bit_exact','max_err_rel_peak','within_1e-5','stations_x_realtime','roofline','parity')}) for k,v in c.items() if k not in ('units',)})[:900])
PY
tail -5 $O/bench_driver_form.err
timeout 600 python -m pytest tests/test_gpu_exact_sweep.py::test_libm_pinned_on_this_box tests/test_gpu_filter_fast.py -q 2>&1 | tail -3
