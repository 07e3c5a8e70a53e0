ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06h; mkdir -p $O
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_tiled_persistent.py -q -m gpu -x 2>&1 | tail -15
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06h/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), 'MS/s', d['roofline']['frac'], d['roofline']['step_frac'])
for c in d['configs']: print(json.dumps(c)[:1800])
PY
