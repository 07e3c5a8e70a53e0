# the driver's invocation of bench.py, timed, with the line's key figures printed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06z
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06z/bench.json 2> gpurun_out/r06z/bench.err
t1=$(date +%s.%N); echo "bench.py wall: $(echo "$t1 - $t0" | bc) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06z/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['unit'], round(d['ms_per_step']*1e3,3), 'us/step frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4), 'parity', d['parity']['bit_exact'], 'cpu', d['cpu_baseline']['value'])
print({k: round(v['us_per_cycle'],2) for k,v in d['reference_driven'].items() if isinstance(v,dict) and 'us_per_cycle' in v})
for c in d['configs']: print(c['config'][:50], c.get('ms_per_cycle', c.get('us_per_cycle')), c.get('parity',{}).get('bit_exact'), json.dumps(c.get('provider_fast',{}))[:500])
PY
