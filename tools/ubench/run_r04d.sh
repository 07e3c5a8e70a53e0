#!/bin/bash
# Round 4, experiment d: the collision-free / one-barrier span Spectrogram kernel against round 3's (JST_SPEC_SPAN_V1=1),
# after the suites that cover it; plus the advisor regressions, the bench contract with the secondary configs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04d
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), 'step_frac', round(d['roofline']['step_frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
echo "== suites"
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_advice_r04.py tests/test_gpu_ring_producer.py tests/test_gpu_surfaces.py -q 2>&1 | tail -12
echo "== bench contract (2 ranks + secondary configs)"
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -12
for v in "" 1; do
  JST_SPEC_SPAN_V1=$v
  if [ -z "$v" ]; then unset JST_SPEC_SPAN_V1; else export JST_SPEC_SPAN_V1; fi
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_span_v1_$v.json 2> $O/bench_span_v1_$v.err
  echo "== span V1='$v': $(summ $O/bench_span_v1_$v.json)"
done
unset JST_SPEC_SPAN_V1
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_again.json 2> $O/bench_again.err
echo "== default again: $(summ $O/bench_again.json)"
timeout 600 python bench.py --no-cpu-baseline --no-host-fed > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d/bench_full.json').read().strip().splitlines()[-1])
print('full line:', round(d['value']), 'MS/s; generic per cycle', round(d['value_generic_per_cycle']['value']))
for c in d.get('configs', []):
    print('  ', c.get('config','')[:60], {k: (round(v,3) if isinstance(v,float) else v) for k,v in c.items() if k in ('ms_per_cycle','us_per_cycle','seconds','error','skipped')}, 'frac', round(c.get('roofline',{}).get('frac',0),3), 'parity', c.get('parity',{}).get('bit_exact'))
PY
