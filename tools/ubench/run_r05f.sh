#!/bin/bash
# Round 5, f: the product with fft_quad_kernel as the 4096-point fast side kernel -- fused-chain suites, then bench.py against
# JST_FFT_KERNEL=pipe on the same box (driver form --steps 20 and the default form).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05f
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py tests/test_gpu_fast_provider.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_runtime.py tests/test_gpu_full_sizes.py -x -q 2>&1 | tail -8
for k in quad pipe quad pipe; do
  JST_FFT_KERNEL=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-host-fed > $O/bench20_$k.json 2> $O/bench20_$k.err
  echo "== $k steps20: $(summ $O/bench20_$k.json)"
done
for k in quad pipe; do
  JST_FFT_KERNEL=$k timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_$k.json 2> $O/bench_$k.err
  echo "== $k default: $(summ $O/bench_$k.json)"
done
tail -3 $O/*.err | head -30
