#!/bin/bash
# Round 5, j: the product with fft_quad_kernel + claimed last rounds: fused-chain suites, then bench.py (driver form and default)
# against JST_QUAD_STATIC=1 and JST_FFT_KERNEL=pipe on the same box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05j
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py tests/test_gpu_fast_provider.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_runtime.py tests/test_gpu_full_sizes.py -x -q 2>&1 | tail -4
for rep in 1 2; do
for k in dyn static pipe; do
  E=""; [ $k = static ] && E="JST_QUAD_STATIC=1"; [ $k = pipe ] && E="JST_FFT_KERNEL=pipe"
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-host-fed > $O/bench20_$k.json 2> $O/bench20_$k.err
  echo "== $k steps20: $(summ $O/bench20_$k.json)"
done
done
for k in dyn static pipe; do
  E=""; [ $k = static ] && E="JST_QUAD_STATIC=1"; [ $k = pipe ] && E="JST_FFT_KERNEL=pipe"
  env $E timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_$k.json 2> $O/bench_$k.err
  echo "== $k default: $(summ $O/bench_$k.json)"
done
