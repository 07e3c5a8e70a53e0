#!/bin/bash
# Round 4, experiment a: the one-wavefront-per-transform 4096-point kernel (fft_wave.hh) against the pipelined one --
# correctness first (the fused-chain suites with JST_FFT_KERNEL=wave), then bench.py under both, batched and per cycle.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04a
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
echo "== tests with JST_FFT_KERNEL=wave"
JST_FFT_KERNEL=wave timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py tests/test_gpu_fast_provider.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_runtime.py -x -q 2>&1 | tail -15
echo "== new golden tests (default kernel)"
timeout 900 python -m pytest tests/test_gpu_reference_golden.py -q 2>&1 | tail -15
for k in pipe wave; do
  for mode in "" "--no-batch"; do
    JST_FFT_KERNEL=$k timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_${k}${mode}.json 2> $O/bench_${k}${mode}.err
    echo "== $k $mode: $(summ $O/bench_${k}${mode}.json)"
  done
done
cp cyberether_amd/lib/libjetstream_hip.so $O/base.so
for v in "$@"; do
  cp cyberether_amd/lib/variants/$v.so cyberether_amd/lib/libjetstream_hip.so
  for mode in "" "--no-batch"; do
    JST_FFT_KERNEL=wave timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_${v}${mode}.json 2> $O/bench_${v}${mode}.err
    echo "== $v wave $mode: $(summ $O/bench_${v}${mode}.json)"
  done
done
cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; rm -f $O/base.so
JST_FFT_KERNEL=pipe timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_pipe2.json 2> $O/bench_pipe2.err
echo "== pipe (again): $(summ $O/bench_pipe2.json)"
tail -3 $O/*.err | head -40
