#!/bin/bash
# Round 4, experiment b: the class-aware fast epilogue (JST_EPI_V2) -- the guard's proof by exhaustion first, then the
# fast-provider / chain suites under both kernels, then bench.py: v2 (pipe, wave), the round-3 epilogue (variant epi_v1), v2 again.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04b
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
echo "== exhaustive sweeps + fast provider tests"
timeout 1200 python -m pytest tests/test_gpu_exact_sweep.py tests/test_gpu_fast_provider.py -x -q 2>&1 | tail -12
echo "== chain suites, pipe kernel"
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_reference_golden.py -x -q 2>&1 | tail -5
echo "== chain suites, wave kernel"
JST_FFT_KERNEL=wave timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_batch.py tests/test_gpu_fast_provider.py tests/test_gpu_spectrogram_indices.py -x -q 2>&1 | tail -5
for k in pipe wave; do
  for mode in "" "--no-batch"; do
    JST_FFT_KERNEL=$k timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_${k}${mode}.json 2> $O/bench_${k}${mode}.err
    echo "== v2 $k $mode: $(summ $O/bench_${k}${mode}.json)"
  done
done
cp cyberether_amd/lib/libjetstream_hip.so $O/base.so
for v in "$@"; do
  cp cyberether_amd/lib/variants/$v.so cyberether_amd/lib/libjetstream_hip.so
  for mode in "" "--no-batch"; do
    JST_FFT_KERNEL=pipe timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_${v}${mode}.json 2> $O/bench_${v}${mode}.err
    echo "== $v pipe $mode: $(summ $O/bench_${v}${mode}.json)"
  done
done
cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; rm -f $O/base.so
JST_FFT_KERNEL=pipe timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_pipe2.json 2> $O/bench_pipe2.err
echo "== v2 pipe (again): $(summ $O/bench_pipe2.json)"
JST_FFT_KERNEL=pipe timeout 300 python bench.py --provider generic --no-cpu-baseline --no-alt --no-host-fed > $O/bench_generic.json 2> $O/bench_generic.err
echo "== generic pipe: $(summ $O/bench_generic.json)"
