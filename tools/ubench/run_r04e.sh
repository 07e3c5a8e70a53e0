#!/bin/bash
# Round 4, experiment e: the binade hit update (hit_update.hh) against the additions written out (variant hits_seq), after the
# suites that cover the Spectrogram kernels; the wave kernel without scheduling barriers between its butterflies (wave_nosb).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04e
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), 'step_frac', round(d['roofline']['step_frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1; }
echo "== suites"
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_advice_r04.py tests/test_gpu_surfaces.py tests/test_gpu_chain.py tests/test_gpu_combine.py tests/test_gpu_random_sweep.py -q 2>&1 | tail -8
for mode in "" "--no-batch"; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_base$mode.json 2> $O/bench_base$mode.err
  echo "== binade $mode: $(summ $O/bench_base$mode.json)"
done
cp cyberether_amd/lib/libjetstream_hip.so $O/base.so
cp cyberether_amd/lib/variants/hits_seq.so cyberether_amd/lib/libjetstream_hip.so
for mode in "" "--no-batch"; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed $mode > $O/bench_seq$mode.json 2> $O/bench_seq$mode.err
  echo "== sequential $mode: $(summ $O/bench_seq$mode.json)"
done
cp cyberether_amd/lib/variants/wave_nosb.so cyberether_amd/lib/libjetstream_hip.so
JST_FFT_KERNEL=wave timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_wave_nosb.json 2> $O/bench_wave_nosb.err
echo "== wave_nosb: $(summ $O/bench_wave_nosb.json)"
cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; rm -f $O/base.so
JST_FFT_KERNEL=wave timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_wave.json 2> $O/bench_wave.err
echo "== wave: $(summ $O/bench_wave.json)"
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-host-fed > $O/bench_base2.json 2> $O/bench_base2.err
echo "== binade again: $(summ $O/bench_base2.json)"
