#!/bin/bash
# Round 5, m: span Spectrogram with the two histogram copies interleaved (U32[index][copy][16]) against the separate histograms
# (cyberether_amd/lib/variants/span_separate.so), same box: spectrogram suites first, then bench.py alternately.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05m
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,2), '| parity', d['parity']['bit_exact'])" 2>&1; }
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_quad_kernel.py tests/test_gpu_surfaces.py -x -q 2>&1 | tail -3
cp cyberether_amd/lib/libjetstream_hip.so $O/base.so
for rep in 1 2 3; do
  for v in interleaved separate; do
    if [ $v = separate ]; then cp cyberether_amd/lib/variants/span_separate.so cyberether_amd/lib/libjetstream_hip.so; else cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-host-fed --no-configs > $O/b_$v.json 2> $O/b_$v.err
    echo "== $v: $(summ $O/b_$v.json)"
  done
done
cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; rm -f $O/base.so
