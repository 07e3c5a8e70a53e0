#!/bin/bash
# Round 5, w: span Spectrogram with 1024 / 512 / 256 threads per workgroup (JST_SPAN_THREADS): a 1024-thread workgroup per CU is
# 4096 wavefronts to start, ~12 us at the chip's ~340 wavefronts per microsecond.  Suites, then bench.py --steps 20 alternately.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05w
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,2), '| parity', d['parity']['bit_exact'])" 2>&1; }
{
for t in 512 256; do JST_SPAN_THREADS=$t timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_quad_kernel.py -x -q 2>&1 | tail -1; done
for rep in 1 2 3; do
  for t in 1024 512 256; do
    JST_SPAN_THREADS=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-host-fed --no-configs > $O/b_$t.json 2> $O/b_$t.err
    echo "== threads $t: $(summ $O/b_$t.json)"
  done
done
} 2>&1 | tee $O/log.txt
