#!/bin/bash
# Round 5, r: span Spectrogram variants by JST_SPAN_KERNEL (1 = default: cycles counted in pairs; 2 = the round-3 kernel; 3 = the
# overlapped two-histogram experiment spectrogram_index_span3_kernel): suites first, then bench.py --steps 20
# --warmup 5 alternately, same box.  Edit the `for v in` list to pick the pair.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05r
mkdir -p $O
cd $ROOT
summ() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,2), '| parity', d['parity']['bit_exact'])" 2>&1; }
{
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spectrogram_indices.py tests/test_gpu_quad_kernel.py tests/test_gpu_surfaces.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 1 2; do
    JST_SPAN_KERNEL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-host-fed --no-configs > $O/b_$v.json 2> $O/b_$v.err
    echo "== span kernel $v: $(summ $O/b_$v.json)"
  done
done
} 2>&1 | tee $O/log.txt
