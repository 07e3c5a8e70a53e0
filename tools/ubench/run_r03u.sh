#!/bin/bash
# Round 3, experiment u: side-output layout (tile-major / row-major) x cache policies of the fused kernel's stream loads and
# F32 stores under cycle batching (variants built by build_variant.sh with VARIANT_UNITS="fft_side spectrogram")
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03u
mkdir -p $O
cp $ROOT/cyberether_amd/lib/libjetstream_hip.so $O/base.so
run() {
  name=$1
  timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-host-fed > $O/$name.json 2> $O/$name.err
  echo "== $name: $(python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
u=d['config']['units_ms']; a=d['alt_per_cycle_launch']; au=a['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | span fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| per cycle:', round(a['value']), 'MS/s fused', round(au['spectrum_fused']*1e3,2), 'spectrogram', round(au['spectrogram']*1e3,2), '| parity', d['parity']['bit_exact'], a['parity']['bit_exact'])" 2>&1)"
}
run base
for v in "$@"; do
  cp $ROOT/cyberether_amd/lib/variants/$v.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
  run $v
done
cp $O/base.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
run base2
rm -f $O/base.so
