#!/bin/bash
# Round 3, experiment v: the fused side kernel's A/B switches re-measured under CYCLE BATCHING (16384 transforms per launch:
# ramp, cold start and tail amortised -- the steady state is what counts now); variants: build_variant.sh, VARIANT_UNITS=fft_side
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03v
mkdir -p $O
cp $ROOT/cyberether_amd/lib/libjetstream_hip.so $O/base.so
run() {
  name=$1
  for prov in ${PROVS:-fast generic}; do
  timeout 300 python $ROOT/bench.py --provider $prov --no-cpu-baseline --no-alt > $O/${name}_$prov.json 2> $O/${name}_$prov.err
  echo "== $name $prov: $(python -c "
import json; d=json.loads(open('$O/${name}_$prov.json').read().strip().splitlines()[-1])
u=d['config']['units_ms']
print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step | span fused', round(u['spectrum_fused']*1e3,1), 'spectrogram', round(u['spectrogram']*1e3,1), '| frac', round(d['roofline']['frac'],3), '| parity', d['parity']['bit_exact'])" 2>&1)"
  done
}
run base
for v in "$@"; do
  cp $ROOT/cyberether_amd/lib/variants/$v.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
  run $v
done
cp $O/base.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
run base2
rm -f $O/base.so
