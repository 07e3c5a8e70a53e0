#!/bin/bash
# Round 3, experiment r: cache policies of the fused kernel's stream loads / stores (variants built by build_variant.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03r
mkdir -p $O
cp $ROOT/cyberether_amd/lib/libjetstream_hip.so $O/base.so
cd /tmp && export TMPDIR=/tmp
run() {  # name
  name=$1
  for prov in fast generic; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_${prov}_trace -- \
      python $ROOT/bench.py --provider $prov --no-cpu-baseline --no-alt --no-parity --no-host-fed > $O/${name}_$prov.json 2> $O/${name}_$prov.err
    echo "== $name $prov: $(python -c "import json,sys; d=json.loads(open('$O/${name}_$prov.json').read().strip().splitlines()[-1]); print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step')" 2>&1)"
    python $ROOT/tools/kstats.py $O/${name}_${prov}_trace | head -2
    rm -rf $O/${name}_${prov}_trace
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
